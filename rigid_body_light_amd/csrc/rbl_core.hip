// rbl_core.hip -- context, error plumbing, device buffers, host<->device copies, per-phase timings, parameters / configuration setters.
// Part of the implementation of the C ABI in include/rbl.h (split from the former rbl_api.hip along its sections);
// shared internals are declared in rbl_api_internal.hpp.  Nothing here falls back to a CPU path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "rbl_api_internal.hpp"

// ----------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------
int rbl_fail(rbl_ctx *c, int code, const std::string &msg)
{
  if (c) c->last_error = msg;
  return code;
}

int rbl_hip_fail(rbl_ctx *c, hipError_t e, const char *what)
{
  return rbl_fail(c, RBL_ERR_HIP, std::string("HIP error: ") + hipGetErrorString(e) + " in " + what);
}

int rbl_flags_to_status(rbl_ctx *c, unsigned f)
{
  if (!f) return RBL_OK;
  if (f & RBL_FLAG_BELOW_WALL)  // message of the reference's std::runtime_error, c_rigid_obj.cpp:96
    return rbl_fail(c, RBL_ERR_BELOW_WALL,
                    "A blob has its center below the wall (z<0). Cannot compute mobility- check your configuration.");
  if (f & RBL_FLAG_OVERLAP)     // reference prints this and exit()s, c_rigid_obj.cpp:53-58
    return rbl_fail(c, RBL_ERR_OVERLAP, "ERROR: TWO BLOBS ARE OVERLAPPING OR TOO CLOSELY POSITIONED.");
  if (f & RBL_FLAG_NOT_SPD)
    return rbl_fail(c, RBL_ERR_NOT_SPD, "Cholesky: matrix is not positive definite");
  if (f & RBL_FLAG_INTERNAL)
    return rbl_fail(c, RBL_ERR_HIP, "internal: a tile of the per-body factorisation waited for its dependencies longer than the time limit");
  return rbl_fail(c, RBL_ERR_NONFINITE, "mobility product produced a non-finite value");
}

int rbl_dev_init(rbl_ctx *c)
{
  if (c->dev_ready) return RBL_OK;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return rbl_fail(c, RBL_ERR_NO_DEVICE,
                    "librbl: no HIP device available (this library has no CPU fallback)");
  RBL_HIP(c, hipGetDevice(&c->device));
  hipDeviceProp_t prop;
  RBL_HIP(c, hipGetDeviceProperties(&prop, c->device));
  c->n_cu = prop.multiProcessorCount;
  // the symmetric kernel may use a quarter of the card for its row/column-sum slabs (72 GB of 288: N up to ~580 000
  // blobs); beyond that the ordered kernel (O(N) workspace, ~1.7x the time) takes over
  if (prop.totalGlobalMem / 4 > c->sym_workspace_budget) c->sym_workspace_budget = prop.totalGlobalMem / 4;
  RBL_HIP(c, hipMalloc((void **)&c->d_err, sizeof(unsigned)));
  RBL_HIP(c, hipHostMalloc((void **)&c->h_err, sizeof(unsigned), hipHostMallocDefault));
  RBL_HIP(c, hipMemset(c->d_err, 0, sizeof(unsigned)));
  // auxiliary stream + events for the Cholesky lookahead (optional: failure just disables it)
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // numerically lowest = highest priority
  if (hipStreamCreateWithPriority(&c->chol_aux.stream, hipStreamNonBlocking, prio_hi) == hipSuccess) {
    for (int i = 0; i < 3; ++i)
      if (hipEventCreateWithFlags(&c->chol_aux.ev[i], hipEventDisableTiming) != hipSuccess) {
        c->chol_aux.stream = nullptr;
        break;
      }
  } else {
    c->chol_aux.stream = nullptr;
  }
  c->dev_ready = true;
  return RBL_OK;
}

int rbl_dev_reserve(rbl_ctx *c, RblDevBuf &b, size_t bytes)
{
  if (bytes <= b.bytes) return RBL_OK;
  if (b.p) {
    RBL_HIP(c, hipStreamSynchronize(c->stream));
    RBL_HIP(c, hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    b.p = nullptr;
    return rbl_fail(c, RBL_ERR_ALLOC, std::string("hipMalloc failed: ") + hipGetErrorString(e));
  }
  b.bytes = bytes;
  return RBL_OK;
}

// ---- per-phase timings (include/rbl.h: rbl_set_timing / rbl_get_timings) ----------------------------------------------
static hipEvent_t timing_event(rbl_ctx *c)
{
  if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

RblPhase::RblPhase(rbl_ctx *ctx, int ph) : c(ctx), phase(ph)
{
  if (!c || !c->timing_on || !c->dev_ready) return;
  if (ph == RBL_T_TOTAL) { if (c->timing_total_open) return; }
  else if (c->timing_open >= 0) return;                 // part of the phase that is already open
  a = timing_event(c);
  if (!a || hipEventRecord(a, c->stream) != hipSuccess) { if (a) c->ev_pool.push_back(a); a = nullptr; return; }
  live = true;
  if (ph == RBL_T_TOTAL) c->timing_total_open = true; else c->timing_open = ph;
}

RblPhase::~RblPhase()
{
  if (!live) return;
  if (phase == RBL_T_TOTAL) c->timing_total_open = false; else c->timing_open = -1;
  hipEvent_t b = timing_event(c);
  if (b && hipEventRecord(b, c->stream) == hipSuccess) c->ev_spans.push_back({phase, a, b});
  else { c->ev_pool.push_back(a); if (b) c->ev_pool.push_back(b); }
}

static int timing_resolve(rbl_ctx *c)
{
  if (c->ev_spans.empty()) return RBL_OK;
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  for (const rbl_ctx::TimedSpan &sp : c->ev_spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { c->t_ms[sp.phase] += (double)ms; ++c->t_calls[sp.phase]; }
    c->ev_pool.push_back(sp.a); c->ev_pool.push_back(sp.b);
  }
  c->ev_spans.clear();
  return RBL_OK;
}

extern "C" int rbl_set_timing(rbl_ctx *c, int on)
{
  if (!c) return RBL_ERR_ARG;
  if (on) { int rc = rbl_dev_init(c); if (rc) return rc; }
  c->timing_on = on != 0;
  return RBL_OK;
}

extern "C" int rbl_reset_timings(rbl_ctx *c)
{
  if (!c) return RBL_ERR_ARG;
  int rc = timing_resolve(c); if (rc) return rc;
  for (int i = 0; i < RBL_T_COUNT; ++i) { c->t_ms[i] = 0.0; c->t_calls[i] = 0; }
  return RBL_OK;
}

extern "C" int rbl_get_timings(rbl_ctx *c, double *ms, int64_t *calls)
{
  if (!c) return RBL_ERR_ARG;
  int rc = timing_resolve(c); if (rc) return rc;
  for (int i = 0; i < RBL_T_COUNT; ++i) { if (ms) ms[i] = c->t_ms[i]; if (calls) calls[i] = c->t_calls[i]; }
  return RBL_OK;
}

int need_params(rbl_ctx *c)
{
  if (!c) return RBL_ERR_ARG;
  if (!c->S.params_set) return rbl_fail(c, RBL_ERR_STATE, "setParameters has not been called");
  return RBL_OK;
}

int need_config(rbl_ctx *c)
{
  int rc = need_params(c);
  if (rc) return rc;
  // the reference only prints "ERROR CONFIG NOT INITIALIZED YET!!" (:296-298) and
  // then reads unset members; we return an error instead
  if (!c->S.cfg_set) return rbl_fail(c, RBL_ERR_STATE, "ERROR CONFIG NOT INITIALIZED YET!!");
  return RBL_OK;
}


// Host <-> device copies of the host-pointer API.  Caller arrays are pageable; both
// hipMemcpy and hipMemcpyAsync then pin the caller's pages on the fly (measured ~20 ms for
// a fresh 3 MB numpy array on this stack).  Large copies therefore go through the context's
// own pinned staging buffer in 32 MB chunks (DMA + one CPU memcpy); small ones stay direct.
static constexpr size_t RBL_STAGE_BYTES = 32u << 20;

static int stage_ready(rbl_ctx *c)
{
  if (c->h_stage) return RBL_OK;
  RBL_HIP(c, hipHostMalloc(&c->h_stage, RBL_STAGE_BYTES, hipHostMallocDefault));
  return RBL_OK;
}

// Krylov coefficients (<= 512 doubles, slot 0 or 1) to the device through a pinned buffer of the context: a true
// asynchronous copy, so the stream is not drained for it (a pageable source would have to outlive the copy).  The slot is
// written again one solve later at the earliest, behind that solve's own synchronisations.
int upload_coef(rbl_ctx *c, double *d_dst, const double *src, int count, int slot)
{
  if (count > 512 || slot < 0 || slot > 1) {
    RBL_HIP(c, hipMemcpyAsync(d_dst, src, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    RBL_HIP(c, hipStreamSynchronize(c->stream));
    return RBL_OK;
  }
  if (!c->h_coef) RBL_HIP(c, hipHostMalloc((void **)&c->h_coef, sizeof(double) * 1024, hipHostMallocDefault));
  std::memcpy(c->h_coef + 512 * slot, src, sizeof(double) * (size_t)count);
  RBL_HIP(c, hipMemcpyAsync(d_dst, c->h_coef + 512 * slot, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
  return RBL_OK;
}

int copy_h2d(rbl_ctx *c, void *dst, const void *src, size_t bytes)
{
  if (bytes <= (64u << 10)) { RBL_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); return RBL_OK; }
  int rc = stage_ready(c); if (rc) return rc;
  for (size_t off = 0; off < bytes; off += RBL_STAGE_BYTES) {
    const size_t nb = (bytes - off < RBL_STAGE_BYTES) ? bytes - off : RBL_STAGE_BYTES;
    RBL_HIP(c, hipStreamSynchronize(c->stream));   // staging buffer free again
    std::memcpy(c->h_stage, (const char *)src + off, nb);
    RBL_HIP(c, hipMemcpyAsync((char *)dst + off, c->h_stage, nb, hipMemcpyHostToDevice, c->stream));
  }
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  return RBL_OK;
}

int copy_d2h(rbl_ctx *c, void *dst, const void *src, size_t bytes)
{
  if (bytes <= (64u << 10)) { RBL_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); return RBL_OK; }
  int rc = stage_ready(c); if (rc) return rc;
  for (size_t off = 0; off < bytes; off += RBL_STAGE_BYTES) {
    const size_t nb = (bytes - off < RBL_STAGE_BYTES) ? bytes - off : RBL_STAGE_BYTES;
    RBL_HIP(c, hipMemcpyAsync(c->h_stage, (const char *)src + off, nb, hipMemcpyDeviceToHost, c->stream));
    RBL_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy((char *)dst + off, c->h_stage, nb);
  }
  return RBL_OK;
}

// Small device -> host read that the host needs NOW (Krylov coefficients, norms): through a pinned buffer of the context and
// a stream drain -- 11.6 us on MI355X; with a pageable target the runtime stages the copy itself and the same read costs
// 21 us (tools/launch_costs.hip).
constexpr size_t RBL_PIN_BYTES = (size_t)1 << 20;
int read_back(rbl_ctx *c, void *dst, const void *d_src, size_t bytes)
{
  if (bytes <= RBL_PIN_BYTES) {
    if (!c->h_pin) RBL_HIP(c, hipHostMalloc(&c->h_pin, RBL_PIN_BYTES, hipHostMallocDefault));
    RBL_HIP(c, hipMemcpyAsync(c->h_pin, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    RBL_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(dst, c->h_pin, bytes);
    return RBL_OK;
  }
  RBL_HIP(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  return RBL_OK;
}

// read + clear the latched device flags (stream must be idle for h_err to be valid)
int finish_and_check(rbl_ctx *c)
{
  RBL_HIP(c, hipMemcpyAsync(c->h_err, c->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  RBL_HIP(c, hipMemsetAsync(c->d_err, 0, sizeof(unsigned), c->stream));
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  return rbl_flags_to_status(c, *c->h_err);
}

// Enqueue rows [row_begin,row_end) of U = [B] M [B] F on the context stream, choosing the
// kernel variant: 1 = symmetric (unordered pairs, needs the full row range), 0 = ordered rows.
// tune_variant: 0 = heuristic, 1 = force ordered, 2 = force symmetric.
RblParams ctx_params(const rbl_ctx *c)
{
  RblParams P = rbl_make_params(c->S.a, c->S.eta);
  P.no_damp = c->no_damp ? 1 : 0;
  return P;
}

rbl_ctx *rbl_create(void) { return new (std::nothrow) rbl_ctx(); }

void rbl_destroy(rbl_ctx *c)
{
  if (!c) return;
  if (c->dev_ready) {
    (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    if (c->ev_check) (void)hipEventDestroy(c->ev_check);
    RblDevBuf *bufs[] = {&c->d_r, &c->d_F, &c->d_U, &c->d_part, &c->d_W, &c->d_cfg,
                         &c->d_XQ, &c->d_mat, &c->d_tmp, &c->d_tmp2, &c->d_chol,
                         &c->d_lever, &c->d_pos, &c->d_invM2, &c->d_NL, &c->d_sad, &c->d_blkL, &c->d_blkLinv, &c->d_blkX, &c->d_blkTmp, &c->d_blkXf, &c->d_blkAug, &c->d_commStage, &c->d_tlQ, &c->d_tlCb, &c->d_tlCs, &c->d_tlA, &c->d_tlLinv, &c->d_tlX, &c->d_tlT, &c->d_tlZ, &c->d_ktl, &c->d_bfL, &c->d_bfLinv, &c->d_bfX, &c->d_bfPC, &c->d_pcw, &c->d_pcMK, &c->d_bd, &c->d_bd2, &c->d_gm, &c->d_step, &c->d_hist};
    for (RblDevBuf *b : bufs)
      if (b->p) (void)hipFree(b->p);
    if (c->chol_aux.stream) {
      (void)hipStreamSynchronize(c->chol_aux.stream);
      for (int i = 0; i < 3; ++i)
        if (c->chol_aux.ev[i]) (void)hipEventDestroy(c->chol_aux.ev[i]);
      (void)hipStreamDestroy(c->chol_aux.stream);
    }
    for (const rbl_ctx::TimedSpan &sp : c->ev_spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_err2) (void)hipFree(c->d_err2);
    if (c->h_err) (void)hipHostFree(c->h_err);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_coef) (void)hipHostFree(c->h_coef);
  }
  delete c;
}

const char *rbl_precision(void) { return "double"; }

const char *rbl_last_error(const rbl_ctx *c) { return c ? c->last_error.c_str() : "null context"; }

int rbl_set_parameters(rbl_ctx *c, double a, double dt, double kBT, double eta, const double *cfg,
                       int N_blb)
{
  if (!c || !cfg || N_blb <= 0) return rbl_fail(c, RBL_ERR_ARG, "setParameters: bad arguments");
  RblBodyState &S = c->S;
  S.a = a; S.dt = dt; S.kBT = kBT; S.eta = eta;
  S.ref_cfg.assign(cfg, cfg + (size_t)3 * N_blb);
  double mean[3] = {0, 0, 0};  // removeMean, c_rigid_obj.cpp:176-181 (on our private copy)
  for (int k = 0; k < N_blb; ++k)
    for (int d = 0; d < 3; ++d) mean[d] += S.ref_cfg[3 * k + d];
  for (int d = 0; d < 3; ++d) mean[d] /= (double)N_blb;
  for (int k = 0; k < N_blb; ++k)
    for (int d = 0; d < 3; ++d) S.ref_cfg[3 * k + d] -= mean[d];
  double rmax2 = 0.0;
  for (int k = 0; k < N_blb; ++k) {
    const double *p_ = &S.ref_cfg[3 * (size_t)k];
    rmax2 = std::max(rmax2, p_[0] * p_[0] + p_[1] * p_[1] + p_[2] * p_[2]);
  }
  c->body_radius = std::sqrt(rmax2) + a;               // the sphere the two-level factor's far-field model gives a body
  c->tl_valid = false; c->pc_keep_once = false;
  S.N_blb = N_blb;
  S.params_set = true;
  S.M_scale = 1.0;
  c->dev_bodies_valid = false; c->dev_pc_valid = false; c->dev_blk_valid = false; c->dev_xq_valid = false;
  c->bf_valid = false;                                 // the body-frame factor belongs to (a, eta, cfg)
  c->dev_cfg_valid = false;
  return RBL_OK;
}

int rbl_set_blk_pc(rbl_ctx *c, int v) { if (!c) return RBL_ERR_ARG; c->S.block_pc = v != 0; c->S.pc_set = false; c->dev_pc_valid = false; return RBL_OK; }
int rbl_set_wall_pc(rbl_ctx *c, int v) { if (!c) return RBL_ERR_ARG; c->S.wall = v != 0; c->dev_pc_valid = false; c->dev_blk_valid = false; return RBL_OK; }

int rbl_set_config(rbl_ctx *c, const double *X, const double *Q, int N_bod)
{
  if (!c || !X || !Q || N_bod <= 0) return rbl_fail(c, RBL_ERR_ARG, "setConfig: bad arguments");
  RblBodyState &S = c->S;
  if (S.N_bod != N_bod) { c->dev_blk_valid = false; c->tl_valid = false; c->tl_age = 0; }   // (a kept coarse operator of another body count is not even the right size)
  c->pc_keep_once = false;                               // (the promise of rbl_evolve_X_Q_RFD was about the configuration it committed)
  S.N_bod = N_bod;
  S.X.assign(X, X + (size_t)3 * N_bod);
  S.Q.resize((size_t)4 * N_bod);
  for (int j = 0; j < N_bod; ++j) {  // scalar-first, normalised (:212-216)
    const double w = Q[4 * j], x = Q[4 * j + 1], y = Q[4 * j + 2], z = Q[4 * j + 3];
    const double nrm = std::sqrt(w * w + x * x + y * y + z * z);
    S.Q[4 * j] = w / nrm; S.Q[4 * j + 1] = x / nrm; S.Q[4 * j + 2] = y / nrm; S.Q[4 * j + 3] = z / nrm;
  }
  S.cfg_set = true;
  S.K_set = false;
  c->dev_bodies_valid = false; c->dev_pc_valid = false; c->dev_xq_valid = false;   // (block factors: aged in sync_bodies)
  // NOTE the reference does NOT reset PC_mat_Set here (SURVEY.md 8b "state quirks");
  // a stale preconditioner after set_config is a trap, so we do invalidate it.
  S.pc_set = false;
  return RBL_OK;
}

int rbl_get_config(const rbl_ctx *c, double *X, double *Q)
{
  if (!c || !c->S.cfg_set) return RBL_ERR_STATE;
  std::memcpy(X, c->S.X.data(), sizeof(double) * c->S.X.size());
  std::memcpy(Q, c->S.Q.data(), sizeof(double) * c->S.Q.size());
  return RBL_OK;
}

int rbl_get_sizes(const rbl_ctx *c, int *N_bod, int *N_blb)
{
  if (!c) return RBL_ERR_ARG;
  if (N_bod) *N_bod = c->S.N_bod;
  if (N_blb) *N_blb = c->S.N_blb;
  return RBL_OK;
}

int rbl_set_stream(rbl_ctx *c, void *s)
{
  if (!c) return RBL_ERR_ARG;
  int rc = rbl_dev_init(c); if (rc) return rc;
  c->stream = (hipStream_t)s;
  return RBL_OK;
}

int rbl_sync_check(rbl_ctx *c)
{
  if (!c) return RBL_ERR_ARG;
  int rc = rbl_dev_init(c); if (rc) return rc;
  return finish_and_check(c);
}
