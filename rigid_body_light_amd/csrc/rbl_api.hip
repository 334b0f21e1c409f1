// rbl_api.hip -- implementation of the C ABI in include/rbl.h.
// Host bookkeeping lives in rbl_host.cpp, kernels in rbl_kernels.hip / rbl_dense.hip.
// Nothing here falls back to a CPU path: compute entry points return
// RBL_ERR_NO_DEVICE when no HIP device can be initialised.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "rbl_internal.hpp"

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

static int need_params(rbl_ctx *c)
{
  if (!c) return RBL_ERR_ARG;
  if (!c->S.params_set) return rbl_fail(c, RBL_ERR_STATE, "setParameters has not been called");
  return RBL_OK;
}

static int need_config(rbl_ctx *c)
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
static int upload_coef(rbl_ctx *c, double *d_dst, const double *src, int count, int slot)
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

static int copy_h2d(rbl_ctx *c, void *dst, const void *src, size_t bytes)
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

static int copy_d2h(rbl_ctx *c, void *dst, const void *src, size_t bytes)
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
static int read_back(rbl_ctx *c, void *dst, const void *d_src, size_t bytes)
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
static int finish_and_check(rbl_ctx *c)
{
  RBL_HIP(c, hipMemcpyAsync(c->h_err, c->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  RBL_HIP(c, hipMemsetAsync(c->d_err, 0, sizeof(unsigned), c->stream));
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  return rbl_flags_to_status(c, *c->h_err);
}

// Enqueue rows [row_begin,row_end) of U = [B] M [B] F on the context stream, choosing the
// kernel variant: 1 = symmetric (unordered pairs, needs the full row range), 0 = ordered rows.
// tune_variant: 0 = heuristic, 1 = force ordered, 2 = force symmetric.
static RblParams ctx_params(const rbl_ctx *c)
{
  RblParams P = rbl_make_params(c->S.a, c->S.eta);
  P.no_damp = c->no_damp ? 1 : 0;
  return P;
}

// ---- multi-GPU (rbl_set_comm): this rank's bodies (contiguous split, sizes differ by at most one -- the partition of
// rigid_body_light_amd/dist.py) and the caller's sum all-reduce.  Per-body work (Cholesky factors, substitutions) is done for
// the own bodies only, written into a zeroed full-length vector and completed by the all-reduce (an all-gather by sums).
// (a callback with world == 1 keeps the multi-GPU code path on, with one share: what the world-1 RCCL test drives)
static bool comm_on(const rbl_ctx *c) { return c->comm_fn != nullptr; }

static void comm_body_range(const rbl_ctx *c, int *b0, int *b1)
{
  const int nb = c->S.N_bod, base = nb / c->comm_world, rem = nb % c->comm_world, r = c->comm_rank;
  *b0 = r * base + (r < rem ? r : rem);
  *b1 = *b0 + base + (r < rem ? 1 : 0);
}

static int comm_allreduce(rbl_ctx *c, double *d_buf, int64_t count)
{
  RblPhase ph(c, RBL_T_COLLECTIVE);
  if (c->comm_fn(c->comm_user, d_buf, count)) return rbl_fail(c, RBL_ERR_HIP, "all-reduce callback failed");
  return RBL_OK;
}

static int apply_M_enqueue(rbl_ctx *c, bool wall, const double *d_F, const double *d_r, int64_t nbl,
                           int64_t row_begin, int64_t row_end, double *d_out)
{
  const RblParams P = ctx_params(c);
  const bool full = (row_begin == 0 && row_end == nbl);
  bool sym = full;   // measured faster at every size, N = 120 ... 128 400 (profiles/r01_apply_M_all_configs.md)
  // its row/column-sum slabs grow like N^2/128 * 24 B (1.7 GB at 128 400 blobs, ~100 GB at 10^6):
  // beyond a budget the ordered kernel (O(N) workspace, ~1.6x the time) takes over
  if (sym && rbl_apply_M_sym_bytes(nbl, c->n_cu, 1, 1, c->sym_tune) > c->sym_workspace_budget) sym = false;
  if (c->tune_variant == 1) sym = false;
  if (c->tune_variant == 2) sym = full;
  int rc;
  if (full && comm_on(c)) {   // multi-GPU: this rank's tile pairs, then the sum over the ranks
    if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, c->comm_world, 1, c->sym_tune)))) return rc;
    RblSymTune tune = c->sym_tune;
    if (c->force_relaxed) tune.relaxed = 1;
    {
      RblPhase ph(c, RBL_T_PRODUCT);
      rbl_launch_apply_M_sym(c->stream, P, wall, d_F, d_r, nbl, c->comm_rank, c->comm_world, d_out, (double *)c->d_part.p,
                             c->n_cu, c->d_err, 1, tune);
    }
    return comm_allreduce(c, d_out, 3 * nbl);
  }
  RblPhase ph(c, RBL_T_PRODUCT);
  if (sym) {
    if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, 1, 1, c->sym_tune)))) return rc;
    RblSymTune tune = c->sym_tune;
    if (c->force_relaxed) tune.relaxed = 1;
    rbl_launch_apply_M_sym(c->stream, P, wall, d_F, d_r, nbl, 0, 1, d_out, (double *)c->d_part.p, c->n_cu,
                           c->d_err, 1, tune);
  } else {
    int js = 1;
    const size_t pb = rbl_apply_M_part_bytes(nbl, row_end - row_begin, c->n_cu, c->tune_jsplit, &js);
    if ((rc = rbl_dev_reserve(c, c->d_part, pb))) return rc;
    rbl_launch_apply_M(c->stream, P, wall, d_F, d_r, nbl, row_begin, row_end, d_out, (double *)c->d_part.p,
                       js, 0, c->d_err);
  }
  return RBL_OK;
}

// nrhs right-hand sides, column-major n3 x nrhs on the device.  >= 4 vectors go through the
// fp64-MFMA kernel in passes of 16; fewer are cheaper one by one on the symmetric kernel.
// tune_variant 3 forces the MFMA kernel, 1/2 force the single-RHS kernels.
static int apply_M_multi_enqueue(rbl_ctx *c, bool wall, const double *d_F, const double *d_r, int64_t nbl,
                                 int nrhs, double *d_out)
{
  const int64_t n3 = 3 * nbl;
  bool mfma = nrhs >= 4;
  if (c->tune_variant == 3) mfma = true;
  if (c->tune_variant == 1 || c->tune_variant == 2) mfma = false;
  int rc;
  if (comm_on(c)) {   // multi-GPU: pairs of vectors through the sharded two-vector kernel, one all-reduce per pair
    int k = 0;
    for (; k + 2 <= nrhs; k += 2) {
      if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, c->comm_world, 2, c->sym_tune)))) return rc;
      RblSymTune tune = c->sym_tune;
      if (c->force_relaxed) tune.relaxed = 1;
      {
        RblPhase ph(c, RBL_T_PRODUCT);
        rbl_launch_apply_M_sym(c->stream, ctx_params(c), wall, d_F + (size_t)k * n3, d_r, nbl, c->comm_rank, c->comm_world,
                               d_out + (size_t)k * n3, (double *)c->d_part.p, c->n_cu, c->d_err, 2, tune);
      }
      if ((rc = comm_allreduce(c, d_out + (size_t)k * n3, 2 * n3))) return rc;
    }
    for (; k < nrhs; ++k)
      if ((rc = apply_M_enqueue(c, wall, d_F + (size_t)k * n3, d_r, nbl, 0, nbl, d_out + (size_t)k * n3))) return rc;
    return RBL_OK;
  }
  RblPhase ph(c, RBL_T_PRODUCT);
  if (!mfma) {   // 1-3 vectors: pairs of vectors through the two-vector symmetric kernel, a single one alone
    int k = 0;
    const bool sym2 = c->tune_variant != 1 && rbl_apply_M_sym_bytes(nbl, c->n_cu, 1, 2, c->sym_tune) <= c->sym_workspace_budget;
    for (; sym2 && k + 2 <= nrhs; k += 2) {
      if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, 1, 2, c->sym_tune)))) return rc;
      RblSymTune tune = c->sym_tune;
      if (c->force_relaxed) tune.relaxed = 1;
      rbl_launch_apply_M_sym(c->stream, ctx_params(c), wall, d_F + (size_t)k * n3, d_r, nbl, 0, 1,
                             d_out + (size_t)k * n3, (double *)c->d_part.p, c->n_cu, c->d_err, 2, tune);
    }
    for (; k < nrhs; ++k)
      if ((rc = apply_M_enqueue(c, wall, d_F + (size_t)k * n3, d_r, nbl, 0, nbl, d_out + (size_t)k * n3))) return rc;
    return RBL_OK;
  }
  if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_mrhs_bytes(nbl, c->n_cu)))) return rc;
  const RblParams P = ctx_params(c);
  for (int k = 0; k < nrhs; k += 16) {
    const int nb = (nrhs - k < 16) ? nrhs - k : 16;
    rbl_launch_apply_M_mrhs(c->stream, P, wall, d_F + (size_t)k * n3, d_r, nbl, nb, d_out + (size_t)k * n3,
                            (double *)c->d_part.p, c->n_cu, c->d_err);
  }
  return RBL_OK;
}

extern "C" {

// ============================================================================
// 1. reference-bound methods
// ============================================================================
rbl_ctx *rbl_create(void) { return new (std::nothrow) rbl_ctx(); }

void rbl_destroy(rbl_ctx *c)
{
  if (!c) return;
  if (c->dev_ready) {
    (void)hipStreamSynchronize(c->stream);
    RblDevBuf *bufs[] = {&c->d_r, &c->d_F, &c->d_U, &c->d_part, &c->d_W, &c->d_cfg,
                         &c->d_XQ, &c->d_mat, &c->d_tmp, &c->d_tmp2, &c->d_chol,
                         &c->d_lever, &c->d_pos, &c->d_invM2, &c->d_NL, &c->d_sad, &c->d_blkL, &c->d_blkLinv, &c->d_blkX, &c->d_blkTmp, &c->d_blkXf, &c->d_blkAug, &c->d_tlQ, &c->d_tlCb, &c->d_tlCs, &c->d_tlA, &c->d_tlLinv, &c->d_tlX, &c->d_tlT, &c->d_tlZ, &c->d_ktl, &c->d_bfL, &c->d_bfLinv, &c->d_bfX, &c->d_bfPC, &c->d_pcw, &c->d_pcMK, &c->d_bd, &c->d_bd2, &c->d_gm, &c->d_step, &c->d_hist};
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
  c->tl_valid = false;
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
  if (S.N_bod != N_bod) c->dev_blk_valid = false;
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

int rbl_set_K_mats(rbl_ctx *c)
{
  int rc = need_config(c);
  if (rc) return rc;
  return rbl_body_set_K(c->S, c->last_error);
}

static int need_K(rbl_ctx *c)
{
  int rc = need_config(c);
  if (rc) return rc;
  if (!c->S.K_set) return rbl_body_set_K(c->S, c->last_error);
  return RBL_OK;
}

int rbl_K_x_U(rbl_ctx *c, const double *U, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  rbl_body_K_x_U(c->S, U, out);
  return RBL_OK;
}

int rbl_KT_x_Lam(rbl_ctx *c, const double *lam, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  rbl_body_KT_x_Lam(c->S, lam, out);
  return RBL_OK;
}

int rbl_Kinv_x_V(rbl_ctx *c, const double *V, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  rbl_body_Kinv_x_V(c->S, V, out);
  return RBL_OK;
}

int rbl_KTinv_x_F(rbl_ctx *c, const double *F, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  rbl_body_KTinv_x_F(c->S, F, out);
  return RBL_OK;
}

// (X, Q, ref_cfg) resident on the device: uploaded once per configuration change (pageable host
// vectors -> one synchronisation there), so the per-step position kernel is launch-only.
static int ensure_xq_dev(rbl_ctx *c)
{
  if (c->dev_xq_valid) return RBL_OK;
  RblBodyState &S = c->S;
  int rc = rbl_dev_reserve(c, c->d_XQ, sizeof(double) * 7 * (size_t)S.N_bod); if (rc) return rc;
  rc = rbl_dev_reserve(c, c->d_cfg, sizeof(double) * 3 * (size_t)S.N_blb); if (rc) return rc;
  double *dX = (double *)c->d_XQ.p;
  // one copy for [X | Q] (a time step uploads the configuration four times), the reference shape only when it changed
  c->h_xq.resize(7 * (size_t)S.N_bod);
  std::memcpy(c->h_xq.data(), S.X.data(), sizeof(double) * 3 * (size_t)S.N_bod);
  std::memcpy(c->h_xq.data() + 3 * (size_t)S.N_bod, S.Q.data(), sizeof(double) * 4 * (size_t)S.N_bod);
  RBL_HIP(c, hipMemcpyAsync(dX, c->h_xq.data(), sizeof(double) * 7 * (size_t)S.N_bod, hipMemcpyHostToDevice, c->stream));
  if (!c->dev_cfg_valid)
    RBL_HIP(c, hipMemcpyAsync(c->d_cfg.p, S.ref_cfg.data(), sizeof(double) * 3 * (size_t)S.N_blb, hipMemcpyHostToDevice, c->stream));
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  c->dev_xq_valid = true; c->dev_cfg_valid = true;
  return RBL_OK;
}

// positions of bodies [b0,b1) -> d_out
static int positions_dev(rbl_ctx *c, int b0, int b1, double *d_out)
{
  int rc = ensure_xq_dev(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const double *dX = (const double *)c->d_XQ.p, *dQ = dX + 3 * (size_t)S.N_bod;
  rbl_launch_blob_positions(c->stream, dX, dQ, (const double *)c->d_cfg.p, S.N_blb, b0, b1, d_out);
  return RBL_OK;
}

int rbl_blob_positions_dev(rbl_ctx *c, int body_begin, int body_end, double *d_out)
{
  int rc = need_config(c); if (rc) return rc;
  rc = rbl_dev_init(c); if (rc) return rc;
  if (body_begin < 0 || body_end > c->S.N_bod || body_begin > body_end)
    return rbl_fail(c, RBL_ERR_SIZE, "blob_positions_dev: body range out of bounds");
  return positions_dev(c, body_begin, body_end, d_out);
}

int rbl_multi_body_pos(rbl_ctx *c, double *out)
{
  int rc = need_config(c); if (rc) return rc;
  rc = rbl_dev_init(c); if (rc) return rc;
  const size_t n3 = (size_t)3 * c->S.N_bod * c->S.N_blb;
  rc = rbl_dev_reserve(c, c->d_r, sizeof(double) * n3); if (rc) return rc;
  rc = positions_dev(c, 0, c->S.N_bod, (double *)c->d_r.p); if (rc) return rc;
  { int rc__ = copy_d2h(c, out, c->d_r.p, sizeof(double) * n3); if (rc__) return rc__; }
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  return RBL_OK;
}

static int apply_M_host(rbl_ctx *c, const double *F, const double *r, int64_t n3, int nrhs, double *out)
{
  int rc = need_params(c); if (rc) return rc;
  if (n3 <= 0 || n3 % 3 != 0 || nrhs < 1)
    return rbl_fail(c, RBL_ERR_SIZE, "Positions and forces must have total length 3N, where N is the number of blobs");
  rc = rbl_dev_init(c); if (rc) return rc;
  const int64_t nbl = n3 / 3;
  const size_t vb = sizeof(double) * (size_t)n3;
  if ((rc = rbl_dev_reserve(c, c->d_r, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_F, vb * nrhs))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_U, vb * nrhs))) return rc;
  { int rc__ = copy_h2d(c, c->d_r.p, r, vb); if (rc__) return rc__; }
  { int rc__ = copy_h2d(c, c->d_F.p, F, vb * nrhs); if (rc__) return rc__; }
  if ((rc = apply_M_multi_enqueue(c, c->S.wall, (const double *)c->d_F.p, (const double *)c->d_r.p, nbl, nrhs,
                                  (double *)c->d_U.p)))
    return rc;
  { int rc__ = copy_d2h(c, out, c->d_U.p, vb * nrhs); if (rc__) return rc__; }
  return finish_and_check(c);
}

int rbl_apply_M(rbl_ctx *c, const double *F, const double *r_vecs, int64_t n3, double *out)
{
  return apply_M_host(c, F, r_vecs, n3, 1, out);
}

int rbl_apply_M_multi(rbl_ctx *c, const double *F, const double *r_vecs, int64_t n3, int nrhs,
                      double *out)
{
  return apply_M_host(c, F, r_vecs, n3, nrhs, out);
}

// ---- preconditioner --------------------------------------------------------
// diag_invM (:489-543): per-blob inverse of the self block, times 8 pi eta a.
static int build_diag_invM(rbl_ctx *c)
{
  RblBodyState &S = c->S;
  const size_t N = (size_t)S.N_bod * S.N_blb;
  S.invM_diag.assign(9 * N, 0.0);
  const double nf = 8.0 * M_PI * S.eta * S.a;
  for (size_t i = 0; i < N; ++i) {
    double dxx = 4.0 / 3.0, dzz = 4.0 / 3.0;
    if (S.wall) {  // self wall term (:98-104), h = z_i / a
      const size_t b = i / S.N_blb;
      const double z = S.X[3 * b + 2] + S.lever[3 * i + 2];
      const double h = z / S.a;
      if (h < 0.0) return rbl_flags_to_status(c, RBL_FLAG_BELOW_WALL);
      const double iz = 1.0 / h, iz3 = iz * iz * iz, iz5 = iz3 * iz * iz;
      dxx += -(9 * iz - 2 * iz3 + iz5) / 12.0;
      dzz += -(9 * iz - 4 * iz3 + iz5) / 6.0;
    }
    S.invM_diag[9 * i] = nf / dxx;
    S.invM_diag[9 * i + 4] = nf / dxx;
    S.invM_diag[9 * i + 8] = nf / dzz;
  }
  return RBL_OK;
}

int rbl_apply_PC_dev(rbl_ctx *c, const double *d_in, double *d_out);

int rbl_apply_PC(rbl_ctx *c, const double *in, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  if (c->S.block_pc) {
    // Block_diag_invM (:461-487) lives on the GPU: batched per-body Cholesky + substitution
    if ((rc = rbl_dev_init(c))) return rc;
    const size_t nv = (size_t)3 * c->S.N_bod * c->S.N_blb + (size_t)6 * c->S.N_bod;
    if ((rc = rbl_dev_reserve(c, c->d_tmp, sizeof(double) * 2 * nv))) return rc;
    double *din = (double *)c->d_tmp.p, *dout = din + nv;
    if ((rc = copy_h2d(c, din, in, sizeof(double) * nv))) return rc;
    if ((rc = rbl_apply_PC_dev(c, din, dout))) return rc;
    if ((rc = copy_d2h(c, out, dout, sizeof(double) * nv))) return rc;
    return finish_and_check(c);
  }
  if (!c->S.pc_set) {
    rc = build_diag_invM(c);
    if (rc) return rc;
  }
  rc = rbl_body_apply_PC(c->S, in, out, c->last_error);
  return rc;
}

// ---- K / Kinv as CSC (get_K :978, get_Kinv :986) ----------------------------
int rbl_get_K_csc(rbl_ctx *c, int64_t *nnz, int64_t *nrows, int64_t *ncols, double *data,
                  int32_t *indices, int32_t *indptr)
{
  int rc = need_K(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const int nb = S.N_bod, nl = S.N_blb;
  if (nnz) *nnz = (int64_t)9 * nb * nl;
  if (nrows) *nrows = (int64_t)3 * nb * nl;
  if (ncols) *ncols = (int64_t)6 * nb;
  if (!data || !indices || !indptr) return RBL_OK;
  int64_t p = 0;
  for (int b = 0; b < nb; ++b) {
    const int32_t r0 = 3 * b * nl;
    for (int cc = 0; cc < 6; ++cc) {
      indptr[6 * b + cc] = (int32_t)p;
      for (int k = 0; k < nl; ++k) {
        const double *l = &S.lever[3 * ((size_t)b * nl + k)];
        const int32_t r = r0 + 3 * k;
        switch (cc) {  // structural pattern of :370-382 (explicit zeros are kept)
          case 0: indices[p] = r;     data[p++] = 1.0; break;
          case 1: indices[p] = r + 1; data[p++] = 1.0; break;
          case 2: indices[p] = r + 2; data[p++] = 1.0; break;
          case 3: indices[p] = r + 1; data[p++] = -l[2]; indices[p] = r + 2; data[p++] = l[1]; break;
          case 4: indices[p] = r;     data[p++] = l[2];  indices[p] = r + 2; data[p++] = -l[0]; break;
          case 5: indices[p] = r;     data[p++] = -l[1]; indices[p] = r + 1; data[p++] = l[0]; break;
        }
      }
    }
  }
  indptr[6 * nb] = (int32_t)p;
  return RBL_OK;
}

int rbl_get_Kinv_csc(rbl_ctx *c, int64_t *nnz, int64_t *nrows, int64_t *ncols, double *data,
                     int32_t *indices, int32_t *indptr)
{
  int rc = need_K(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const int nb = S.N_bod, nl = S.N_blb;
  // Kinv = KTKi * K^T, pruned (:390): column 3k+d holds KTKi_b * (row 3k+d of K)^T
  int64_t p = 0;
  const bool fill = data && indices && indptr;
  for (int b = 0; b < nb; ++b) {
    const double *B = &S.KTKinv[(size_t)36 * b];
    for (int k = 0; k < nl; ++k) {
      const double *l = &S.lever[3 * ((size_t)b * nl + k)];
      const double Krow[3][6] = {{1, 0, 0, 0, l[2], -l[1]}, {0, 1, 0, -l[2], 0, l[0]}, {0, 0, 1, l[1], -l[0], 0}};
      for (int d = 0; d < 3; ++d) {
        const int64_t col = 3 * ((int64_t)b * nl + k) + d;
        if (fill) indptr[col] = (int32_t)p;
        for (int rr = 0; rr < 6; ++rr) {
          double v = 0.0;
          for (int q = 0; q < 6; ++q) v += B[6 * rr + q] * Krow[d][q];
          if (std::fabs(v) > 1e-12) {  // Eigen pruned(): |v| <= dummy_precision dropped
            if (fill) { indices[p] = 6 * b + rr; data[p] = v; }
            ++p;
          }
        }
      }
    }
  }
  if (fill) indptr[(int64_t)3 * nb * nl] = (int32_t)p;
  if (nnz) *nnz = p;
  if (nrows) *nrows = (int64_t)6 * nb;
  if (ncols) *ncols = (int64_t)3 * nb * nl;
  return RBL_OK;
}

int rbl_evolve_X_Q(rbl_ctx *c, const double *U)
{
  int rc = need_config(c); if (rc) return rc;
  RblBodyState &S = c->S;
  std::vector<double> Udt((size_t)6 * S.N_bod), Xo, Qo;
  for (size_t i = 0; i < Udt.size(); ++i) Udt[i] = U[i] * S.dt;  // :869 (on a copy)
  rbl_body_update_X_Q(S, Udt.data(), Xo, Qo);
  S.X.swap(Xo);
  S.Q.swap(Qo);
  c->dev_bodies_valid = false; c->dev_pc_valid = false; c->dev_xq_valid = false;
  rc = rbl_body_set_K(S, c->last_error);                          // :876
  S.pc_set = false;                                               // :877
  return rc;
}

// ============================================================================
// 2. unbound reference members + extensions
// ============================================================================
int rbl_rotne_prager_tensor(rbl_ctx *c, const double *r, int64_t n3, int scale_damp, double *out)
{
  int rc = need_params(c); if (rc) return rc;
  if (n3 <= 0 || n3 % 3 != 0) return rbl_fail(c, RBL_ERR_SIZE, "r_vecs must have length 3N");
  rc = rbl_dev_init(c); if (rc) return rc;
  const size_t mb = sizeof(double) * (size_t)n3 * (size_t)n3;
  if ((rc = rbl_dev_reserve(c, c->d_r, sizeof(double) * n3))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_mat, mb))) return rc;
  { int rc__ = copy_h2d(c, c->d_r.p, r, sizeof(double) * n3); if (rc__) return rc__; }
  rbl_launch_build_M(c->stream, rbl_make_params(c->S.a, c->S.eta), c->S.wall, scale_damp != 0,
                     (const double *)c->d_r.p, n3 / 3, (double *)c->d_mat.p, c->d_err);
  { int rc__ = copy_d2h(c, out, c->d_mat.p, mb); if (rc__) return rc__; }
  return finish_and_check(c);
}

int rbl_cholesky_lower(rbl_ctx *c, double *M, int64_t n)
{
  if (!c || !M || n <= 0) return rbl_fail(c, RBL_ERR_ARG, "cholesky_lower: bad arguments");
  int rc = rbl_dev_init(c); if (rc) return rc;
  const size_t mb = sizeof(double) * (size_t)n * (size_t)n;
  if ((rc = rbl_dev_reserve(c, c->d_mat, mb))) return rc;
  { int rc__ = copy_h2d(c, c->d_mat.p, M, mb); if (rc__) return rc__; }
  if ((rc = rbl_dev_reserve(c, c->d_chol, rbl_cholesky_work_bytes(n)))) return rc;
  rc = rbl_launch_cholesky(c->stream, (double *)c->d_mat.p, n, true, c->d_err, (double *)c->d_chol.p, c->d_chol.bytes, &c->chol_aux);
  if (rc) return rbl_fail(c, rc, "cholesky launch failed");
  { int rc__ = copy_d2h(c, M, c->d_mat.p, mb); if (rc__) return rc__; }
  return finish_and_check(c);
}

int rbl_set_lanczos(rbl_ctx *c, int max_iter, double tol)
{
  if (!c || max_iter < 2 || !(tol > 0)) return RBL_ERR_ARG;
  c->lanczos_max_iter = max_iter; c->lanczos_tol = tol;
  return RBL_OK;
}

int rbl_get_lanczos_report(const rbl_ctx *c, int *iters, double *resid)
{
  if (!c) return RBL_ERR_ARG;
  if (iters) *iters = c->lanczos_iters;
  if (resid) *resid = c->lanczos_resid;
  return RBL_OK;
}

}  // extern "C"

// symmetric tridiagonal eigen-decomposition (implicit QL), m <= a few hundred.
// d[0..m) diagonal, e[0..m-1) off-diagonal; on return d = eigenvalues, Z (m x m,
// row-major) has eigenvectors in its COLUMNS.  Returns false if it fails to converge.
static bool tridiag_ql(std::vector<double> &d, std::vector<double> &e_in, std::vector<double> &Z, int m)
{
  std::vector<double> e(m, 0.0);
  for (int i = 0; i + 1 < m; ++i) e[i] = e_in[i];
  Z.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) Z[(size_t)i * m + i] = 1.0;
  for (int l = 0; l < m; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < m - 1; ++mm) {
        const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
        if (std::fabs(e[mm]) <= 2.3e-16 * dd) break;
      }
      if (mm != l) {
        if (iter++ == 200) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, cth = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i], b = cth * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
          s = f / r; cth = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * cth * b;
          p = s * r;
          d[i + 1] = g + p;
          g = cth * r - b;
          for (int k = 0; k < m; ++k) {
            f = Z[(size_t)k * m + i + 1];
            Z[(size_t)k * m + i + 1] = s * Z[(size_t)k * m + i] + cth * f;
            Z[(size_t)k * m + i] = cth * Z[(size_t)k * m + i] - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[mm] = 0.0;
      }
    } while (mm != l);
  }
  return true;
}

// ---------------------------------------------------------------------------
// M^{1/2} W on the device.  A = B Mob B with B ALWAYS applied (:668) and the wall
// term in Mob per wall_PC.
//   CHOLESKY: dense build of B Mob B -> in-place lower Cholesky -> L W  (reference)
//   LANCZOS : Krylov approximation of the symmetric square root with the
//             matrix-free matvec (no O(n^2) memory).
// ---------------------------------------------------------------------------
static int sync_bodies(rbl_ctx *c);
static int pc_block_factors(rbl_ctx *c, int b0 = 0, int b1 = -1);

// op(L_b) applied to bodies [b0, b0 + nbo) of nv vectors `pitch` doubles apart (in / out: the FULL vectors, body 0 first);
// mode 0: (L L^T)^-1, 1: L^-1, 2: L^-T.  Small bodies go through their explicit inverses (two matrix-vector products
// instead of two chains of substitution steps), the others through the substitution kernel.  In place is fine.
// Free space (no wall term in M): every body's mobility is the SAME body-frame matrix seen through the body's rotation,
// M_b = (I x R_b) M_body (I x R_b)^T (the RPY block of a pair depends on the separation vector only, which rotates with the
// body).  So there is nothing to factor per configuration: M_body = L L^T once per rbl_set_parameters, and the factor used
// for body b is G_b = (I x R_b) L  (G G^T = M_b; not triangular, which nothing here needs):
//   (G G^T)^-1 v = R (L L^T)^-1 R^T v,   G^-1 v = L^-1 R^T v,   G^-T v = R L^-T v,   G x = R L x.
// One matrix for all bodies also means the factor is read from cache instead of HBM.
static bool bf_on(const rbl_ctx *c) { return c->blk_bodyframe && (!c->S.wall || c->bf_wall_approx); }

static int bf_build(rbl_ctx *c)
{
  if (c->bf_valid) return RBL_OK;
  RblPhase ph(c, RBL_T_FACTOR);
  const RblBodyState &S = c->S;
  const int64_t m = 3 * (int64_t)S.N_blb, msz = m * m;
  int rc = ensure_xq_dev(c); if (rc) return rc;         // d_cfg: the blob positions in the body frame
  if ((rc = rbl_dev_reserve(c, c->d_bfL, sizeof(double) * (size_t)msz))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_bfLinv, rbl_cholesky_batched_work_bytes(m, 1)))) return rc;
  double *Lb = (double *)c->d_bfL.p;
  rbl_launch_build_M_batched(c->stream, rbl_make_params(S.a, S.eta), false, (const double *)c->d_cfg.p, S.N_blb, 1, Lb, msz, c->d_err);
  if ((rc = rbl_launch_cholesky_batched(c->stream, Lb, m, 1, msz, c->d_err, (double *)c->d_bfLinv.p)))
    return rbl_fail(c, rc, "body-frame cholesky launch failed");
  c->bf_inv = false; c->bf_tables = false;
  if (c->blk_explicit && m > 512 && c->blk_large != 0) {   // large bodies: ONE explicit inverse for all bodies and all time
    int chunk = 1;
    if ((rc = rbl_dev_reserve(c, c->d_bfX, rbl_block_inverse_bytes(m, 1)))) return rc;
    if ((rc = rbl_dev_reserve(c, c->d_blkAug, rbl_block_inverse_large_aug_bytes(m, 1, &chunk)))) return rc;
    if ((rc = rbl_launch_block_inverse_large(c->stream, Lb, m, 1, msz, (const double *)c->d_bfLinv.p, (double *)c->d_bfX.p, nullptr,
                                             (double *)c->d_blkAug.p)))
      return rbl_fail(c, rc, "body-frame inverse (large body) launch failed");
    c->bf_inv = true;
  }
  if (c->blk_explicit && m <= 512) {      // (every size the inversion kernel takes: the one-launch preconditioner pays at any of them)
    if ((rc = rbl_dev_reserve(c, c->d_bfX, rbl_block_inverse_bytes(m, 1)))) return rc;
    if ((rc = rbl_launch_block_inverse(c->stream, Lb, m, 1, msz, (const double *)c->d_bfLinv.p, (double *)c->d_bfX.p)))
      return rbl_fail(c, rc, "body-frame inverse launch failed");
    c->bf_inv = true;
    // tables of the whole block preconditioner in the body frame: M_body^-1, M_body^-1 K_body, chol(K_body^T M_body^-1 K_body)
    if ((rc = rbl_dev_reserve(c, c->d_bfPC, sizeof(double) * ((size_t)msz + 6 * (size_t)m + 36)))) return rc;
    double *Minv = (double *)c->d_bfPC.p;
    rbl_launch_bf_tables(c->stream, (const double *)c->d_bfX.p + (size_t)msz, (const double *)c->d_cfg.p, m, Minv, Minv + (size_t)msz,
                         Minv + (size_t)msz + 6 * (size_t)m, c->d_err);
    c->bf_tables = true;
  }
  c->bf_valid = true;
  c->tl_valid = false;
  return RBL_OK;
}

// the factors the block operations below work with: body-frame (free space) or per-configuration Cholesky (wall)
static int blk_prepare(rbl_ctx *c, int b0, int b1)
{
  const int hi = b1 < 0 ? c->S.N_bod : b1;
  if (b0 < 0 || b0 >= hi || hi > c->S.N_bod) return rbl_fail(c, RBL_ERR_ARG, "block factors: need 0 <= body_begin < body_end <= N_bodies");
  if (bf_on(c)) return bf_build(c);
  return pc_block_factors(c, b0, b1);
}

static int blk_solve(rbl_ctx *c, int b0, int nbo, const double *in, double *out, int nv, int64_t pitch, int mode, bool allow_f32 = true)
{
  if (nbo <= 0) return RBL_OK;
  RblPhase ph(c, RBL_T_PERBODY);
  const int64_t m = 3 * (int64_t)c->S.N_blb, msz = m * m;
  const size_t off = (size_t)b0 * (size_t)m;
  if (bf_on(c)) {
    const size_t tmpn = (size_t)m * (size_t)c->S.N_bod;
    int rc = ensure_xq_dev(c); if (rc) return rc;       // the rotations read the quaternions on the device: current ones (M_RFD displaces them)
    if ((rc = rbl_dev_reserve(c, c->d_blkTmp, sizeof(double) * 3 * tmpn))) return rc;
    double *tmp = (double *)c->d_blkTmp.p;
    const double *dQ = (const double *)c->d_XQ.p + 3 * (size_t)c->S.N_bod + 4 * (size_t)b0;
    for (int v0 = 0; v0 < nv; v0 += 3) {
      const int g = nv - v0 >= 3 ? 3 : nv - v0;
      const double *pi = in + (size_t)v0 * (size_t)pitch + off;
      double *po = out + (size_t)v0 * (size_t)pitch + off;
      if (c->bf_inv) {                                  // small bodies: X = L^-1 explicit, rotations fused into the products
        if (mode == 0) rc = rbl_launch_block_inv_apply(c->stream, (const double *)c->d_bfX.p, m, nbo, pi, po, m, g, pitch, 0, tmp + off, dQ);
        else if (pi != po) rc = rbl_launch_block_inv_apply(c->stream, (const double *)c->d_bfX.p, m, nbo, pi, po, m, g, pitch, mode, nullptr, dQ);
        else {
          rc = rbl_launch_block_inv_apply(c->stream, (const double *)c->d_bfX.p, m, nbo, pi, tmp + off, m, g, pitch, mode, nullptr, dQ);
          for (int v = 0; v < g && !rc; ++v)
            RBL_HIP(c, hipMemcpyAsync(po + (size_t)v * (size_t)pitch, tmp + off + (size_t)v * (size_t)pitch,
                                      sizeof(double) * (size_t)m * (size_t)nbo, hipMemcpyDeviceToDevice, c->stream));
        }
      } else if (m <= 512) {                            // short chains: substitution through the ONE shared factor, rotations fused
        rc = rbl_launch_block_solve_multi(c->stream, (const double *)c->d_bfL.p, m, nbo, 0, (const double *)c->d_bfLinv.p, pi, po, m, g,
                                          pitch, mode | 0x100, dQ);
      } else {                                          // large bodies: rotate, substitute (batch stride 0), rotate back
        const double *src = pi;
        if (mode != 2) {                                // R^T first (scratch laid out like the vectors)
          rbl_launch_rotate_bodies(c->stream, dQ, pi, tmp + off, c->S.N_blb, nbo, g, pitch, 1);
          src = tmp + off;
        }
        double *dst = (mode == 1) ? po : tmp + off;
        rc = rbl_launch_block_solve_multi(c->stream, (const double *)c->d_bfL.p, m, nbo, 0, (const double *)c->d_bfLinv.p, src, dst, m, g,
                                          pitch, mode | 0x100);
        if (!rc && mode != 1) rbl_launch_rotate_bodies(c->stream, dQ, dst, po, c->S.N_blb, nbo, g, pitch, 0);
      }
      if (rc) return rc;
    }
    return RBL_OK;
  }
  if (c->blk_inv_valid) {
    const size_t tmpn = (size_t)m * (size_t)c->S.N_bod;
    int rc = rbl_dev_reserve(c, c->d_blkTmp, sizeof(double) * 3 * tmpn); if (rc) return rc;
    // the single-precision copy (large bodies, rbl_set_tuning 84) serves whoever tolerates a factor that is exact to 6e-8 only
    const int f32 = (c->blk_f32_valid && allow_f32) ? 1 : 0;
    const size_t xsz = 2 * (size_t)(rbl_block_inverse_ld(m) * m);              // entries of one body's two layouts
    const double *X = f32 ? (const double *)((const float *)c->d_blkXf.p + (size_t)b0 * xsz)
                          : (const double *)c->d_blkX.p + (size_t)b0 * xsz;
    double *tmp = (double *)c->d_blkTmp.p;
    for (int v0 = 0; v0 < nv; v0 += 3) {              // groups of three vectors share the scratch
      const int g = nv - v0 >= 3 ? 3 : nv - v0;
      const double *pi = in + (size_t)v0 * (size_t)pitch + off;
      double *po = out + (size_t)v0 * (size_t)pitch + off;
      if (mode == 0) rc = rbl_launch_block_inv_apply(c->stream, X, m, nbo, pi, po, m, g, pitch, 0, tmp + off, nullptr, f32);
      else if (pi != po) rc = rbl_launch_block_inv_apply(c->stream, X, m, nbo, pi, po, m, g, pitch, mode, nullptr, nullptr, f32);
      else {                                          // in place: through the scratch
        rc = rbl_launch_block_inv_apply(c->stream, X, m, nbo, pi, tmp + off, m, g, pitch, mode, nullptr, nullptr, f32);
        for (int v = 0; v < g && !rc; ++v)
          RBL_HIP(c, hipMemcpyAsync(po + (size_t)v * (size_t)pitch, tmp + off + (size_t)v * (size_t)pitch,
                                    sizeof(double) * (size_t)m * (size_t)nbo, hipMemcpyDeviceToDevice, c->stream));
      }
      if (rc) return rc;
    }
    return RBL_OK;
  }
  const size_t lstride = rbl_cholesky_batched_work_bytes(m, 1) / sizeof(double);
  return rbl_launch_block_solve_multi(c->stream, (const double *)c->d_blkL.p + (size_t)b0 * (size_t)msz, m, nbo, msz,
                                      (const double *)c->d_blkLinv.p + (size_t)b0 * lstride, in + off, out + off, m, nv, pitch, mode);
}

// out = G_b in for bodies [b0, b0 + nbo) of ONE vector (in / out: the full vectors; not in place)
static int blk_trmv(rbl_ctx *c, int b0, int nbo, const double *in, double *out)
{
  if (nbo <= 0) return RBL_OK;
  RblPhase ph(c, RBL_T_PERBODY);
  const int64_t m = 3 * (int64_t)c->S.N_blb;
  const size_t off = (size_t)b0 * (size_t)m;
  if (bf_on(c)) { const int rc = ensure_xq_dev(c); if (rc) return rc; }
  const double *dQ0 = bf_on(c) ? (const double *)c->d_XQ.p + 3 * (size_t)c->S.N_bod + 4 * (size_t)b0 : nullptr;
  if (m <= 7936)                                        // every row independent, rotation fused (the vector of a body in 64 KB of LDS)
    return bf_on(c) ? rbl_launch_block_trmv_small(c->stream, (const double *)c->d_bfL.p, m, nbo, 0, in + off, out + off, m, dQ0)
                    : rbl_launch_block_trmv_small(c->stream, (const double *)c->d_blkL.p + (size_t)b0 * (size_t)(m * m), m, nbo, m * m,
                                                  in + off, out + off, m, nullptr);
  if (bf_on(c)) {                                       // G x = R (L x)
    int rc = rbl_launch_block_trmv(c->stream, (const double *)c->d_bfL.p, m, nbo, 0, in + off, out + off, m);
    if (rc) return rc;
    const double *dQ = (const double *)c->d_XQ.p + 3 * (size_t)c->S.N_bod + 4 * (size_t)b0;
    rbl_launch_rotate_bodies(c->stream, dQ, out + off, out + off, c->S.N_blb, nbo, 1, 0, 0);
    return RBL_OK;
  }
  return rbl_launch_block_trmv(c->stream, (const double *)c->d_blkL.p + (size_t)b0 * (size_t)(m * m), m, nbo, m * m, in + off,
                               out + off, m);
}

// ---- two-level factor of the preconditioned Lanczos root (round 3) --------------------------------------------------
// G = B L H with H = I + Q (L_E - I) Q^T: the block-Jacobi factor L times a low-rank correction that carries the monopole
// far field between the bodies (rbl_body_dev.hip: k_tl_orth for the algebra).  Any invertible G keeps
// x = G (G^-1 M G^-T)^{1/2} W an exact root; this one moves the collective translations of the bodies -- the modes whose
// Euclidean-norm error converges last under block-Jacobi -- into the factor: 27 bodies of shell_N_162 above a wall need
// 3-4 Lanczos iterations to 1e-3 instead of 6-7, 8-10 instead of 14-17 to 1e-6 (tests/experiments/two_level_root.py).
// Built once per configuration: Z = L^-1 K_t (three vectors through the per-body factors), a 3 N_bod-square sphere tensor,
// its Cholesky factor and explicit inverse -- all replicated on every rank of a multi-GPU context (Z by own bodies + sum).
// Not usable (tl_ok = false: plain block-Jacobi) when I + E is not positive definite or the small system does not fit.
static int tl_build(rbl_ctx *c)
{
  if (c->tl_valid) return RBL_OK;
  c->tl_valid = true; c->tl_ok = false;
  const RblBodyState &S = c->S;
  const int Nb = S.N_bod;
  const int64_t nt = 3 * (int64_t)Nb, n3 = 3 * (int64_t)Nb * S.N_blb;
  if (!c->tl_on || Nb < 2 || sizeof(double) * ((size_t)nt + 256) > 65536) return RBL_OK;
  RblPhase ph(c, RBL_T_FACTOR);
  int rc;
  if (!c->d_err2) {
    RBL_HIP(c, hipMalloc((void **)&c->d_err2, sizeof(unsigned)));
    RBL_HIP(c, hipMemsetAsync(c->d_err2, 0, sizeof(unsigned), c->stream));
  }
  if ((rc = rbl_dev_reserve(c, c->d_tlQ, sizeof(double) * 3 * (size_t)n3))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tlCb, sizeof(double) * 9 * (size_t)Nb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tlCs, sizeof(double) * (size_t)(nt * nt)))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tlA, sizeof(double) * (size_t)(nt * nt)))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tlLinv, rbl_cholesky_batched_work_bytes(nt, 1)))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tlX, rbl_block_inverse_bytes(nt, 1)))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tlT, sizeof(double) * 2 * 3 * (size_t)nt))) return rc;
  double *Q = (double *)c->d_tlQ.p;
  rbl_launch_tl_unit(c->stream, n3, Q);                                  // K_t: one vector per direction holds that column of every body
  if (comm_on(c)) {
    int b0, b1; comm_body_range(c, &b0, &b1);
    if ((rc = rbl_dev_reserve(c, c->d_tlZ, sizeof(double) * 3 * (size_t)n3))) return rc;
    double *t = (double *)c->d_tlZ.p;
    RBL_HIP(c, hipMemsetAsync(t, 0, sizeof(double) * 3 * (size_t)n3, c->stream));
    if ((rc = blk_solve(c, b0, b1 - b0, Q, t, 3, n3, 1, false))) return rc;
    if ((rc = comm_allreduce(c, t, 3 * n3))) return rc;
    RBL_HIP(c, hipMemcpyAsync(Q, t, sizeof(double) * 3 * (size_t)n3, hipMemcpyDeviceToDevice, c->stream));
  } else if ((rc = blk_solve(c, 0, Nb, Q, Q, 3, n3, 1, false))) return rc;   // Z = L^-1 K_t (G^-1 K_t with body-frame factors)
  rbl_launch_tl_orth(c->stream, Q, n3, S.N_blb, Nb, (double *)c->d_tlCb.p, c->d_err2);
  // far-field model: the pair tensor of spheres of the bodies' outer radius at the body centres; the wall term only when no
  // sphere reaches the wall (any SPD model keeps the root exact -- it only has to resemble the true coupling)
  double zmin = 1.0e300;
  for (int b = 0; b < Nb; ++b) zmin = std::min(zmin, S.X[3 * (size_t)b + 2]);
  const bool wall_s = S.wall && zmin > 1.1 * c->body_radius;
  if ((rc = ensure_xq_dev(c))) return rc;
  rbl_launch_build_M(c->stream, rbl_make_params(c->body_radius, S.eta), wall_s, false, (const double *)c->d_XQ.p, Nb, (double *)c->d_tlCs.p,
                     c->d_err2);
  rbl_launch_tl_E(c->stream, (const double *)c->d_tlCs.p, (const double *)c->d_tlCb.p, Nb, (double *)c->d_tlA.p);
  if ((rc = rbl_launch_cholesky_batched(c->stream, (double *)c->d_tlA.p, nt, 1, nt * nt, c->d_err2, (double *)c->d_tlLinv.p)))
    return rbl_fail(c, rc, "two-level factor: cholesky launch failed");
  if (nt <= 512) rc = rbl_launch_block_inverse(c->stream, (const double *)c->d_tlA.p, nt, 1, nt * nt, (const double *)c->d_tlLinv.p, (double *)c->d_tlX.p);
  else {
    int chunk = 1;
    if ((rc = rbl_dev_reserve(c, c->d_blkAug, rbl_block_inverse_large_aug_bytes(nt, 1, &chunk)))) return rc;
    rc = rbl_launch_block_inverse_large(c->stream, (const double *)c->d_tlA.p, nt, 1, nt * nt, (const double *)c->d_tlLinv.p, (double *)c->d_tlX.p,
                                        nullptr, (double *)c->d_blkAug.p);
  }
  if (rc) return rbl_fail(c, rc, "two-level factor: inverse launch failed");
  RBL_HIP(c, hipMemcpyAsync(c->h_err, c->d_err2, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  RBL_HIP(c, hipMemsetAsync(c->d_err2, 0, sizeof(unsigned), c->stream));
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  c->tl_ok = (*c->h_err == 0);                                           // (not SPD / sphere below the wall: block-Jacobi alone)
  return RBL_OK;
}

// wo_v = Op w_v for nvec vectors `pitch` apart (wo == w allowed);  op 0: H = I + Q (L_E - I) Q^T,  1: H^-1,  2: H^-T
static int tl_apply(rbl_ctx *c, const double *w, double *wo, int nvec, int64_t pitch, int op)
{
  const RblBodyState &S = c->S;
  const int Nb = S.N_bod;
  const int64_t nt = 3 * (int64_t)Nb, n3 = 3 * (int64_t)Nb * S.N_blb;
  const double *Q = (const double *)c->d_tlQ.p;
  double *t = (double *)c->d_tlT.p, *sv = t + 3 * nt;                     // (up to three vectors at a time)
  for (int v0 = 0; v0 < nvec; v0 += 3) {
    const int g = nvec - v0 >= 3 ? 3 : nvec - v0;
    const double *wv = w + (size_t)v0 * (size_t)pitch;
    rbl_launch_tl_qt(c->stream, Q, n3, S.N_blb, Nb, wv, pitch, g, t, nt);
    int rc = RBL_OK;
    if (op != 0) rc = rbl_launch_block_inv_apply(c->stream, (const double *)c->d_tlX.p, nt, 1, t, sv, nt, g, nt, op, nullptr);   // all g vectors in one pass
    else
      for (int v = 0; v < g && !rc; ++v)
        rc = rbl_launch_block_trmv_small(c->stream, (const double *)c->d_tlA.p, nt, 1, 0, t + (size_t)v * nt, sv + (size_t)v * nt, 0, nullptr);
    if (rc) return rbl_fail(c, rc, "two-level factor: application failed");
    rbl_launch_tl_addq(c->stream, Q, n3, S.N_blb, sv, t, nt, wv, wo + (size_t)v0 * (size_t)pitch, pitch, g);
  }
  return RBL_OK;
}

// y_v = (B M B) x_v for nvec (1 or 2) vectors stored back to back; two vectors share the pair coefficients
// precond: y_v = L^-1 M L^-T x_v with the per-body Cholesky factors L L^T = M_body (block Jacobi), M undamped
static int apply_A_dev(rbl_ctx *c, const RblParams &P, const double *d_r, int64_t nbl,
                       const double *d_x, double *d_y, double *d_tmp, int nvec = 1, bool precond = false)
{
  const int64_t n = 3 * nbl;
  if (precond) {
    int rc;
    // B G (G^-1 M G^-T)^{1/2} W is an exact root for any invertible G applied CONSISTENTLY; the single-precision copy of L^-1
    // and the fp64 L of the final product agree to 6e-8 only, so it serves the loose tolerances (>= 1e-5) and no others
    const bool lz32 = c->lanczos_tol >= 1.0e-5;
    const bool tl = c->tl_ok;                          // two-level factor: G^-1 = H^-1 L^-1, G^-T = L^-T H^-T
    const size_t vbytes = sizeof(double) * (size_t)nvec * (size_t)n;
    if (comm_on(c)) {   // every rank substitutes through ITS bodies' factors only; sums complete the vectors
      int b0, b1; comm_body_range(c, &b0, &b1);
      const double *src = d_x;
      if (tl) {                                        // (d_y is free until the product: H^-T x goes there)
        if ((rc = tl_apply(c, d_x, d_y, nvec, n, 2))) return rc;
        src = d_y;
      }
      RBL_HIP(c, hipMemsetAsync(d_tmp, 0, vbytes, c->stream));
      if ((rc = blk_solve(c, b0, b1 - b0, src, d_tmp, nvec, n, 2, lz32)))
        return rbl_fail(c, rc, "preconditioned square root: bodies with more than 2730 blobs are not supported");
      if ((rc = comm_allreduce(c, d_tmp, (int64_t)nvec * n))) return rc;
      c->no_damp = true;
      rc = apply_M_multi_enqueue(c, c->S.wall, d_tmp, d_r, nbl, nvec, d_y);
      c->no_damp = false;
      if (rc) return rc;
      RBL_HIP(c, hipMemsetAsync(d_tmp, 0, vbytes, c->stream));
      if ((rc = blk_solve(c, b0, b1 - b0, d_y, d_tmp, nvec, n, 1, lz32))) return rc;
      if ((rc = comm_allreduce(c, d_tmp, (int64_t)nvec * n))) return rc;
      RBL_HIP(c, hipMemcpyAsync(d_y, d_tmp, vbytes, hipMemcpyDeviceToDevice, c->stream));
      return tl ? tl_apply(c, d_y, d_y, nvec, n, 1) : RBL_OK;
    }
    if (tl) {                                          // H^-T x staged in d_y (free until the product), then out of place through L^-T
      if ((rc = tl_apply(c, d_x, d_y, nvec, n, 2))) return rc;
      rc = blk_solve(c, 0, c->S.N_bod, d_y, d_tmp, nvec, n, 2, lz32);
    } else rc = blk_solve(c, 0, c->S.N_bod, d_x, d_tmp, nvec, n, 2, lz32);   // both vectors in one pass over L
    if (rc) return rbl_fail(c, rc, "preconditioned square root: bodies with more than 2730 blobs are not supported");
    double *prod = d_y;                                // explicit inverses do not work in place: product into their scratch
    if (c->blk_inv_valid || (bf_on(c) && c->bf_inv)) {
      if ((rc = rbl_dev_reserve(c, c->d_blkTmp, sizeof(double) * 3 * (size_t)n))) return rc;
      prod = (double *)c->d_blkTmp.p;
    }
    c->no_damp = true;
    rc = apply_M_multi_enqueue(c, c->S.wall, d_tmp, d_r, nbl, nvec, prod);
    c->no_damp = false;
    if (rc) return rc;
    if ((rc = blk_solve(c, 0, c->S.N_bod, prod, d_y, nvec, n, 1, lz32))) return rc;
    return tl ? tl_apply(c, d_y, d_y, nvec, n, 1) : RBL_OK;
  }
  if (c->S.wall) return apply_M_multi_enqueue(c, true, d_x, d_r, nbl, nvec, d_y);   // kernel applies B M B itself
  for (int v = 0; v < nvec; ++v)                                                     // free-space M, damping around it
    rbl_launch_scale_by_damp(c->stream, P, d_r, nbl, d_x + (size_t)v * n, d_tmp + (size_t)v * n);
  int rc = apply_M_multi_enqueue(c, false, d_tmp, d_r, nbl, nvec, d_y);
  for (int v = 0; v < nvec; ++v)
    rbl_launch_scale_by_damp(c->stream, P, d_r, nbl, d_y + (size_t)v * n, d_y + (size_t)v * n);
  return rc;
}

// coefficients of  |W| T_m^{1/2} e_1  in the Krylov basis (T_m = tridiag(alpha, beta))
static int lanczos_coeffs(rbl_ctx *c, const std::vector<double> &alpha, const std::vector<double> &beta, int m,
                          double wnorm, std::vector<double> &y)
{
  std::vector<double> d(alpha.begin(), alpha.begin() + m), e(beta.begin(), beta.begin() + (m - 1)), Z;
  if (!tridiag_ql(d, e, Z, m)) return rbl_fail(c, RBL_ERR_NONFINITE, "Lanczos: tridiagonal eigensolve failed");
  y.assign(m, 0.0);
  double dmax = 0.0;
  for (int k = 0; k < m; ++k) dmax = std::max(dmax, std::fabs(d[k]));
  for (int k = 0; k < m; ++k) {
    if (d[k] < 0.0) {
      if (d[k] < -1e-10 * dmax) return rbl_fail(c, RBL_ERR_NOT_SPD, "Lanczos: operator is not positive semi-definite");
      d[k] = 0.0;
    }
    const double sk = std::sqrt(d[k]) * Z[k];  // Z[0*m + k] = first component of eigvec k
    for (int p = 0; p < m; ++p) y[p] += Z[(size_t)p * m + k] * sk;
  }
  for (int p = 0; p < m; ++p) y[p] *= wnorm;
  return RBL_OK;
}

// The recurrence runs entirely on the device (alpha, beta stay there); the host reads them back only to test
// convergence: every iteration when a product is expensive, every 4th when the iteration is launch-bound (small
// systems), so the stream is not drained twice per iteration.
// Round 3: every new vector is re-orthogonalised against the WHOLE basis (classical Gram-Schmidt twice,
// rbl_launch_lanczos_step_reorth).  The plain three-term recurrence loses orthogonality as soon as a Ritz value has
// converged; the square-root estimate then stagnates (cfg 2, tolerance 1e-9: 300 iterations, true error 5e-6) while the
// "relative change" of the coefficient vector keeps shrinking.  The basis is stored for the final combination anyway.
// Stopping estimate: d_m = |x_m - x_{m-1}| / |x_m| is the size of the LAST correction, not of the error; with the
// corrections shrinking by rho = d_m / d_{m-1} per iteration the error of x_m is the tail of a geometric series,
// d_m rho / (1 - rho) -- that is what is compared with the tolerance and reported (rbl_get_lanczos_report).
static int mhalf_lanczos_dev(rbl_ctx *c, const double *d_r, int64_t nbl, const double *d_W, double *d_out,
                             int nvec = 1, bool precond = false)
{
  // precond (RBL_MHALF_LANCZOS_PC): Lanczos on S = L^-1 M L^-T (eigenvalues clustered around 1: a handful of
  // iterations), then  x = B L S^{1/2} W, whose covariance is B L S L^T B = B M B exactly -- another valid
  // square root of the same matrix (Chow & Saad's preconditioned sampling).
  // nvec = 1 or 2 independent recurrences advanced in lock step: with two (the Brownian step's W1, W2) every
  // iteration is ONE two-vector product whose pair coefficients are shared.
  const int64_t n = 3 * nbl;
  const bool reorth = c->lanczos_reorth;
  const int maxit = c->lanczos_max_iter;
  const RblParams P = rbl_make_params(c->S.a, c->S.eta);
  int rc;
  // workspace: V[(maxit+1)][nvec][n] | u[nvec][n], tmp[nvec][n] | per vector: alpha[maxit], beta[maxit], |W|, coef[maxit] |
  //            Gram-Schmidt column scratch | partial sums
  const size_t vbytes = sizeof(double) * (size_t)n;
  const size_t nsc = (size_t)4 * maxit + 1;                            // alpha, beta, |W|, coef (two sets: estimate, correction)
  const size_t nh = (size_t)maxit + 2;
  const bool out_norm = precond && c->lanczos_out_norm;                // stopping estimate in the norm of the increment itself
  const size_t ndot = 2 + 2 * 512;                                     // rbl_launch_dot2 scratch
  const size_t npart = reorth ? (size_t)nvec * rbl_gmres_part_doubles() + rbl_lanczos_part_doubles() : rbl_lanczos_part_doubles();
  if ((rc = rbl_dev_reserve(c, c->d_tmp, vbytes * (size_t)(maxit + 1) * nvec))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tmp2, vbytes * (out_norm ? 4 : 2) * nvec + sizeof(double) * (nsc * nvec + nh * nvec + npart + ndot)))) return rc;
  double *V = (double *)c->d_tmp.p;
  double *u = (double *)c->d_tmp2.p, *tmp = u + (size_t)nvec * n, *ex = tmp + (size_t)nvec * n;   // ex: 2 nvec vectors (out_norm only)
  double *sc = ex + (out_norm ? (size_t)2 * nvec * n : 0);
  double *d_hcol = sc + nsc * nvec, *d_part = d_hcol + nh * nvec, *d_dot = d_part + npart;
  double *d_part_init = reorth ? d_part + (size_t)nvec * rbl_gmres_part_doubles() : d_part;
  auto d_alpha = [&](int v) { return sc + nsc * v; };
  auto d_beta = [&](int v) { return sc + nsc * v + maxit; };
  auto d_wn = [&](int v) { return sc + nsc * v + 2 * maxit; };
  auto d_coef = [&](int v) { return sc + nsc * v + 2 * maxit + 1; };
  auto Vp = [&](int it, int v) { return V + ((size_t)it * nvec + v) * n; };
  for (int v = 0; v < nvec; ++v) rbl_launch_lanczos_init(c->stream, n, d_W + (size_t)v * n, d_wn(v), Vp(0, v), d_part_init);
  const int check_every = (nbl > 20000) ? 1 : 4;
  std::vector<double> hs(nsc * nvec), alpha, beta, y_prev, y_pp;
  std::vector<std::vector<double>> y_cur(nvec);
  std::vector<double> wnorm(nvec, 0.0), resid(nvec, 1.0), d_last(nvec, -1.0);
  std::vector<int> m_last(nvec, -1);
  auto change = [](const std::vector<double> &ya, const std::vector<double> &yb) {   // |ya - [yb; 0]| / |ya|
    double dn = 0.0, yn = 0.0;
    for (size_t p = 0; p < ya.size(); ++p) {
      const double yp = p < yb.size() ? yb[p] : 0.0;
      dn += (ya[p] - yp) * (ya[p] - yp);
      yn += ya[p] * ya[p];
    }
    return yn > 0.0 ? std::sqrt(dn / yn) : 0.0;
  };
  int m = 0, next_check = check_every;
  bool done = false;
  static const bool trace = std::getenv("RBL_LANCZOS_TRACE") != nullptr;      // diagnostic: the estimate's history on stderr
  for (int it = 0; it < maxit && !done; ++it) {
    // inexact Krylov: an estimate wanted to lanczos_tol >= 1e-4 does not notice a product error of ~1e-6
    c->sym_tune.relaxed = (c->gmres_relax && c->lanczos_tol >= 1.0e-4) ? 1 : 0;
    rc = apply_A_dev(c, P, d_r, nbl, Vp(it, 0), u, tmp, nvec, precond);
    c->sym_tune.relaxed = 0;
    if (rc) return rc;
    // both recurrences of a pair in the same launches (vectors n apart, their scalars nsc apart)
    if (reorth)
      rbl_launch_lanczos_step_reorth(c->stream, n, it + 1, u, V, Vp(it + 1, 0), d_alpha(0) + it, d_beta(0) + it, (int64_t)nsc, d_hcol,
                                     (int64_t)nh, d_part, nvec);
    else
      rbl_launch_lanczos_step(c->stream, n, u, Vp(it, 0), it > 0 ? Vp(it - 1, 0) : nullptr, it > 0 ? d_beta(0) + (it - 1) : nullptr,
                              d_alpha(0) + it, d_beta(0) + it, Vp(it + 1, 0), d_part, nvec, n, (int64_t)nsc);
    m = it + 1;
    // (the host test is an O(m^3) eigen-solve: tests thin out as the basis grows -- every iteration up to 16, then every m/16-th)
    if (m < next_check && m != maxit) continue;
    next_check = m + std::max(check_every, m / 16);
    if ((rc = read_back(c, hs.data(), sc, sizeof(double) * hs.size()))) return rc;
    bool all_conv = true;
    for (int v = 0; v < nvec; ++v) {
      const double *h = hs.data() + nsc * v;
      alpha.assign(h, h + m);
      beta.assign(h + maxit, h + maxit + m);
      wnorm[v] = h[2 * maxit];
      if (!(wnorm[v] > 0.0)) { y_cur[v].assign(m, 0.0); resid[v] = 0.0; continue; }   // zero noise vector -> zero increment
      for (int k = 0; k < m; ++k)
        if (!std::isfinite(alpha[k]) || !std::isfinite(beta[k])) return rbl_fail(c, RBL_ERR_NONFINITE, "Lanczos: non-finite recurrence");
      int mv = m;
      for (int k = 0; k < m - 1; ++k)                     // breakdown before the last step: Krylov space exhausted
        if (!(beta[k] > 1e-300)) { mv = k + 1; break; }
      if ((rc = lanczos_coeffs(c, alpha, beta, mv, wnorm[v], y_cur[v]))) return rc;
      if (mv > 1) {  // change of the coefficient vector == change of the estimate (V orthonormal), extrapolated to the error
        if ((rc = lanczos_coeffs(c, alpha, beta, mv - 1, wnorm[v], y_prev))) return rc;
        const double dm = change(y_cur[v], y_prev);
        double dm1 = (m_last[v] == mv - 1) ? d_last[v] : -1.0;      // the previous correction: kept from the last test ...
        if (dm1 < 0.0 && mv > 2) {                                   // ... or evaluated now (tests every 4th iteration)
          if ((rc = lanczos_coeffs(c, alpha, beta, mv - 2, wnorm[v], y_pp))) return rc;
          dm1 = change(y_prev, y_pp);
        }
        double rho = (dm1 > 0.0) ? dm / dm1 : 0.5;
        if (rho > 0.95) rho = 0.95;                                  // (not contracting yet: at least 19 x the last correction)
        resid[v] = dm * rho / (1.0 - rho);
        d_last[v] = dm; m_last[v] = mv;
        if (trace) std::fprintf(stderr, "rbl lanczos%s: vector %d  m = %d  change %.3e  rho %.3f  error estimate %.3e\n", precond ? " (pc)" : "", v, mv, dm, rho, resid[v]);
      }
      y_cur[v].resize(m, 0.0);                            // a recurrence that broke down early contributes no further vectors
      const bool conv = resid[v] < c->lanczos_tol || mv < m || !(beta[mv - 1] > 1e-300);
      all_conv = all_conv && conv;
    }
    if (all_conv && out_norm && m > 1) {
      // Preconditioned root: the recurrence lives in the variables z = S^{1/2} W, where the change of the coefficient vector
      // measures the error in the ENERGY norm of the increment x = B L z (x^T (B M B)^-1 x = z^T S^-1 z ~ |z|^2).  In the
      // Euclidean norm of x itself the factor L weighs the slowly converging collective modes ~sqrt(lambda_max / lambda_mean)
      // times heavier (cfg 3: the root identity |G s - B M v| / |B M v| came out at 1e-2 for an energy-norm estimate of
      // 3e-4).  So once the cheap estimate has passed, the last correction is evaluated where the caller sees it:
      // d = |B L V (y_m - y_{m-1})| / |B L V y_m|, extrapolated with the same rho; a few combinations and factor products
      // per test, only near convergence.
      int b0 = 0, b1 = c->S.N_bod;
      if (comm_on(c)) comm_body_range(c, &b0, &b1);
      std::vector<double> cf((size_t)2 * m);
      double worst = 0.0;
      for (int v = 0; v < nvec; ++v) {
        if (!(wnorm[v] > 0.0)) continue;
        const std::vector<double> &yc = y_cur[v];
        std::vector<double> yp;
        const int mv = m_last[v] > 1 ? m_last[v] : m;
        alpha.assign(hs.data() + nsc * v, hs.data() + nsc * v + m);
        beta.assign(hs.data() + nsc * v + maxit, hs.data() + nsc * v + maxit + m);
        if ((rc = lanczos_coeffs(c, alpha, beta, mv - 1, wnorm[v], yp))) return rc;
        for (int p_ = 0; p_ < m; ++p_) { cf[p_] = yc[p_]; cf[m + p_] = yc[p_] - (p_ < (int)yp.size() ? yp[p_] : 0.0); }
        RBL_HIP(c, hipMemcpyAsync(d_coef(v), cf.data(), sizeof(double) * 2 * (size_t)m, hipMemcpyHostToDevice, c->stream));
        RBL_HIP(c, hipStreamSynchronize(c->stream));                 // (pageable source; two tests per solve at most)
        double *zx = u + (size_t)v * n, *zd = tmp + (size_t)v * n, *ox = ex + (size_t)(2 * v) * n, *od = ox + n;
        rbl_launch_lanczos_combine(c->stream, n, Vp(0, v), d_coef(v), m, zx, (int64_t)nvec * n);
        rbl_launch_lanczos_combine(c->stream, n, Vp(0, v), d_coef(v) + m, m, zd, (int64_t)nvec * n);
        for (int w = 0; w < 2; ++w) {
          double *zin = w ? zd : zx, *o = w ? od : ox;
          if (c->tl_ok && (rc = tl_apply(c, zin, zin, 1, n, 0))) return rc;
          if (comm_on(c)) RBL_HIP(c, hipMemsetAsync(o, 0, sizeof(double) * (size_t)n, c->stream));
          if ((rc = blk_trmv(c, b0, b1 - b0, zin, o))) return rc;
          if (comm_on(c) && (rc = comm_allreduce(c, o, n))) return rc;
          rbl_launch_scale_by_damp(c->stream, P, d_r, nbl, o, o);
        }
        double h2[2] = {0.0, 0.0}, hx[2] = {0.0, 0.0};
        rbl_launch_dot2(c->stream, od, od, nullptr, n, d_dot);
        if ((rc = read_back(c, h2, d_dot, sizeof(double) * 2))) return rc;
        rbl_launch_dot2(c->stream, ox, ox, nullptr, n, d_dot);
        if ((rc = read_back(c, hx, d_dot, sizeof(double) * 2))) return rc;
        const double dout = hx[0] > 0.0 ? std::sqrt(h2[0] / hx[0]) : 0.0;
        // rho of the coefficient sequence (same contraction, other norm); resid[v] = d_m rho / (1 - rho) in the energy norm
        const double ratio = d_last[v] > 0.0 ? resid[v] / d_last[v] : 1.0;
        const double est = dout * ratio;
        if (trace) std::fprintf(stderr, "rbl lanczos (pc): vector %d  m = %d  energy-norm estimate %.3e  increment-norm change %.3e  estimate %.3e\n", v, m, resid[v], dout, est);
        resid[v] = est;
        worst = std::max(worst, est);
      }
      if (!(worst < c->lanczos_tol) && m < maxit) all_conv = false;
    }
    if (all_conv) done = true;
  }
  c->lanczos_iters = m;
  c->lanczos_resid = *std::max_element(resid.begin(), resid.end());
  // d_out_v = V_v[:, :m] y_v
  for (int v = 0; v < nvec; ++v) {
    if ((int)y_cur[v].size() < m) y_cur[v].resize(m, 0.0);
    if ((rc = upload_coef(c, d_coef(v), y_cur[v].data(), m, v))) return rc;
  }
  for (int v = 0; v < nvec; ++v)
    rbl_launch_lanczos_combine(c->stream, n, Vp(0, v), d_coef(v), m, d_out + (size_t)v * n, (int64_t)nvec * n);
  if (precond) {   // x = B (L y)
    int b0 = 0, b1 = c->S.N_bod;
    if (comm_on(c)) comm_body_range(c, &b0, &b1);
    for (int v = 0; v < nvec; ++v) {
      double *o = d_out + (size_t)v * n;
      if (c->tl_ok && (rc = tl_apply(c, o, o, 1, n, 0))) return rc;       // x = B L (H y)
      if (comm_on(c)) RBL_HIP(c, hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)n, c->stream));
      if ((rc = blk_trmv(c, b0, b1 - b0, o, tmp))) return rc;
      if (comm_on(c) && (rc = comm_allreduce(c, tmp, n))) return rc;
      rbl_launch_scale_by_damp(c->stream, P, d_r, nbl, tmp, o);
    }
  }
  return RBL_OK;
}

// nvec noise vectors (columns of d_W, stride n) -> nvec increments.  The dense factorisation is done ONCE
// for all of them (the reference calls M_half_W() once per vector, :927-936, and refactors each time).
static int mhalf_dev_multi(rbl_ctx *c, const double *d_r, int64_t nbl, const double *d_W, int nvec, int method,
                           double *d_out)
{
  const int64_t n = 3 * nbl;
  int rc;
  RblPhase ph_total(c, RBL_T_TOTAL);
  if (method == RBL_MHALF_LANCZOS || method == RBL_MHALF_LANCZOS_PC) {
    const bool pc = method == RBL_MHALF_LANCZOS_PC;
    if (pc) {   // block-Jacobi factors of the object's own configuration: d_r must be its own blob positions
      if (!c->S.cfg_set || nbl != (int64_t)c->S.N_bod * c->S.N_blb)
        return rbl_fail(c, RBL_ERR_SIZE, "M_half_W (preconditioned Lanczos) works on the object's own configuration only");
      if ((rc = sync_bodies(c))) return rc;
      int b0 = 0, b1 = -1;
      if (comm_on(c)) comm_body_range(c, &b0, &b1);
      if (b1 != b0 && (rc = blk_prepare(c, b0, b1))) return rc;
      if ((rc = tl_build(c))) return rc;
    }
    int v = 0;   // pairs of vectors in lock step (shared pair coefficients), a single one alone
    for (; v + 2 <= nvec; v += 2)
      if ((rc = mhalf_lanczos_dev(c, d_r, nbl, d_W + (size_t)v * n, d_out + (size_t)v * n, 2, pc))) return rc;
    for (; v < nvec; ++v)
      if ((rc = mhalf_lanczos_dev(c, d_r, nbl, d_W + (size_t)v * n, d_out + (size_t)v * n, 1, pc))) return rc;
    return RBL_OK;
  }
  if (method != RBL_MHALF_CHOLESKY) return rbl_fail(c, RBL_ERR_ARG, "M_half_W: unknown method");
  RblPhase ph_dense(c, RBL_T_DENSE);
  const size_t mb = sizeof(double) * (size_t)n * (size_t)n;
  if ((rc = rbl_dev_reserve(c, c->d_mat, mb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tmp, rbl_trmv_part_bytes(n)))) return rc;
  const RblParams P = rbl_make_params(c->S.a, c->S.eta);
  rbl_launch_build_M(c->stream, P, c->S.wall, true, d_r, nbl, (double *)c->d_mat.p, c->d_err);  // :667-669
  if ((rc = rbl_dev_reserve(c, c->d_chol, rbl_cholesky_work_bytes(n)))) return rc;
  rc = rbl_launch_cholesky(c->stream, (double *)c->d_mat.p, n, false, c->d_err, (double *)c->d_chol.p, c->d_chol.bytes, &c->chol_aux);   // :670-671
  if (rc) return rbl_fail(c, rc, "cholesky launch failed");
  for (int v = 0; v < nvec; ++v)
    rbl_launch_trmv_lower(c->stream, (const double *)c->d_mat.p, n, d_W + (size_t)v * n, d_out + (size_t)v * n,
                          (double *)c->d_tmp.p);  // :672
  return RBL_OK;
}

static int mhalf_dev(rbl_ctx *c, const double *d_r, int64_t nbl, const double *d_W, int method, double *d_out)
{
  return mhalf_dev_multi(c, d_r, nbl, d_W, 1, method, d_out);
}

extern "C" {

int rbl_M_half_W_dev(rbl_ctx *c, const double *d_r, int64_t n_blobs, const double *d_W, int method,
                     double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n_blobs <= 0) return rbl_fail(c, RBL_ERR_SIZE, "M_half_W_dev: n_blobs must be positive");
  return mhalf_dev(c, d_r, n_blobs, d_W, method, d_out);
}

int rbl_M_half_W_r(rbl_ctx *c, const double *r, int64_t n3, const double *W, uint64_t seed, int method,
                   double *out)
{
  int rc = need_params(c); if (rc) return rc;
  if (n3 <= 0 || n3 % 3 != 0) return rbl_fail(c, RBL_ERR_SIZE, "r_vecs must have length 3N");
  if ((rc = rbl_dev_init(c))) return rc;
  const size_t vb = sizeof(double) * (size_t)n3;
  if ((rc = rbl_dev_reserve(c, c->d_r, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_W, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_U, vb))) return rc;
  { int rc__ = copy_h2d(c, c->d_r.p, r, vb); if (rc__) return rc__; }
  if (W) { int rc__ = copy_h2d(c, c->d_W.p, W, vb); if (rc__) return rc__; }
  else rbl_launch_normal(c->stream, seed, 0, n3, (double *)c->d_W.p);  // replaces rand_vector :730-741
  if ((rc = mhalf_dev(c, (const double *)c->d_r.p, n3 / 3, (const double *)c->d_W.p, method, (double *)c->d_U.p))) return rc;
  { int rc__ = copy_d2h(c, out, c->d_U.p, vb); if (rc__) return rc__; }
  return finish_and_check(c);
}

int rbl_M_half_W(rbl_ctx *c, const double *W, uint64_t seed, int method, double *out)
{
  int rc = need_config(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  const int64_t n3 = (int64_t)3 * c->S.N_bod * c->S.N_blb;  // :663
  const size_t vb = sizeof(double) * (size_t)n3;
  if ((rc = rbl_dev_reserve(c, c->d_r, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_W, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_U, vb))) return rc;
  if ((rc = positions_dev(c, 0, c->S.N_bod, (double *)c->d_r.p))) return rc;  // multi_body_pos :662
  if (W) { int rc__ = copy_h2d(c, c->d_W.p, W, vb); if (rc__) return rc__; }
  else rbl_launch_normal(c->stream, seed, 0, n3, (double *)c->d_W.p);
  if ((rc = mhalf_dev(c, (const double *)c->d_r.p, n3 / 3, (const double *)c->d_W.p, method, (double *)c->d_U.p))) return rc;
  { int rc__ = copy_d2h(c, out, c->d_U.p, vb); if (rc__) return rc__; }
  return finish_and_check(c);
}

int rbl_debug_pair_blocks(rbl_ctx *c, const double *ri, const double *rj, const int32_t *ii,
                          const int32_t *jj, int64_t n, int wall, int mode, double *out9)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n <= 0) return RBL_OK;
  if ((rc = rbl_dev_reserve(c, c->d_tmp, sizeof(double) * 6 * (size_t)n + sizeof(int32_t) * 2 * (size_t)n))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tmp2, sizeof(double) * 9 * (size_t)n))) return rc;
  double *dri = (double *)c->d_tmp.p, *drj = dri + 3 * n;
  int32_t *dii = (int32_t *)(drj + 3 * n), *djj = dii + n;
  RBL_HIP(c, hipMemcpyAsync(dri, ri, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
  RBL_HIP(c, hipMemcpyAsync(drj, rj, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
  RBL_HIP(c, hipMemcpyAsync(dii, ii, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  RBL_HIP(c, hipMemcpyAsync(djj, jj, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  rbl_launch_pair_blocks(c->stream, rbl_make_params(c->S.a, c->S.eta), wall != 0, mode, dri, drj, dii,
                         djj, n, (double *)c->d_tmp2.p, c->d_err);
  RBL_HIP(c, hipMemcpyAsync(out9, c->d_tmp2.p, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, c->stream));
  return finish_and_check(c);
}

// ============================================================================
// 3. device-pointer API
// ============================================================================
int rbl_set_stream(rbl_ctx *c, void *s)
{
  if (!c) return RBL_ERR_ARG;
  int rc = rbl_dev_init(c); if (rc) return rc;
  c->stream = (hipStream_t)s;
  return RBL_OK;
}

int rbl_apply_M_dev(rbl_ctx *c, const double *d_F, const double *d_r, int64_t n_blobs,
                    int64_t row_begin, int64_t row_end, double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n_blobs <= 0 || row_begin < 0 || row_end > n_blobs || row_begin > row_end)
    return rbl_fail(c, RBL_ERR_SIZE, "apply_M_dev: row range out of bounds");
  return apply_M_enqueue(c, c->S.wall, d_F, d_r, n_blobs, row_begin, row_end, d_out);
}

int rbl_apply_M_multi_dev(rbl_ctx *c, const double *d_F, const double *d_r, int64_t n_blobs, int nrhs,
                          double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n_blobs <= 0 || nrhs < 1) return rbl_fail(c, RBL_ERR_SIZE, "apply_M_multi_dev: need n_blobs > 0, nrhs >= 1");
  return apply_M_multi_enqueue(c, c->S.wall, d_F, d_r, n_blobs, nrhs, d_out);
}

// ---- per-body (block-Jacobi) Cholesky factors of the object's own configuration, for callers that compose the
// preconditioned square root themselves (the multi-GPU driver): L L^T = M_body (wall term per wall_PC, undamped)
int rbl_block_solve_range_dev(rbl_ctx *c, const double *d_in, double *d_out, int mode, int body_begin, int body_end)
{
  if (!c) return RBL_ERR_ARG;
  int rc = sync_bodies(c); if (rc) return rc;
  if (mode < 0 || mode > 7 || mode == 4 || !d_in || !d_out)
    return rbl_fail(c, RBL_ERR_ARG, "block_solve_dev: mode 0 (L L^T)^-1, 1 L^-1, 2 L^-T, 3 L x; 5 G^-1, 6 G^-T, 7 G x (factor of the preconditioned root)");
  if (body_end < 0) body_end = c->S.N_bod;
  if ((rc = blk_prepare(c, body_begin, body_end))) return rc;
  const int nb = body_end - body_begin;
  if (mode >= 5) {        // the whole factor of the preconditioned Lanczos root, G = L H (two-level) or L: all bodies, not in place
    if (body_begin != 0 || body_end != c->S.N_bod || d_in == d_out)
      return rbl_fail(c, RBL_ERR_ARG, "block_solve_dev: modes 5-7 take all bodies and do not work in place");
    if (comm_on(c)) return rbl_fail(c, RBL_ERR_ARG, "block_solve_dev: modes 5-7 are single-GPU test hooks");
    if ((rc = tl_build(c))) return rc;
    const int64_t n3 = (int64_t)3 * c->S.N_bod * c->S.N_blb;
    if (mode == 5) {                                   // G^-1 = H^-1 L^-1
      if ((rc = blk_solve(c, 0, nb, d_in, d_out, 1, 0, 1, false))) return rc;
      return c->tl_ok ? tl_apply(c, d_out, d_out, 1, n3, 1) : RBL_OK;
    }
    if ((rc = rbl_dev_reserve(c, c->d_tlZ, sizeof(double) * 3 * (size_t)n3))) return rc;
    double *t = (double *)c->d_tlZ.p;
    RBL_HIP(c, hipMemcpyAsync(t, d_in, sizeof(double) * (size_t)n3, hipMemcpyDeviceToDevice, c->stream));
    if (mode == 6) {                                   // G^-T = L^-T H^-T
      if (c->tl_ok && (rc = tl_apply(c, t, t, 1, n3, 2))) return rc;
      return blk_solve(c, 0, nb, t, d_out, 1, 0, 2, false);
    }
    if (c->tl_ok && (rc = tl_apply(c, t, t, 1, n3, 0))) return rc;       // G x = L (H x)
    return blk_trmv(c, 0, nb, t, d_out);
  }
  if (mode == 3 && d_in == d_out) return rbl_fail(c, RBL_ERR_ARG, "block_solve_dev: mode 3 does not work in place");
  rc = mode == 3 ? blk_trmv(c, body_begin, nb, d_in, d_out) : blk_solve(c, body_begin, nb, d_in, d_out, 1, 0, mode);
  if (rc) return rbl_fail(c, rc, "block_solve_dev: bodies with more than 2730 blobs are not supported");
  return RBL_OK;
}

int rbl_set_block_refresh(rbl_ctx *c, int every)
{
  if (!c || every < 1) return rbl_fail(c, RBL_ERR_ARG, "set_block_refresh: every >= 1");
  c->blk_refresh = every; c->blk_age = 0; c->dev_blk_valid = false;
  return RBL_OK;
}

int rbl_block_solve_dev(rbl_ctx *c, const double *d_in, double *d_out, int mode)
{
  return rbl_block_solve_range_dev(c, d_in, d_out, mode, 0, -1);
}

// transient switch: the matvec entry points skip the damping B (plain, wall-corrected M) while it is on
int rbl_set_no_damp(rbl_ctx *c, int on)
{
  if (!c) return RBL_ERR_ARG;
  c->no_damp = on != 0;
  return RBL_OK;
}

int rbl_apply_M_sym_multi_dev(rbl_ctx *c, const double *d_F, const double *d_r, int64_t n_blobs, int nrhs,
                              int i_first, int i_step, double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n_blobs <= 0 || i_step < 1 || i_first < 0 || i_first >= i_step || nrhs < 1 || nrhs > 2)
    return rbl_fail(c, RBL_ERR_SIZE, "apply_M_sym_multi_dev: need n_blobs > 0, 0 <= i_first < i_step, nrhs 1 or 2");
  if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(n_blobs, c->n_cu, i_step, nrhs, c->sym_tune)))) return rc;
  rbl_launch_apply_M_sym(c->stream, ctx_params(c), c->S.wall, d_F, d_r, n_blobs, i_first,
                         i_step, d_out, (double *)c->d_part.p, c->n_cu, c->d_err, nrhs, c->sym_tune);
  return RBL_OK;
}

int rbl_apply_M_sym_dev(rbl_ctx *c, const double *d_F, const double *d_r, int64_t n_blobs, int i_first,
                        int i_step, double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n_blobs <= 0 || i_step < 1 || i_first < 0 || i_first >= i_step)
    return rbl_fail(c, RBL_ERR_SIZE, "apply_M_sym_dev: need n_blobs > 0 and 0 <= i_first < i_step");
  if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(n_blobs, c->n_cu, i_step, 1, c->sym_tune)))) return rc;
  RblSymTune tune = c->sym_tune;
  if (c->force_relaxed) tune.relaxed = 1;
  rbl_launch_apply_M_sym(c->stream, ctx_params(c), c->S.wall, d_F, d_r, n_blobs, i_first,
                         i_step, d_out, (double *)c->d_part.p, c->n_cu, c->d_err, 1, tune);
  return RBL_OK;
}

int rbl_apply_M_sym_info(rbl_ctx *c, int64_t n_blobs, int i_step, int nrhs, int *rows_per_lane, int *chunk_tiles,
                         int64_t *workspace_bytes)
{
  if (!c || n_blobs <= 0 || i_step < 1 || nrhs < 1 || nrhs > 2) return rbl_fail(c, RBL_ERR_ARG, "apply_M_sym_info: bad arguments");
  int rc = rbl_dev_init(c); if (rc) return rc;
  int ni = 0, ch = 0;
  const size_t b = rbl_apply_M_sym_bytes(n_blobs, c->n_cu, i_step, nrhs, c->sym_tune, &ni, &ch);
  if (rows_per_lane) *rows_per_lane = ni;
  if (chunk_tiles) *chunk_tiles = ch;
  if (workspace_bytes) *workspace_bytes = (int64_t)b;
  return RBL_OK;
}

int rbl_rotne_prager_tensor_dev(rbl_ctx *c, const double *d_r, int64_t n_blobs, int scale_damp,
                                double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  rbl_launch_build_M(c->stream, rbl_make_params(c->S.a, c->S.eta), c->S.wall, scale_damp != 0, d_r,
                     n_blobs, d_out, c->d_err);
  return RBL_OK;
}

int rbl_cholesky_lower_dev(rbl_ctx *c, double *d_M, int64_t n, int zero_upper)
{
  if (!c) return RBL_ERR_ARG;
  int rc = rbl_dev_init(c); if (rc) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_chol, rbl_cholesky_work_bytes(n)))) return rc;
  rc = rbl_launch_cholesky(c->stream, d_M, n, zero_upper != 0, c->d_err, (double *)c->d_chol.p, c->d_chol.bytes, &c->chol_aux);
  return rc ? rbl_fail(c, rc, "cholesky launch failed") : RBL_OK;
}

int rbl_trmv_lower_dev(rbl_ctx *c, const double *d_L, int64_t n, const double *d_W, double *d_out)
{
  if (!c) return RBL_ERR_ARG;
  int rc = rbl_dev_init(c); if (rc) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_tmp, rbl_trmv_part_bytes(n)))) return rc;
  rbl_launch_trmv_lower(c->stream, d_L, n, d_W, d_out, (double *)c->d_tmp.p);
  return RBL_OK;
}

int rbl_set_comm(rbl_ctx *c, int rank, int world, rbl_allreduce_fn fn, void *user)
{
  if (!c || world < 1 || rank < 0 || rank >= world) return rbl_fail(c, RBL_ERR_ARG, "set_comm: need 0 <= rank < world");
  c->comm_rank = rank; c->comm_world = (fn ? world : 1); c->comm_fn = fn; c->comm_user = user;
  return RBL_OK;
}

int rbl_sync_check(rbl_ctx *c)
{
  if (!c) return RBL_ERR_ARG;
  int rc = rbl_dev_init(c); if (rc) return rc;
  return finish_and_check(c);
}

int rbl_set_tuning(rbl_ctx *c, int jsplit, int variant)
{
  if (!c) return RBL_ERR_ARG;
  if (variant == 31 || variant == 32) { c->gmres_pc_sign_fix = (variant == 32); return RBL_OK; }
  if (variant == 41 || variant == 42) { c->gmres_small = (variant == 42); return RBL_OK; }           // one-kernel GMRES for small systems off / on
  if (variant == 73 || variant == 74) { c->bf_wall_approx = (variant == 74); c->dev_pc_valid = false; c->dev_blk_valid = false; return RBL_OK; }   // wall case: free-space body-frame factor as an APPROXIMATE block factor off / on
  if (variant >= 63 && variant <= 65) { c->blk_large = variant - 63; c->dev_blk_valid = false; c->blk_inv_valid = false; c->bf_valid = false; c->dev_pc_valid = false; return RBL_OK; }   // explicit inverses of large bodies never / always / when it pays
  if (variant == 83 || variant == 84) { c->blk_f32 = (variant == 84); c->dev_blk_valid = false; c->blk_inv_valid = false; c->dev_pc_valid = false; return RBL_OK; }   // single-precision copy of the large inverses off / on
  if (variant == 93 || variant == 94) { c->sym_tune.queue = (variant == 93) ? -1 : 0; return RBL_OK; }   // large systems: one unit per workgroup in launch order / work queue (default)
  if (variant == 91 || variant == 92) { c->gmres_predict = (variant == 92); c->gmres_last_used = 0; return RBL_OK; }   // launch-bound GMRES: convergence test every 4th iteration / where the previous solve and the residual's rate put it (default)
  if (variant == 87 || variant == 88) { c->tl_on = (variant == 88); c->tl_valid = false; return RBL_OK; }   // preconditioned root: block-Jacobi factor alone / two-level factor (default)
  if (variant == 85 || variant == 86) { c->lanczos_out_norm = (variant == 86); return RBL_OK; }       // preconditioned root: stop on the energy-norm / increment-norm (default) estimate
  if (variant == 81 || variant == 82) { c->lanczos_reorth = (variant == 82); return RBL_OK; }       // Lanczos: three-term recurrence only / full re-orthogonalisation (default)
  if (variant == 71 || variant == 72) { c->blk_bodyframe = (variant == 72); c->bf_valid = false; c->dev_pc_valid = false; c->dev_blk_valid = false; c->blk_inv_valid = false; return RBL_OK; }   // body-frame factors in free space off / on
  if (variant == 61 || variant == 62) { c->blk_explicit = (variant == 62); c->dev_blk_valid = false; c->blk_inv_valid = false; c->bf_valid = false; c->dev_pc_valid = false; return RBL_OK; }   // explicit inverses of small bodies off / on
  if (variant == 51 || variant == 52) { c->gmres_relax = (variant == 52); return RBL_OK; }           // inexact-Krylov relaxed products in GMRES off / on
  if (variant == 53 || variant == 54) { c->force_relaxed = (variant == 54); return RBL_OK; }         // hook: every full product relaxed off / on   // GMRES: reference-sign / restored-sign PC
  if (variant == 21 || variant == 22) { c->sym_tune.ni2 = variant - 20; return RBL_OK; }   // experiment: rows per lane of the 2-vector kernel
  c->sym_tune.chunk = (variant == 2 ? jsplit : 0);   // with the symmetric kernel forced, jsplit = chunk length C
  c->tune_jsplit = jsplit; c->tune_variant = variant;
  return RBL_OK;
}

}  // extern "C"

// ---- device-resident body state + geometric operators ------------
static int sync_bodies(rbl_ctx *c)
{
  int rc = need_config(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (c->dev_bodies_valid) return RBL_OK;
  RblBodyState &S = c->S;
  const size_t N = (size_t)S.N_bod * S.N_blb;
  if ((rc = ensure_xq_dev(c))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_lever, sizeof(double) * 3 * N))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_pos, sizeof(double) * 3 * N))) return rc;
  const double *dX = (const double *)c->d_XQ.p, *dQ = dX + 3 * (size_t)S.N_bod;
  rbl_launch_body_geom(c->stream, dX, dQ, (const double *)c->d_cfg.p, S.N_blb, (int64_t)N, (double *)c->d_lever.p,
                       (double *)c->d_pos.p);
  c->dev_bodies_valid = true;
  c->dev_pc_valid = false;
  c->tl_valid = false;
  // the per-body Cholesky factors follow every configuration change unless the caller asked to keep them for a few
  // (rbl_set_block_refresh): as a preconditioner, or as the L of B L (L^-1 M L^-T)^{1/2} W, any nearby factor serves
  if (c->dev_blk_valid && ++c->blk_age >= c->blk_refresh) c->dev_blk_valid = false;   // blk_age: changes since the build
  return RBL_OK;
}

extern "C" {

int rbl_sync_bodies_dev(rbl_ctx *c) { return sync_bodies(c); }

int rbl_positions_dev(rbl_ctx *c, const double **d_pos, int64_t *n_blobs)
{
  int rc = sync_bodies(c); if (rc) return rc;
  if (d_pos) *d_pos = (const double *)c->d_pos.p;
  if (n_blobs) *n_blobs = (int64_t)c->S.N_bod * c->S.N_blb;
  return RBL_OK;
}

int rbl_K_x_U_dev(rbl_ctx *c, const double *d_U, double *d_out)
{
  int rc = sync_bodies(c); if (rc) return rc;
  rbl_launch_K_x_U(c->stream, (const double *)c->d_lever.p, d_U, c->S.N_blb, (int64_t)c->S.N_bod * c->S.N_blb, d_out,
                   nullptr, 0.0);
  return RBL_OK;
}

int rbl_KT_x_Lam_dev(rbl_ctx *c, const double *d_lam, double *d_out)
{
  int rc = sync_bodies(c); if (rc) return rc;
  rbl_launch_KT_x_Lam(c->stream, (const double *)c->d_lever.p, d_lam, c->S.N_blb, c->S.N_bod, d_out);
  return RBL_OK;
}

// Block_diag_invM on the device (:461-487): per-body dense mobility (batched k_build_M), batched
// in-place Cholesky on the matrix cores, then invM_b v = (L L^T)^-1 v by k_block_solve.
// per-body mobility (wall-corrected per wall_PC, undamped) and its Cholesky factor, for every body at once
// Bodies [b0, b1) (default: all).  Storage is always laid out for all bodies (body b at offset b); factors that are
// valid for a range containing the requested one are re-used, otherwise exactly the requested range is rebuilt.
static int pc_block_factors(rbl_ctx *c, int b0, int b1)
{
  const RblBodyState &S = c->S;
  if (b1 < 0) b1 = S.N_bod;
  if (b0 < 0 || b0 >= b1 || b1 > S.N_bod) return rbl_fail(c, RBL_ERR_ARG, "block factors: need 0 <= body_begin < body_end <= N_bodies");
  if (c->dev_blk_valid && c->blk_b0 <= b0 && b1 <= c->blk_b1) return RBL_OK;
  RblPhase ph(c, RBL_T_FACTOR);
  const int64_t m = 3 * (int64_t)S.N_blb, msz = m * m;
  const size_t lstride = rbl_cholesky_batched_work_bytes(m, 1) / sizeof(double);      // L_kk^-1 blocks of one body
  int rc;
  if ((rc = rbl_dev_reserve(c, c->d_blkL, sizeof(double) * (size_t)msz * S.N_bod))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_blkLinv, rbl_cholesky_batched_work_bytes(m, S.N_bod)))) return rc;
  const RblParams P = rbl_make_params(S.a, S.eta);
  double *Lb = (double *)c->d_blkL.p + (size_t)b0 * (size_t)msz;
  for (int q0 = b0; q0 < b1; q0 += 65535)               // bodies ride in gridDim.z
    rbl_launch_build_M_batched(c->stream, P, S.wall, (const double *)c->d_pos.p + (size_t)q0 * (size_t)m, S.N_blb,
                               (b1 - q0 < 65535) ? b1 - q0 : 65535, Lb + (size_t)(q0 - b0) * (size_t)msz, msz, c->d_err);
  rc = rbl_launch_cholesky_batched(c->stream, Lb, m, b1 - b0, msz, c->d_err, (double *)c->d_blkLinv.p + (size_t)b0 * lstride);
  if (rc) return rbl_fail(c, rc, "batched cholesky launch failed");
  c->blk_inv_valid = false; c->blk_f32_valid = false;
  if (c->blk_explicit && m > 512 && (c->blk_large == 1 || (c->blk_large == 2 && comm_on(c)))) {
    // large bodies (shell_N_642 / 2562): explicit inverses through the factorisation's own MFMA kernels -- a rank's few
    // bodies are then applied by batched triangular matrix-vector products over the whole chip instead of one latency
    // chain of 3 N_blb / 32 steps per body on one CU each
    int chunk = 1;
    if ((rc = rbl_dev_reserve(c, c->d_blkX, rbl_block_inverse_bytes(m, S.N_bod)))) return rc;
    if (c->blk_f32 && (rc = rbl_dev_reserve(c, c->d_blkXf, rbl_block_inverse_bytes(m, S.N_bod) / 2))) return rc;
    if ((rc = rbl_dev_reserve(c, c->d_blkAug, rbl_block_inverse_large_aug_bytes(m, b1 - b0, &chunk)))) return rc;
    if ((rc = rbl_launch_block_inverse_large(c->stream, Lb, m, b1 - b0, msz, (const double *)c->d_blkLinv.p + (size_t)b0 * lstride,
                                             (double *)c->d_blkX.p + (size_t)b0 * 2 * (size_t)(rbl_block_inverse_ld(m) * m),
                                             c->blk_f32 ? (float *)c->d_blkXf.p + (size_t)b0 * 2 * (size_t)(rbl_block_inverse_ld(m) * m) : nullptr,
                                             (double *)c->d_blkAug.p)))
      return rbl_fail(c, rc, "block inverse (large bodies) launch failed");
    c->blk_inv_valid = true; c->blk_f32_valid = c->blk_f32;
  }
  if (c->blk_explicit && rbl_block_inverse_fits(m)) {     // small bodies: explicit L^-1, sweeps become matrix-vector products
    if ((rc = rbl_dev_reserve(c, c->d_blkX, rbl_block_inverse_bytes(m, S.N_bod)))) return rc;
    if ((rc = rbl_launch_block_inverse(c->stream, Lb, m, b1 - b0, msz, (const double *)c->d_blkLinv.p + (size_t)b0 * lstride,
                                       (double *)c->d_blkX.p + (size_t)b0 * 2 * (size_t)msz)))
      return rbl_fail(c, rc, "block inverse launch failed");
    c->blk_inv_valid = true;
  }
  c->dev_blk_valid = true; c->blk_b0 = b0; c->blk_b1 = b1; c->blk_age = 0;
  c->tl_valid = false;
  return RBL_OK;
}

static int pc_block_build(rbl_ctx *c)
{
  RblPhase ph(c, RBL_T_FACTOR);
  const RblBodyState &S = c->S;
  const int64_t m = 3 * (int64_t)S.N_blb, N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  int b0 = 0, b1 = S.N_bod;                              // multi-GPU: this rank's bodies only (rbl_set_comm)
  if (comm_on(c)) comm_body_range(c, &b0, &b1);
  const int nbo = b1 - b0;
  const size_t off = (size_t)b0 * (size_t)m;
  int rc;
  if (nbo > 0 && (rc = blk_prepare(c, b0, b1))) return rc;
  if (bf_on(c) && c->bf_tables) return RBL_OK;           // free space, small bodies: everything was built with the body-frame factor
  if ((rc = rbl_dev_reserve(c, c->d_NL, sizeof(double) * 36 * (size_t)S.N_bod))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_pcw, sizeof(double) * (size_t)(2 * n3 + 6 * 6 * S.N_bod + 2 * 6 * S.N_bod)))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_pcMK, sizeof(double) * 6 * (size_t)n3))) return rc;
  // Ninv_b = K_b^T invM_b K_b, column by column (bodies do not couple), then its 6x6 Cholesky; the six
  // solved columns invM_b K_b are kept (d_pcMK): every application needs invM K U
  double *w1 = (double *)c->d_pcw.p, *cols = w1 + 2 * n3, *Uunit = cols + 36 * (size_t)S.N_bod;
  (void)w1;
  double *MK = (double *)c->d_pcMK.p;
  for (int cc = 0; cc < 6; ++cc) {                       // the six columns of K ...
    rbl_launch_unit_U(c->stream, S.N_bod, cc, Uunit);
    rbl_launch_K_x_U(c->stream, (const double *)c->d_lever.p, Uunit, S.N_blb, N, MK + (size_t)cc * n3, nullptr, 0.0);
  }
  if (nbo <= 0) return RBL_OK;
  // ... solved in place, three per pass over the factors (the sweeps are latency chains: 6 single solves cost 10 ms at cfg 3)
  if ((rc = blk_solve(c, b0, nbo, MK, MK, 6, n3, 0)))
    return rbl_fail(c, rc, "block-diagonal PC: bodies with more than 2730 blobs are not supported on the device");
  for (int cc = 0; cc < 6; ++cc)
    rbl_launch_KT_x_Lam(c->stream, (const double *)c->d_lever.p + off, MK + (size_t)cc * n3 + off, S.N_blb, nbo,
                        cols + (size_t)cc * 6 * S.N_bod + (size_t)6 * b0);
  rbl_launch_pc_block_ninv(c->stream, cols, S.N_bod, (double *)c->d_NL.p, c->d_err, b0, b1);
  return RBL_OK;
}

static int pc_block_apply_local(rbl_ctx *c, const double *d_in, double *d_out, bool shard)
{
  RblPhase ph(c, RBL_T_PERBODY);
  const RblBodyState &S = c->S;
  const int64_t m = 3 * (int64_t)S.N_blb, N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  double *w1 = (double *)c->d_pcw.p, *w2 = w1 + n3, *f6 = w2 + n3 + 36 * (size_t)S.N_bod + 6 * (size_t)S.N_bod;
  const double *lev = (const double *)c->d_lever.p;
  int b0 = 0, b1 = S.N_bod;
  if (shard) comm_body_range(c, &b0, &b1);
  const int nbo = b1 - b0;
  const size_t off = (size_t)b0 * (size_t)m;
  int rc;
  if (shard) RBL_HIP(c, hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)(n3 + 6 * S.N_bod), c->stream));
  c->ktl_of = nullptr;
  if (bf_on(c) && c->bf_tables) {                        // the whole application in the body frame, one launch
    if ((rc = ensure_xq_dev(c))) return rc;
    if (nbo > 0) {
      double *ktl = nullptr;
      if (c->ktl_arm && !shard) {
        if ((rc = rbl_dev_reserve(c, c->d_ktl, sizeof(double) * 6 * (size_t)S.N_bod))) return rc;
        ktl = (double *)c->d_ktl.p;
      }
      const double *T = (const double *)c->d_bfPC.p;
      if ((rc = rbl_dev_reserve(c, c->d_blkTmp, sizeof(double) * 3 * (size_t)n3))) return rc;
      if ((rc = rbl_launch_pc_bodyframe(c->stream, T, T + (size_t)(m * m), T + (size_t)(m * m) + 6 * (size_t)m, (const double *)c->d_cfg.p,
                                        (const double *)c->d_XQ.p + 3 * (size_t)S.N_bod, m, b0, nbo, d_in, n3, c->pc_fsign, d_out, ktl,
                                        (double *)c->d_blkTmp.p)))
        return rbl_fail(c, rc, "body-frame preconditioner launch failed");
      if (ktl) c->ktl_of = d_out;
    }
    return RBL_OK;
  }
  if (nbo > 0) {
    if ((rc = blk_solve(c, b0, nbo, d_in, w1, 1, 0, 0))) return rc;                                      // invM slip
    // K^T (invM slip);  U (:601-608);  Lambda = invM (slip + K U) (:610) = invM slip + (invM K) U: no second pass over
    // the factors -- one launch (k_pc_block_tail); inside GMRES it also leaves K^T Lambda for the saddle product
    double *ktl = nullptr;
    if (c->ktl_arm && !shard) {
      if ((rc = rbl_dev_reserve(c, c->d_ktl, sizeof(double) * 6 * (size_t)S.N_bod))) return rc;
      ktl = (double *)c->d_ktl.p;
    }
    rbl_launch_pc_block_tail(c->stream, lev, w1, (const double *)c->d_pcMK.p, n3, (const double *)c->d_NL.p, d_in + n3, S.N_blb,
                             b0, nbo, c->pc_fsign, d_out + n3, d_out, ktl);
    if (ktl) c->ktl_of = d_out;
  }
  (void)f6; (void)off;
  return RBL_OK;
}

static int pc_block_apply(rbl_ctx *c, const double *d_in, double *d_out)
{
  const bool shard = comm_on(c);                         // own bodies only, completed by one all-reduce of the result
  int rc = pc_block_apply_local(c, d_in, d_out, shard);
  if (rc || !shard) return rc;
  return comm_allreduce(c, d_out, (int64_t)3 * c->S.N_bod * c->S.N_blb + (int64_t)6 * c->S.N_bod);
}

int rbl_apply_PC_dev(rbl_ctx *c, const double *d_in, double *d_out)
{
  int rc = sync_bodies(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  if (S.block_pc) {
    if (!c->dev_pc_valid) {
      if ((rc = pc_block_build(c))) return rc;
      c->dev_pc_valid = true;
    }
    return pc_block_apply(c, d_in, d_out);
  }
  if (!c->dev_pc_valid) {
    const size_t N = (size_t)S.N_bod * S.N_blb;
    if ((rc = rbl_dev_reserve(c, c->d_invM2, sizeof(double) * 2 * N))) return rc;
    if ((rc = rbl_dev_reserve(c, c->d_NL, sizeof(double) * 36 * (size_t)S.N_bod))) return rc;
    rbl_launch_pc_diag_build(c->stream, rbl_make_params(S.a, S.eta), S.wall, (const double *)c->d_lever.p,
                             (const double *)c->d_pos.p, S.N_blb, S.N_bod, (double *)c->d_invM2.p, (double *)c->d_NL.p,
                             c->d_err);
    c->dev_pc_valid = true;
  }
  rbl_launch_pc_diag_apply(c->stream, (const double *)c->d_lever.p, (const double *)c->d_invM2.p, (const double *)c->d_NL.p,
                           S.N_blb, S.N_bod, d_in, d_out, c->pc_fsign);
  return RBL_OK;
}

// Everything a solver iteration needs that is NOT a plain kernel launch (uploads, workspace
// growth, preconditioner build) done now, so that the iteration itself -- apply_saddle_dev,
// apply_PC_dev, K ops -- is launch-only and can be captured in a hipGraph.
int rbl_prepare_dev(rbl_ctx *c)
{
  int rc = sync_bodies(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  if ((rc = rbl_dev_reserve(c, c->d_sad, sizeof(double) * (size_t)n3))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(N, c->n_cu, 1, 1, c->sym_tune)))) return rc;
  if (!c->dev_pc_valid) {   // build the preconditioner eagerly (apply on a scratch vector)
    const size_t nv = (size_t)n3 + 6 * (size_t)S.N_bod;
    if ((rc = rbl_dev_reserve(c, c->d_tmp, sizeof(double) * 2 * nv))) return rc;
    RBL_HIP(c, hipMemsetAsync(c->d_tmp.p, 0, sizeof(double) * 2 * nv, c->stream));
    if ((rc = rbl_apply_PC_dev(c, (const double *)c->d_tmp.p, (double *)c->d_tmp.p + nv))) return rc;
  }
  return finish_and_check(c);
}

// [M lambda - K U ; K^T lambda] on the object's own configuration (src/Rigid.py:73-80)
int rbl_apply_saddle_dev(rbl_ctx *c, const double *d_x, double *d_out)
{
  int rc = sync_bodies(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  if ((rc = rbl_dev_reserve(c, c->d_sad, sizeof(double) * (size_t)n3))) return rc;
  if ((rc = apply_M_enqueue(c, S.wall, d_x, (const double *)c->d_pos.p, N, 0, N, (double *)c->d_sad.p))) return rc;
  if (c->ktl_arm && c->ktl_of == d_x) {      // GMRES: d_x came out of the block preconditioner together with its K^T Lambda
    rbl_launch_saddle_tail(c->stream, (const double *)c->d_lever.p, d_x + n3, S.N_blb, N, S.N_bod, d_out,
                           (const double *)c->d_sad.p, (const double *)c->d_ktl.p);
    return RBL_OK;
  }
  rbl_launch_K_x_U(c->stream, (const double *)c->d_lever.p, d_x + n3, S.N_blb, N, d_out, (const double *)c->d_sad.p, -1.0);
  rbl_launch_KT_x_Lam(c->stream, (const double *)c->d_lever.p, d_x, S.N_blb, S.N_bod, d_out + n3);
  return RBL_OK;
}

}  // extern "C"

// ---- GMRES on the saddle operator --------------------------------------------
// Right-preconditioned GMRES(max_iter), no restart:  A = apply_saddle (src/Rigid.py:73-80), P^-1 = apply_PC
// (c_rigid_obj.cpp:589-616), Arnoldi with classical Gram-Schmidt applied twice.  Everything stays on the
// device and the stream is not drained inside the loop: the Hessenberg matrix lives in HBM and is read back
// once at the end (fixed work, rtol <= 0) or, for the convergence test, every iteration (large systems) /
// every 4th (small, launch-bound ones).
extern "C" {

static int gmres_saddle_core(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                             double *resid_out);

// Small systems (<= 256 blobs, diagonal PC): geometry, preconditioner build and the whole Arnoldi / Givens loop in ONE
// kernel launch on one CU (rbl_small.hip) -- launch-bound otherwise (cfg 1: ~6 launches per iteration).
static int gmres_small(rbl_ctx *c, const double *d_rhs, const double *d_x0, int max_iter, double rtol, double *d_x,
                       int *iters_out, double *resid_out)
{
  int rc = ensure_xq_dev(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const size_t wd = rbl_gmres_small_work_doubles(S.N_blb, S.N_bod, max_iter);
  if ((rc = rbl_dev_reserve(c, c->d_gm, sizeof(double) * (wd + 2)))) return rc;
  double *work = (double *)c->d_gm.p, *scal = work + wd;
  const double *dX = (const double *)c->d_XQ.p, *dQ = dX + 3 * (size_t)S.N_bod;
  rc = rbl_launch_gmres_small(c->stream, rbl_make_params(S.a, S.eta), S.wall, dX, dQ, (const double *)c->d_cfg.p, S.N_blb,
                              S.N_bod, d_rhs, d_x0, d_x, max_iter, rtol, c->gmres_pc_sign_fix ? 1.0 : c->pc_fsign, work, scal,
                              c->d_err);
  if (rc == RBL_ERR_SIZE) return rc;       // the caller falls back to the general solver
  if (rc) return rbl_fail(c, rc, "gmres (one-kernel solver): launch failed");
  double hs[2] = {0.0, 0.0};
  RBL_HIP(c, hipMemcpyAsync(hs, scal, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
  if ((rc = finish_and_check(c))) return rc;
  int it = 0;
  std::memcpy(&it, &hs[0], sizeof(int));
  if (iters_out) *iters_out = it;
  if (resid_out) *resid_out = hs[1];
  return RBL_OK;
}

// use_x0 != 0: d_x holds an initial guess (e.g. the previous time step's solution): the solver iterates on the
// residual b - A x0 (one extra product) and the tolerance stays relative to |b|.
int rbl_gmres_saddle_dev(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int use_x0,
                         int *iters_out, double *resid_out)
{
  RblPhase ph_total(c, RBL_T_TOTAL);
  {
    int rc = need_config(c); if (rc) return rc;
    if ((rc = rbl_dev_init(c))) return rc;
    if (!d_rhs || !d_x || max_iter < 1) return rbl_fail(c, RBL_ERR_ARG, "gmres: bad arguments");
    if (c->gmres_small && !comm_on(c) && rbl_gmres_small_fits(c->S.N_blb, c->S.N_bod, max_iter, c->S.block_pc)) {
      rc = gmres_small(c, d_rhs, use_x0 ? d_x : nullptr, max_iter, rtol, d_x, iters_out, resid_out);
      if (rc != RBL_ERR_SIZE) return rc;
      c->gmres_small = false;                // this runtime does not grant the LDS the one-kernel solver needs
    }
  }
  if (!use_x0) return gmres_saddle_core(c, d_rhs, max_iter, rtol, d_x, iters_out, resid_out);
  int rc = sync_bodies(c); if (rc) return rc;
  if (!d_rhs || !d_x) return rbl_fail(c, RBL_ERR_ARG, "gmres: bad arguments");
  const int64_t nsys = (int64_t)3 * c->S.N_bod * c->S.N_blb + (int64_t)6 * c->S.N_bod;
  const size_t vb = sizeof(double) * (size_t)nsys;
  if ((rc = rbl_dev_reserve(c, c->d_bd2, 3 * vb + sizeof(double) * (2 + 2 * 512)))) return rc;   // + dot2 scratch
  double *r0 = (double *)c->d_bd2.p, *dx = r0 + nsys, *x0 = dx + nsys, *dn = x0 + nsys;
  RBL_HIP(c, hipMemcpyAsync(x0, d_x, vb, hipMemcpyDeviceToDevice, c->stream));
  if ((rc = rbl_apply_saddle_dev(c, x0, r0))) return rc;
  rbl_launch_axpby(c->stream, nsys, 1.0, d_rhs, -1.0, r0, r0);                          // r0 = b - A x0
  double nn[2] = {0.0, 0.0};
  rbl_launch_dot2(c->stream, d_rhs, d_rhs, nullptr, nsys, dn);                          // |b|^2
  if ((rc = read_back(c, &nn[0], dn, sizeof(double)))) return rc;
  rbl_launch_dot2(c->stream, r0, r0, nullptr, nsys, dn);                                // |r0|^2
  if ((rc = read_back(c, &nn[1], dn, sizeof(double)))) return rc;
  const double nb2 = nn[0], nr2 = nn[1];
  const double scale = (nb2 > 0.0 && nr2 > 0.0) ? std::sqrt(nb2 / nr2) : 1.0;          // |b| / |r0|
  if (nr2 == 0.0) { if (iters_out) *iters_out = 0; if (resid_out) *resid_out = 0.0; return RBL_OK; }   // x0 already solves it
  double resid = 0.0;
  if ((rc = gmres_saddle_core(c, r0, max_iter, rtol > 0.0 ? rtol * scale : rtol, dx, iters_out, &resid))) return rc;
  rbl_launch_axpby(c->stream, nsys, 1.0, x0, 1.0, dx, d_x);                             // x = x0 + dx
  if (resid_out) *resid_out = resid / scale;
  return finish_and_check(c);
}

static int gmres_saddle_core_(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                              double *resid_out);

// any invertible right preconditioner leaves the solution unchanged: inside the solve the force block of apply_PC
// takes the sign that makes A P^-1 ~ I (see rbl_ctx::pc_fsign); the bound apply_PC keeps the reference's convention
static int gmres_saddle_core(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                             double *resid_out)
{
  const double keep = c->pc_fsign;
  if (c->gmres_pc_sign_fix) c->pc_fsign = 1.0;
  const int rc = gmres_saddle_core_(c, d_rhs, max_iter, rtol, d_x, iters_out, resid_out);
  c->pc_fsign = keep;
  return rc;
}

static int gmres_saddle_core_(rbl_ctx *c, const double *d_rhs, int max_iter, double rtol, double *d_x, int *iters_out,
                              double *resid_out)
{
  int rc = sync_bodies(c); if (rc) return rc;
  if (!d_rhs || !d_x || max_iter < 1) return rbl_fail(c, RBL_ERR_ARG, "gmres: bad arguments");
  if (max_iter + 1 > rbl_gmres_max_vectors()) return rbl_fail(c, RBL_ERR_ARG, "gmres: at most 255 iterations (no restart)");
  const RblBodyState &S = c->S;
  const int64_t nsys = (int64_t)3 * S.N_bod * S.N_blb + (int64_t)6 * S.N_bod;
  const int m = max_iter, ldh = m + 1;
  const size_t vb = sizeof(double) * (size_t)nsys;
  // workspace: V[(m+1)][nsys] | w | z | H[(m+1) x m] column-major | beta | y[m] | partial sums
  const size_t need = vb * (size_t)(m + 3) + sizeof(double) * ((size_t)ldh * m + 1 + m + rbl_gmres_part_doubles() +
                                                               rbl_lanczos_part_doubles());
  if ((rc = rbl_dev_reserve(c, c->d_gm, need))) return rc;
  // (beta sits in FRONT of H: a convergence test reads back 1 + ldh * used doubles, not the whole ldh x m array)
  double *V = (double *)c->d_gm.p, *w = V + (size_t)(m + 1) * nsys, *z = w + nsys, *d_beta = z + nsys, *H = d_beta + 1,
         *d_y = H + (size_t)ldh * m, *part = d_y + m, *part2 = part + rbl_gmres_part_doubles();
  RBL_HIP(c, hipMemsetAsync(H, 0, sizeof(double) * (size_t)ldh * m, c->stream));
  rbl_launch_lanczos_init(c->stream, nsys, d_rhs, d_beta, V, part2);                    // V_0 = b/|b|, beta = |b|
  std::vector<double> Hh((size_t)ldh * m + 1), y;
  // convergence test: every iteration when an iteration is expensive.  When it is launch-bound a test (copy + stream
  // drain) costs as much as half an iteration: the first one waits until two iterations before the count the previous
  // solve needed (time steps repeat), later ones follow the residual's rate, at most four iterations apart; a test
  // looks at every iteration since the one before, so the solve still ends at the first iteration that passes.
  const int check_every = ((int64_t)S.N_bod * S.N_blb > 20000) ? 1 : 4;
  int next_check = check_every, last_checked = 0;
  if (check_every > 1 && c->gmres_predict && c->gmres_last_used > 0) next_check = c->gmres_last_used >= 8 ? c->gmres_last_used - 2 : c->gmres_last_used;
  int used = 0;
  double resid = 1.0;
  // least squares min |beta e1 - H_k y| by Givens rotations on a host copy; returns the residual estimate
  auto solve_ls = [&](int k, std::vector<double> &yout) -> double {
    std::vector<double> R(Hh.begin() + 1, Hh.begin() + 1 + (size_t)ldh * k), g((size_t)k + 1, 0.0);
    const double beta = Hh[0];
    g[0] = beta;
    std::vector<double> cs((size_t)k), sn((size_t)k);
    for (int j = 0; j < k; ++j) {
      double *col = R.data() + (size_t)j * ldh;
      for (int i = 0; i < j; ++i) {
        const double t = cs[i] * col[i] + sn[i] * col[i + 1];
        col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1];
        col[i] = t;
      }
      const double den = std::hypot(col[j], col[j + 1]);
      cs[j] = den > 0.0 ? col[j] / den : 1.0;
      sn[j] = den > 0.0 ? col[j + 1] / den : 0.0;
      col[j] = den; col[j + 1] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
    }
    yout.assign((size_t)k, 0.0);
    for (int i = k - 1; i >= 0; --i) {
      double v = g[i];
      for (int j = i + 1; j < k; ++j) v -= R[(size_t)j * ldh + i] * yout[j];
      const double d = R[(size_t)i * ldh + i];
      yout[i] = d != 0.0 ? v / d : 0.0;
    }
    return beta > 0.0 ? std::fabs(g[k]) / beta : 0.0;
  };
  for (int j = 0; j < m; ++j) {
    const double *vj = V + (size_t)j * nsys;
    c->ktl_arm = true;                                 // the PC's K^T Lambda by-product feeds the product that follows
    if ((rc = rbl_apply_PC_dev(c, vj, z))) { c->ktl_arm = false; return rc; }
    // inexact Krylov: the j-th product may be in error by ~ rtol / |r_{j-1}| (relative); the relaxed kernel's ~1e-6 is
    // admissible once the residual estimate is below rtol x 1e5 (an order of magnitude in hand)
    c->sym_tune.relaxed = (c->gmres_relax && rtol > 0.0 && check_every == 1 && resid <= rtol * 1.0e5) ? 1 : 0;
    rc = rbl_apply_saddle_dev(c, z, w);
    c->sym_tune.relaxed = 0;
    c->ktl_arm = false; c->ktl_of = nullptr;
    if (rc) return rc;
    double *Hcol = H + (size_t)j * ldh;
    // classical Gram-Schmidt twice, H[j+1][j] = |w|, V_{j+1} = w / |w|: four launches
    rbl_launch_arnoldi_step(c->stream, V, nsys, j + 1, w, Hcol, V + (size_t)(j + 1) * nsys, part);
    used = j + 1;
    if (rtol > 0.0 && (used >= next_check || used == m)) {
      if ((rc = read_back(c, Hh.data(), d_beta, sizeof(double) * (1 + (size_t)ldh * used)))) return rc;
      // the test may have become true anywhere since the last look: take the first k that passes
      int hit = 0;
      double r_before = resid;
      for (int k = last_checked + 1; k <= used; ++k) {
        r_before = resid;
        resid = solve_ls(k, y);
        if (resid < rtol) { hit = k; break; }
      }
      if (hit) { used = hit; break; }
      last_checked = used;
      int ahead = 1;
      if (check_every > 1 && !c->gmres_predict) ahead = check_every - (used % check_every);
      else if (check_every > 1) {                        // iterations the residual still needs at its current rate
        ahead = check_every;
        if (resid > 0.0 && r_before > resid) {
          const double rem = std::log(rtol / resid) / std::log(resid / r_before);
          ahead = rem < 1.0 ? 1 : (rem > (double)check_every ? check_every : (int)rem);
        }
      }
      next_check = used + ahead;
    }
  }
  if (!(rtol > 0.0) || y.size() != (size_t)used) {
    if ((rc = read_back(c, Hh.data(), d_beta, sizeof(double) * (1 + (size_t)ldh * used)))) return rc;
    resid = solve_ls(used, y);
  }
  for (int k = 0; k < used; ++k)
    if (!std::isfinite(y[k])) return rbl_fail(c, RBL_ERR_NONFINITE, "gmres: non-finite Hessenberg solve");
  if ((rc = upload_coef(c, d_y, y.data(), used, 0))) return rc;
  rbl_launch_lanczos_combine(c->stream, nsys, V, d_y, used, z);                        // z = V y
  if ((rc = rbl_apply_PC_dev(c, z, d_x))) return rc;                                   // x = P^-1 z
  if (iters_out) *iters_out = used;
  if (resid_out) *resid_out = resid;
  if (rtol > 0.0) c->gmres_last_used = used;
  return finish_and_check(c);
}

}  // extern "C"

// ---- whole time steps in one call (what krylov.py's steppers do, for hosts without a Python driver) ----------
extern "C" {

static int step_buffers(rbl_ctx *c, int64_t n3, int64_t nb6, double **rhs, double **x, double **slip, double **force)
{
  const int64_t nsys = n3 + nb6;
  int rc = rbl_dev_reserve(c, c->d_step, sizeof(double) * (size_t)(2 * nsys + n3 + nb6));
  if (rc) return rc;
  if (c->step_x_size != nsys) { c->step_hist_n = 0; c->step_x_size = nsys; }
  *x = (double *)c->d_step.p;
  *rhs = *x + nsys;
  *slip = *rhs + nsys;
  *force = *slip + n3;
  return RBL_OK;
}

// One deterministic time step on the object's own configuration: solve [M -K; K^T 0][lambda; U] = [slip; -F] by
// right-preconditioned GMRES (rbl_gmres_saddle_dev), then evolve_X_Q(U) (:865-878).  F_body: host, 6 N_bod;
// slip: host, 3 N_blobs, or NULL for zero.  warm_start: 0 cold; 1 start from the previous call's solution x_n; 2 from
// 2 x_n - x_{n-1}; 3 from 3 x_n - 3 x_{n-1} + x_{n-2} (as far as the history reaches): under a smooth forcing the solution
// moves smoothly with the configuration, and at cfg 3 GMRES then needs 12 / 6 / 2-3 iterations to 1e-8 instead of 18.
int rbl_step_deterministic(rbl_ctx *c, const double *F_body, const double *slip, int max_iter, double rtol,
                           int warm_start, int *iters, double *resid)
{
  int rc = need_K(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!F_body) return rbl_fail(c, RBL_ERR_ARG, "step_deterministic: F_body is NULL");
  const int64_t n3 = (int64_t)3 * c->S.N_bod * c->S.N_blb, nb6 = (int64_t)6 * c->S.N_bod;
  double *rhs, *x, *dslip, *dforce;
  if ((rc = step_buffers(c, n3, nb6, &rhs, &x, &dslip, &dforce))) return rc;
  if (slip) { if ((rc = copy_h2d(c, rhs, slip, sizeof(double) * (size_t)n3))) return rc; }
  else RBL_HIP(c, hipMemsetAsync(rhs, 0, sizeof(double) * (size_t)n3, c->stream));
  if ((rc = copy_h2d(c, dforce, F_body, sizeof(double) * (size_t)nb6))) return rc;
  rbl_launch_axpby(c->stream, nb6, -1.0, dforce, 0.0, nullptr, rhs + n3);
  const int64_t nsys = n3 + nb6;
  if ((rc = rbl_dev_reserve(c, c->d_hist, sizeof(double) * (size_t)(3 * nsys)))) return rc;
  double *H = (double *)c->d_hist.p;
  auto slot = [&](int age) { return H + (size_t)((c->step_hist_head + age) % 3) * (size_t)nsys; };   // age 0 = newest
  int order = warm_start < 0 ? 0 : (warm_start > 3 ? 3 : warm_start);
  if (order > c->step_hist_n) order = c->step_hist_n;
  if (order == 1) RBL_HIP(c, hipMemcpyAsync(x, slot(0), sizeof(double) * (size_t)nsys, hipMemcpyDeviceToDevice, c->stream));
  if (order == 2) rbl_launch_axpby(c->stream, nsys, 2.0, slot(0), -1.0, slot(1), x);
  if (order == 3) {
    rbl_launch_axpby(c->stream, nsys, 3.0, slot(0), -3.0, slot(1), x);
    rbl_launch_axpby(c->stream, nsys, 1.0, x, 1.0, slot(2), x);
  }
  if ((rc = rbl_gmres_saddle_dev(c, rhs, max_iter, rtol, x, order > 0 ? 1 : 0, iters, resid))) { c->step_hist_n = 0; return rc; }
  c->step_hist_head = (c->step_hist_head + 2) % 3;                     // the oldest slot becomes the newest
  RBL_HIP(c, hipMemcpyAsync(slot(0), x, sizeof(double) * (size_t)nsys, hipMemcpyDeviceToDevice, c->stream));
  if (c->step_hist_n < 3) ++c->step_hist_n;
  std::vector<double> U((size_t)nb6);
  if ((rc = copy_d2h(c, U.data(), x + n3, sizeof(double) * (size_t)nb6))) return rc;
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  return rbl_evolve_X_Q(c, U.data());
}

// One stochastic midpoint step: right-hand side and predictor configuration at q^n (rbl_RHS_and_Midpoint_dev,
// reference :917-976), saddle solve at q^{n+1/2}, update from q^n with dt U.  W: host, [W1 | W2 | W_rfd] = 9 N_blobs
// standard normals, or NULL to draw them from `seed`.
int rbl_step_brownian(rbl_ctx *c, const double *F_body, const double *slip, const double *W, uint64_t seed, int method,
                      int split_rand, double delta, int max_iter, double rtol, int *iters, double *resid)
{
  int rc = need_K(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!F_body) return rbl_fail(c, RBL_ERR_ARG, "step_brownian: F_body is NULL");
  const int Nb = c->S.N_bod;
  const int64_t n3 = (int64_t)3 * Nb * c->S.N_blb, nb6 = (int64_t)6 * Nb;
  double *rhs, *x, *dslip, *dforce;
  if ((rc = step_buffers(c, n3, nb6, &rhs, &x, &dslip, &dforce))) return rc;
  c->step_hist_n = 0;                                       // the random part of the solution does not carry over
  if (slip) { if ((rc = copy_h2d(c, dslip, slip, sizeof(double) * (size_t)n3))) return rc; }
  else RBL_HIP(c, hipMemsetAsync(dslip, 0, sizeof(double) * (size_t)n3, c->stream));
  if ((rc = copy_h2d(c, dforce, F_body, sizeof(double) * (size_t)nb6))) return rc;
  double *dW = nullptr;
  if (W) {
    if ((rc = rbl_dev_reserve(c, c->d_W, sizeof(double) * 3 * (size_t)n3))) return rc;
    dW = (double *)c->d_W.p;
    if ((rc = copy_h2d(c, dW, W, sizeof(double) * 3 * (size_t)n3))) return rc;
  }
  const std::vector<double> Xn = c->S.X, Qn = c->S.Q;
  std::vector<double> Xh((size_t)3 * Nb), Qh((size_t)4 * Nb);
  if ((rc = rbl_RHS_and_Midpoint_dev(c, dslip, dforce, dW, seed, method, split_rand, delta, rhs, Xh.data(), Qh.data())))
    return rc;
  if ((rc = rbl_set_config(c, Xh.data(), Qh.data(), Nb))) return rc;       // operators at the predictor configuration
  rc = rbl_gmres_saddle_dev(c, rhs, max_iter, rtol, x, 0, iters, resid);
  std::vector<double> U((size_t)nb6);
  if (!rc) rc = copy_d2h(c, U.data(), x + n3, sizeof(double) * (size_t)nb6);
  if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = RBL_ERR_HIP;
  const int rc2 = rbl_set_config(c, Xn.data(), Qn.data(), Nb);              // the update starts from q^n (also on failure)
  if (rc) return rc;
  if (rc2) return rc2;
  return rbl_evolve_X_Q(c, U.data());
}

}  // extern "C"

// ---- random finite differences (reference C++-only members) ---------------
extern "C" {

// core of M_RFD(), c_rigid_obj.cpp:776-794: d_out = (1/delta)[M(q + delta/2 dq) - M(q - delta/2 dq)] W with
// dq = Kinv W.  Wh = host copy of W (Kinv is O(N) host work), d_r: n3 scratch, d_work: 2 n3 scratch.
static int m_rfd_core(rbl_ctx *c, const double *d_W, const double *Wh, double delta, double *d_out,
                      double *d_r, double *d_work)
{
  RblPhase ph_total(c, RBL_T_TOTAL);
  RblBodyState &S = c->S;
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  std::vector<double> uom((size_t)6 * S.N_bod), win((size_t)6 * S.N_bod), Xs, Qs;
  rbl_body_Kinv_x_V(S, Wh, uom.data());                               // UOM = Kinv W (:776)
  const std::vector<double> X0 = S.X, Q0 = S.Q;
  double *dM[2] = {d_work, d_work + n3};
  int rc = RBL_OK;
  for (int sgn = 0; sgn < 2; ++sgn) {                                 // q +- delta/2 dq (:783-788)
    const double f = (sgn == 0 ? 0.5 : -0.5) * delta;
    for (size_t i = 0; i < win.size(); ++i) win[i] = f * uom[i];
    rbl_body_update_X_Q(S, win.data(), Xs, Qs);
    S.X = Xs; S.Q = Qs; c->dev_xq_valid = false;                      // displaced configuration, temporarily
    rc = positions_dev(c, 0, S.N_bod, d_r);
    if (!rc) rc = apply_M_enqueue(c, S.wall, d_W, d_r, N, 0, N, dM[sgn]);   // :790-791
    S.X = X0; S.Q = Q0; c->dev_xq_valid = false;
    if (rc) return rc;
  }
  rbl_launch_axpby(c->stream, n3, 1.0 / delta, dM[0], -1.0 / delta, dM[1], d_out);   // :793
  return RBL_OK;
}

// M_RFD(), c_rigid_obj.cpp:769-796.  The two products run on the GPU at the two displaced configurations.
int rbl_M_RFD(rbl_ctx *c, const double *W, uint64_t seed, double delta, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!(delta > 0.0)) return rbl_fail(c, RBL_ERR_ARG, "M_RFD: delta must be positive");
  RblBodyState &S = c->S;
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  const size_t vb = sizeof(double) * (size_t)n3;
  if ((rc = rbl_dev_reserve(c, c->d_W, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_r, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_U, 2 * vb))) return rc;
  std::vector<double> Wh((size_t)n3);
  if (W) {
    std::memcpy(Wh.data(), W, vb);
    if ((rc = copy_h2d(c, c->d_W.p, W, vb))) return rc;
  } else {  // rand_vector (:730-741) replaced by the seeded device generator
    rbl_launch_normal(c->stream, seed, 0, n3, (double *)c->d_W.p);
    if ((rc = copy_d2h(c, Wh.data(), c->d_W.p, vb))) return rc;
    RBL_HIP(c, hipStreamSynchronize(c->stream));
  }
  double *dU = (double *)c->d_U.p;
  if ((rc = m_rfd_core(c, (const double *)c->d_W.p, Wh.data(), delta, dU, (double *)c->d_r.p, dU))) return rc;
  if ((rc = copy_d2h(c, out, dU, vb))) return rc;
  return finish_and_check(c);
}

// update_X_Q(U), c_rigid_obj.cpp:798-863: the configuration displaced by U (displacement units: translation
// and rotation vector per body), WITHOUT committing it.
int rbl_update_X_Q(rbl_ctx *c, const double *U, double *X_out, double *Q_out)
{
  int rc = need_config(c); if (rc) return rc;
  if (!U || !X_out || !Q_out) return rbl_fail(c, RBL_ERR_ARG, "update_X_Q: null argument");
  std::vector<double> Xo, Qo;
  rbl_body_update_X_Q(c->S, U, Xo, Qo);
  std::memcpy(X_out, Xo.data(), sizeof(double) * Xo.size());
  std::memcpy(Q_out, Qo.data(), sizeof(double) * Qo.size());
  return RBL_OK;
}

// RHS_and_Midpoint(Slip, Force), c_rigid_obj.cpp:917-976 -- device-resident form.  d_W = [W1 | W2 | W_rfd]
// (3 n3) or NULL (drawn from `seed`).  d_RHS = [Slip - (kBT M_RFD + BI) ; -Force]  (n3 + 6 N_bod).
int rbl_RHS_and_Midpoint_dev(rbl_ctx *c, const double *d_Slip, const double *d_Force, const double *d_W,
                             uint64_t seed, int method, int split_rand, double delta, double *d_RHS,
                             double *X_half, double *Q_half)
{
  int rc = need_K(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!d_Slip || !d_Force || !d_RHS || !X_half || !Q_half) return rbl_fail(c, RBL_ERR_ARG, "RHS_and_Midpoint: null argument");
  RblBodyState &S = c->S;
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N, nb6 = (int64_t)6 * S.N_bod;
  const size_t vb = sizeof(double) * (size_t)n3;
  rbl_launch_axpby(c->stream, nb6, -1.0, d_Force, 0.0, nullptr, d_RHS + n3);          // Force *= -1 (:972)
  if (!(S.kBT > 1e-10)) {                                                              // no Brownian terms (:967-970)
    RBL_HIP(c, hipMemcpyAsync(d_RHS, d_Slip, vb, hipMemcpyDeviceToDevice, c->stream));
    std::memcpy(X_half, S.X.data(), sizeof(double) * S.X.size());
    std::memcpy(Q_half, S.Q.data(), sizeof(double) * S.Q.size());
    return finish_and_check(c);
  }
  if (!(S.dt > 0.0) || !(delta > 0.0)) return rbl_fail(c, RBL_ERR_ARG, "RHS_and_Midpoint: dt and delta must be positive");
  // workspace: [W1 | W2 | W_rfd] (when drawn here), M^{1/2}W1, M^{1/2}W2, M_RFD, positions, 2 scratch
  if ((rc = rbl_dev_reserve(c, c->d_bd, 9 * vb))) return rc;
  double *base = (double *)c->d_bd.p;
  double *dWown = base, *dMW = base + 3 * n3 /* 2 vectors */, *dRFD = base + 5 * n3, *dr = base + 6 * n3,
         *dwork = base + 7 * n3;
  if (!d_W) {                                                                          // rand_vector (:730-741)
    rbl_launch_normal(c->stream, seed, 0, 3 * n3, dWown);
    d_W = dWown;
  }
  const int nvec = split_rand ? 2 : 1;
  if ((rc = positions_dev(c, 0, S.N_bod, dr))) return rc;                              // multi_body_pos (:662)
  if ((rc = mhalf_dev_multi(c, dr, N, d_W, nvec, method, dMW))) return rc;             // M_half_W1/2 (:927-936)
  std::vector<double> Wh((size_t)n3), mw1((size_t)n3);
  if ((rc = copy_d2h(c, Wh.data(), d_W + 2 * n3, vb))) return rc;
  if ((rc = copy_d2h(c, mw1.data(), dMW, vb))) return rc;
  RBL_HIP(c, hipStreamSynchronize(c->stream));
  if ((rc = m_rfd_core(c, d_W + 2 * n3, Wh.data(), delta, dRFD, dr, dwork))) return rc;  // M_RFD (:940)
  const double c1 = split_rand ? 2.0 * std::sqrt(S.kBT / S.dt) : std::sqrt(2.0 * S.kBT / S.dt);   // :945-952
  const double c2 = split_rand ? std::sqrt(S.kBT / S.dt) : std::sqrt(2.0 * S.kBT / S.dt);
  // Slip -= kBT M_RFD + BI,  BI = c2 (M^{1/2}W1 - M^{1/2}W2)  or  c2 M^{1/2}W1   (:948,953,963)
  rbl_launch_axpby(c->stream, n3, 1.0, d_Slip, -S.kBT, dRFD, d_RHS);
  rbl_launch_axpby(c->stream, n3, 1.0, d_RHS, -c2, dMW, d_RHS);
  if (split_rand) rbl_launch_axpby(c->stream, n3, 1.0, d_RHS, c2, dMW + n3, d_RHS);
  // predictor: q^{n+1/2} = q^n displaced by (dt/2) Kinv (c1 M^{1/2}W1)   (:955-959)
  std::vector<double> uom((size_t)nb6), Xo, Qo;
  rbl_body_Kinv_x_V(S, mw1.data(), uom.data());
  for (double &u : uom) u *= 0.5 * S.dt * c1;
  rbl_body_update_X_Q(S, uom.data(), Xo, Qo);
  std::memcpy(X_half, Xo.data(), sizeof(double) * Xo.size());
  std::memcpy(Q_half, Qo.data(), sizeof(double) * Qo.size());
  return finish_and_check(c);
}

// host-pointer form of the same
int rbl_RHS_and_Midpoint(rbl_ctx *c, const double *Slip, const double *Force, const double *W, uint64_t seed,
                         int method, int split_rand, double delta, double *RHS, double *X_half, double *Q_half)
{
  int rc = need_K(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!Slip || !Force || !RHS) return rbl_fail(c, RBL_ERR_ARG, "RHS_and_Midpoint: null argument");
  const int64_t n3 = (int64_t)3 * c->S.N_bod * c->S.N_blb, nb6 = (int64_t)6 * c->S.N_bod;
  const size_t vb = sizeof(double) * (size_t)n3, fb = sizeof(double) * (size_t)nb6;
  if ((rc = rbl_dev_reserve(c, c->d_bd2, (W ? 4 : 1) * vb + 2 * (vb + fb)))) return rc;
  double *dSlip = (double *)c->d_bd2.p, *dForce = dSlip + n3, *dRHS = dForce + nb6, *dW = dRHS + n3 + nb6;
  if ((rc = copy_h2d(c, dSlip, Slip, vb))) return rc;
  if ((rc = copy_h2d(c, dForce, Force, fb))) return rc;
  if (W && (rc = copy_h2d(c, dW, W, 3 * vb))) return rc;
  if ((rc = rbl_RHS_and_Midpoint_dev(c, dSlip, dForce, W ? dW : nullptr, seed, method, split_rand, delta, dRHS,
                                     X_half, Q_half))) return rc;
  if ((rc = copy_d2h(c, RHS, dRHS, vb + fb))) return rc;
  return finish_and_check(c);
}

// KTinv_RFD(), c_rigid_obj.cpp:743-767:  K^T (1/delta) [ Kinv(q+)^T - Kinv(q-)^T ] W, W of length 6 N_bod
int rbl_KTinv_RFD(rbl_ctx *c, const double *W, double delta, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  if (!W || !(delta > 0.0)) return rbl_fail(c, RBL_ERR_ARG, "KTinv_RFD: need W and delta > 0");
  const RblBodyState &S = c->S;
  const size_t n3 = (size_t)3 * S.N_bod * S.N_blb;
  std::vector<double> win((size_t)6 * S.N_bod), acc(n3, 0.0), tmp(n3);
  for (int sgn = 0; sgn < 2; ++sgn) {
    const double f = (sgn == 0 ? 0.5 : -0.5) * delta;
    for (size_t i = 0; i < win.size(); ++i) win[i] = f * W[i];
    RblBodyState T = S;                                              // displaced copy (:755-761)
    rbl_body_update_X_Q(S, win.data(), T.X, T.Q);
    if ((rc = rbl_body_set_K(T, c->last_error))) return rc;
    rbl_body_KTinv_x_F(T, W, tmp.data());
    const double w = (sgn == 0 ? 1.0 : -1.0) / delta;
    for (size_t i = 0; i < n3; ++i) acc[i] += w * tmp[i];            // :763-764
  }
  rbl_body_KT_x_Lam(S, acc.data(), out);                             // :766
  return RBL_OK;
}

}  // extern "C"
