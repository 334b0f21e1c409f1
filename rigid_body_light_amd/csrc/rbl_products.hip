// rbl_products.hip -- mobility products: kernel choice and launch for one / several vectors, positions, dense build and Cholesky entry points.
// Part of the implementation of the C ABI in include/rbl.h (split from the former rbl_api.hip along its sections);
// shared internals are declared in rbl_api_internal.hpp.  Nothing here falls back to a CPU path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <vector>

#include "rbl_api_internal.hpp"

#define SYM_SHAPE_MSG "symmetric product: the options (sym_rows_per_lane / sym2_rows_per_lane / sym_waves / sym_chunk) force a kernel shape that does not exist"

int apply_M_enqueue(rbl_ctx *c, bool wall, const double *d_F, const double *d_r, int64_t nbl,
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
  if (full && comm_on(c) && c->comm_split == 1) {
    // multi-GPU, rows by body index: the ordered-pair kernel on this rank's rows, then ONE all-gather of U.
    // Rows follow the bodies when the vector is the object's own configuration, an even split of the blobs otherwise.
    std::vector<int64_t> bounds((size_t)c->comm_world + 1, 0);
    const bool own = c->S.cfg_set && nbl == (int64_t)c->S.N_bod * c->S.N_blb;
    for (int r = 0; r < c->comm_world; ++r) {
      if (own) { int b0, b1; comm_body_range_of(c, r, &b0, &b1); bounds[(size_t)r + 1] = (int64_t)b1 * c->S.N_blb; }
      else bounds[(size_t)r + 1] = nbl * (int64_t)(r + 1) / c->comm_world;
    }
    const int64_t r0 = bounds[(size_t)c->comm_rank], r1 = bounds[(size_t)c->comm_rank + 1];
    int js = 1;
    const size_t pb = rbl_apply_M_part_bytes(nbl, r1 - r0, c->n_cu, c->tune_jsplit, &js);
    if ((rc = rbl_dev_reserve(c, c->d_part, pb))) return rc;
    if (comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(d_out, 0, sizeof(double) * 3 * (size_t)nbl, c->stream));
    if (r1 > r0) {
      RblPhase ph(c, RBL_T_PRODUCT);
      rbl_launch_apply_M(c->stream, P, wall, d_F, d_r, nbl, r0, r1, d_out + 3 * r0, (double *)c->d_part.p, js, 0, c->d_err);
    }
    return comm_allgather_rows(c, d_out, bounds.data(), 3);
  }
  if (full && comm_on(c)) {   // multi-GPU: this rank's tile pairs, then the sum over the ranks
    RblSymTune tune = c->sym_tune;
    tune.fuse = RblSaddleFuse();                      // (a shard's sum is partial: the epilogue waits for the all-reduce)
    if (c->force_relaxed) tune.relaxed = 1;
    if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, c->comm_world, 1, tune)))) return rc;   // (the geometry follows the transient switches)
    {
      RblPhase ph(c, RBL_T_PRODUCT);
      if ((rc = rbl_launch_apply_M_sym(c->stream, P, wall, d_F, d_r, nbl, c->comm_rank, c->comm_world, d_out, (double *)c->d_part.p,
                                       c->n_cu, c->d_err, 1, tune)))
        return rbl_fail(c, rc, SYM_SHAPE_MSG);
    }
    return comm_allreduce(c, d_out, 3 * nbl);
  }
  RblPhase ph(c, RBL_T_PRODUCT);
  if (sym) {
    RblSymTune tune = c->sym_tune;
    if (c->force_relaxed) tune.relaxed = 1;
    if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, 1, 1, tune)))) return rc;   // (the geometry follows the transient switches)
    if ((rc = rbl_launch_apply_M_sym(c->stream, P, wall, d_F, d_r, nbl, 0, 1, d_out, (double *)c->d_part.p, c->n_cu,
                                     c->d_err, 1, tune)))
      return rbl_fail(c, rc, SYM_SHAPE_MSG);
    c->fuse_done = tune.fuse.lever != nullptr;        // the slab reduction also wrote the saddle epilogue (rbl_apply_saddle_dev)
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
int apply_M_multi_enqueue(rbl_ctx *c, bool wall, const double *d_F, const double *d_r, int64_t nbl,
                                 int nrhs, double *d_out, int64_t ldF, int64_t ldO)
{
  const int64_t n3 = 3 * nbl;
  if (ldF <= 0) ldF = n3;                               // doubles between consecutive vectors (default: packed)
  if (ldO <= 0) ldO = n3;
  if (ldF != n3 || ldO != n3) {                          // strided vectors (the lock-step GMRES): one at a time unless the MFMA kernel takes them
    bool mf = nrhs >= 4;
    if (c->tune_variant == 3) mf = true;
    if (c->tune_variant == 1 || c->tune_variant == 2) mf = false;
    if (comm_on(c) || !mf) {
      for (int k = 0; k < nrhs; ++k) {
        const int rc1 = apply_M_enqueue(c, wall, d_F + (size_t)k * (size_t)ldF, d_r, nbl, 0, nbl, d_out + (size_t)k * (size_t)ldO);
        if (rc1) return rc1;
      }
      return RBL_OK;
    }
    int rc2;
    RblPhase ph2(c, RBL_T_PRODUCT);
    if ((rc2 = rbl_dev_reserve(c, c->d_part, rbl_apply_M_mrhs_bytes(nbl, c->n_cu)))) return rc2;
    const RblParams P2 = ctx_params(c);
    for (int k = 0; k < nrhs; k += 16) {
      const int nb = (nrhs - k < 16) ? nrhs - k : 16;
      rbl_launch_apply_M_mrhs(c->stream, P2, wall, d_F + (size_t)k * (size_t)ldF, d_r, nbl, nb, d_out + (size_t)k * (size_t)ldO,
                              (double *)c->d_part.p, c->n_cu, c->d_err, ldF, ldO);
    }
    return RBL_OK;
  }
  bool mfma = nrhs >= 4;
  if (c->tune_variant == 3) mfma = true;
  if (c->tune_variant == 1 || c->tune_variant == 2) mfma = false;
  int rc;
  if (comm_on(c)) {   // multi-GPU: pairs of vectors through the sharded two-vector kernel, one all-reduce per pair
    int k = 0;
    if (c->comm_split == 1) {   // (the row split has no two-vector kernel: one ordered-pair pass per vector)
      for (; k < nrhs; ++k)
        if ((rc = apply_M_enqueue(c, wall, d_F + (size_t)k * n3, d_r, nbl, 0, nbl, d_out + (size_t)k * n3))) return rc;
      return RBL_OK;
    }
    for (; k + 2 <= nrhs; k += 2) {
      RblSymTune tune = c->sym_tune;
      if (c->force_relaxed) tune.relaxed = 1;
      if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, c->comm_world, 2, tune)))) return rc;   // (the geometry follows the transient switches)
      {
        RblPhase ph(c, RBL_T_PRODUCT);
        if ((rc = rbl_launch_apply_M_sym(c->stream, ctx_params(c), wall, d_F + (size_t)k * n3, d_r, nbl, c->comm_rank, c->comm_world,
                                         d_out + (size_t)k * n3, (double *)c->d_part.p, c->n_cu, c->d_err, 2, tune)))
          return rbl_fail(c, rc, SYM_SHAPE_MSG);
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
      RblSymTune tune = c->sym_tune;
      if (c->force_relaxed) tune.relaxed = 1;
      if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(nbl, c->n_cu, 1, 2, tune)))) return rc;   // (the geometry follows the transient switches)
      if ((rc = rbl_launch_apply_M_sym(c->stream, ctx_params(c), wall, d_F + (size_t)k * n3, d_r, nbl, 0, 1,
                                       d_out + (size_t)k * n3, (double *)c->d_part.p, c->n_cu, c->d_err, 2, tune)))
        return rbl_fail(c, rc, SYM_SHAPE_MSG);
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

// (X, Q, ref_cfg) resident on the device: uploaded once per configuration change (pageable host
// vectors -> one synchronisation there), so the per-step position kernel is launch-only.
int ensure_xq_dev(rbl_ctx *c)
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
int positions_dev(rbl_ctx *c, int b0, int b1, double *d_out)
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

// multi_body_pos (:295-300) into a device vector, the way the context's communicator splits the work: every rank all bodies
// (single GPU; tile-pair split: the O(N_bod) body state is replicated and the kernel is cheaper than a collective), or --
// row split, north_star's "all-gather of blob positions before the all-pairs pass" -- this rank's bodies + ONE all-gather
int rbl_multi_body_pos_dev(rbl_ctx *c, double *d_out)
{
  int rc = need_config(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!d_out) return rbl_fail(c, RBL_ERR_ARG, "multi_body_pos_dev: null argument");
  const RblBodyState &S = c->S;
  if (!(comm_on(c) && c->comm_split == 1)) return positions_dev(c, 0, S.N_bod, d_out);
  int b0, b1; comm_body_range(c, &b0, &b1);
  if (comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(d_out, 0, sizeof(double) * 3 * (size_t)S.N_bod * S.N_blb, c->stream));
  if (b1 > b0 && (rc = positions_dev(c, b0, b1, d_out + 3 * (size_t)b0 * S.N_blb))) return rc;
  return comm_allgather_bodies(c, d_out, 0, 3 * (int64_t)S.N_blb, 1, 0);
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

int rbl_apply_M_sym_multi_dev(rbl_ctx *c, const double *d_F, const double *d_r, int64_t n_blobs, int nrhs,
                              int i_first, int i_step, double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n_blobs <= 0 || i_step < 1 || i_first < 0 || i_first >= i_step || nrhs < 1 || nrhs > 2)
    return rbl_fail(c, RBL_ERR_SIZE, "apply_M_sym_multi_dev: need n_blobs > 0, 0 <= i_first < i_step, nrhs 1 or 2");
  if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(n_blobs, c->n_cu, i_step, nrhs, c->sym_tune)))) return rc;
  if ((rc = rbl_launch_apply_M_sym(c->stream, ctx_params(c), c->S.wall, d_F, d_r, n_blobs, i_first,
                                   i_step, d_out, (double *)c->d_part.p, c->n_cu, c->d_err, nrhs, c->sym_tune)))
    return rbl_fail(c, rc, SYM_SHAPE_MSG);
  return RBL_OK;
}

int rbl_apply_M_sym_dev(rbl_ctx *c, const double *d_F, const double *d_r, int64_t n_blobs, int i_first,
                        int i_step, double *d_out)
{
  int rc = need_params(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (n_blobs <= 0 || i_step < 1 || i_first < 0 || i_first >= i_step)
    return rbl_fail(c, RBL_ERR_SIZE, "apply_M_sym_dev: need n_blobs > 0 and 0 <= i_first < i_step");
  RblSymTune tune = c->sym_tune;
  if (c->force_relaxed) tune.relaxed = 1;
  if ((rc = rbl_dev_reserve(c, c->d_part, rbl_apply_M_sym_bytes(n_blobs, c->n_cu, i_step, 1, tune)))) return rc;   // (the geometry follows the transient switches)
  if ((rc = rbl_launch_apply_M_sym(c->stream, ctx_params(c), c->S.wall, d_F, d_r, n_blobs, i_first,
                                   i_step, d_out, (double *)c->d_part.p, c->n_cu, c->d_err, 1, tune)))
    return rbl_fail(c, rc, SYM_SHAPE_MSG);
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

int rbl_apply_M_sym_kernel(rbl_ctx *c, int64_t n_blobs, int i_step, int nrhs, int wall, char *name, int name_len)
{
  if (!c || n_blobs <= 0 || i_step < 1 || nrhs < 1 || nrhs > 2 || !name || name_len < 40)
    return rbl_fail(c, RBL_ERR_ARG, "apply_M_sym_kernel: bad arguments (name buffer of >= 40 bytes)");
  int rc = rbl_dev_init(c); if (rc) return rc;
  rbl_apply_M_sym_kernel_name(n_blobs, c->n_cu, i_step, nrhs, c->sym_tune, wall != 0, name, (size_t)name_len);
  if (!name[0]) return rbl_fail(c, RBL_ERR_ARG, SYM_SHAPE_MSG);     // the same answer a product of that size would get
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
