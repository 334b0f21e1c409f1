// rbl_bodies.hip -- per-body operators: K / K^T / K^-1 (host and device), preconditioners, per-body factors and their applications, the saddle operator.
// Part of the implementation of the C ABI in include/rbl.h (split from the former rbl_api.hip along its sections);
// shared internals are declared in rbl_api_internal.hpp.  Nothing here falls back to a CPU path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "rbl_api_internal.hpp"

int rbl_set_K_mats(rbl_ctx *c)
{
  int rc = need_config(c);
  if (rc) return rc;
  return rbl_body_set_K(c->S, c->last_error);
}

int need_K(rbl_ctx *c)
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
  c->dev_bodies_valid = false; c->dev_pc_valid = false; c->dev_xq_valid = false; c->pc_keep_once = false;
  rc = rbl_body_set_K(S, c->last_error);                          // :876
  S.pc_set = false;                                               // :877
  return rc;
}

// op(L_b) applied to bodies [b0, b0 + nbo) of nv vectors `pitch` doubles apart (in / out: the FULL vectors, body 0 first);
// mode 0: (L L^T)^-1, 1: L^-1, 2: L^-T.  Small bodies go through their explicit inverses (two matrix-vector products
// instead of two chains of substitution steps), the others through the substitution kernel.  In place is fine.
// Free space (no wall term in M): every body's mobility is the SAME body-frame matrix seen through the body's rotation,
// M_b = (I x R_b) M_body (I x R_b)^T (the RPY block of a pair depends on the separation vector only, which rotates with the
// body).  So there is nothing to factor per configuration: M_body = L L^T once per rbl_set_parameters, and the factor used
// for body b is G_b = (I x R_b) L  (G G^T = M_b; not triangular, which nothing here needs):
//   (G G^T)^-1 v = R (L L^T)^-1 R^T v,   G^-1 v = L^-1 R^T v,   G^-T v = R L^-T v,   G x = R L x.
// One matrix for all bodies also means the factor is read from cache instead of HBM.
bool bf_on(const rbl_ctx *c) { return c->blk_bodyframe && (!c->S.wall || c->bf_wall_approx); }

int bf_build(rbl_ctx *c)
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
  if (c->blk_explicit && rbl_block_inverse_large_fits(m) && c->blk_large != 0) {   // large bodies: ONE explicit inverse for all bodies and all time
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
int blk_prepare(rbl_ctx *c, int b0, int b1)
{
  const int hi = b1 < 0 ? c->S.N_bod : b1;
  if (b0 < 0 || b0 >= hi || hi > c->S.N_bod) return rbl_fail(c, RBL_ERR_ARG, "block factors: need 0 <= body_begin < body_end <= N_bodies");
  if (bf_on(c)) return bf_build(c);
  return pc_block_factors(c, b0, b1);
}

int blk_solve(rbl_ctx *c, int b0, int nbo, const double *in, double *out, int nv, int64_t pitch, int mode, bool allow_f32)
{
  if (nbo <= 0) return RBL_OK;
  RblPhase ph(c, RBL_T_PERBODY);
  const int64_t m = 3 * (int64_t)c->S.N_blb, msz = m * m;
  const size_t off = (size_t)b0 * (size_t)m;
  // the forms that go through the scratch vectors (body-frame factor, explicit inverses) lay them out `pitch` apart like the
  // caller's: fine for vectors one blob-vector apart (every caller before the lock-step GMRES), out of the scratch's bounds for
  // vectors further apart -- those go one at a time
  if (nv > 1 && pitch != m * (int64_t)c->S.N_bod && (bf_on(c) || c->blk_inv_valid)) {
    for (int v = 0; v < nv; ++v) {
      const int rc1 = blk_solve(c, b0, nbo, in + (size_t)v * (size_t)pitch, out + (size_t)v * (size_t)pitch, 1, 0, mode, allow_f32);
      if (rc1) return rc1;
    }
    return RBL_OK;
  }
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
        const int form = c->shared_gemm ? 0 : 2;          // (bit 1 of the last argument: batched matrix-vector form instead of the MFMA product)
        if (mode == 0) rc = rbl_launch_block_inv_apply(c->stream, (const double *)c->d_bfX.p, m, nbo, pi, po, m, g, pitch, 0, tmp + off, dQ, form);
        else if (pi != po) rc = rbl_launch_block_inv_apply(c->stream, (const double *)c->d_bfX.p, m, nbo, pi, po, m, g, pitch, mode, nullptr, dQ, form);
        else {
          rc = rbl_launch_block_inv_apply(c->stream, (const double *)c->d_bfX.p, m, nbo, pi, tmp + off, m, g, pitch, mode, nullptr, dQ, form);
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
    // the single-precision copy (large bodies, RBL_OPT_BLOCK_INVERSE_F32) serves whoever tolerates a factor that is exact to 6e-8 only
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
                                      (const double *)c->d_blkLinv.p + (size_t)b0 * lstride, in + off, out + off, m, nv, pitch,
                                      mode | (c->blk_pipe ? 0 : 0x200));
}

// out = G_b in for bodies [b0, b0 + nbo) of ONE vector (in / out: the full vectors; not in place)
int blk_trmv(rbl_ctx *c, int b0, int nbo, const double *in, double *out)
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

// the same for nv vectors `pitch` doubles apart: ONE launch on the matrix cores where every body shares the body-frame factor
int blk_trmv_multi(rbl_ctx *c, int b0, int nbo, const double *in, double *out, int nv, int64_t pitch)
{
  if (nbo <= 0 || nv <= 0) return RBL_OK;
  const int64_t m = 3 * (int64_t)c->S.N_blb;
  if (nv > 1 && bf_on(c) && c->shared_gemm && rbl_shared_gemm_fits(m)) {
    RblPhase ph(c, RBL_T_PERBODY);
    int rc = ensure_xq_dev(c); if (rc) return rc;
    const size_t off = (size_t)b0 * (size_t)m;
    const double *dQ0 = (const double *)c->d_XQ.p + 3 * (size_t)c->S.N_bod + 4 * (size_t)b0;
    return rbl_launch_shared_gemm(c->stream, (const double *)c->d_bfL.p, m, m, 1, in + off, out + off, m, pitch, nbo, nv, dQ0, 2);   // y_b = R_b (L x_b)
  }
  for (int v = 0; v < nv; ++v) {
    const int rc = blk_trmv(c, b0, nbo, in + (size_t)v * (size_t)pitch, out + (size_t)v * (size_t)pitch);
    if (rc) return rc;
  }
  return RBL_OK;
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
  if (rc) return rbl_fail(c, rc, "block_solve_dev: the per-body factor application failed");
  return RBL_OK;
}

int rbl_block_solve_dev(rbl_ctx *c, const double *d_in, double *d_out, int mode)
{
  return rbl_block_solve_range_dev(c, d_in, d_out, mode, 0, -1);
}

// ---- device-resident body state + geometric operators ------------
int sync_bodies(rbl_ctx *c)
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
  if (comm_on(c) && c->comm_split == 1) {
    // row split (north_star / SURVEY.md 8e): every rank evaluates the geometry of ITS bodies, one fused all-gather shares the
    // blob positions (and the lever arms the replicated K operators work with) before the all-pairs pass
    int b0, b1; comm_body_range(c, &b0, &b1);
    const size_t o3 = 3 * (size_t)b0 * (size_t)S.N_blb;
    if (comm_gather_needs_zero(c)) {
      RBL_HIP(c, hipMemsetAsync(c->d_lever.p, 0, sizeof(double) * 3 * N, c->stream));
      RBL_HIP(c, hipMemsetAsync(c->d_pos.p, 0, sizeof(double) * 3 * N, c->stream));
    }
    rbl_launch_body_geom(c->stream, dX + 3 * (size_t)b0, dQ + 4 * (size_t)b0, (const double *)c->d_cfg.p, S.N_blb,
                         (int64_t)(b1 - b0) * S.N_blb, (double *)c->d_lever.p + o3, (double *)c->d_pos.p + o3);
    if ((rc = comm_allgather_bodies2(c, (double *)c->d_lever.p, 0, 3 * (int64_t)S.N_blb, (double *)c->d_pos.p, 0, 3 * (int64_t)S.N_blb))) return rc;
  } else
    rbl_launch_body_geom(c->stream, dX, dQ, (const double *)c->d_cfg.p, S.N_blb, (int64_t)N, (double *)c->d_lever.p,
                         (double *)c->d_pos.p);
  c->dev_bodies_valid = true;
  if (c->pc_keep_once) { c->pc_keep_once = false; return RBL_OK; }   // rbl_evolve_X_Q_RFD (:892): the preconditioner of q serves q + delta U
  c->dev_pc_valid = false;
  if (c->tl_valid && c->tl_ok && ++c->tl_age < c->tl_refresh) c->tl_q_stale = true;   // coarse operator kept (RBL_OPT_TWO_LEVEL_REFRESH), its basis not
  else c->tl_valid = false;
  // the per-body Cholesky factors follow every configuration change unless the caller asked to keep them for a few
  // (rbl_set_block_refresh): as a preconditioner, or as the L of B L (L^-1 M L^-T)^{1/2} W, any nearby factor serves
  if (c->dev_blk_valid && ++c->blk_age >= c->blk_refresh) c->dev_blk_valid = false;   // blk_age: changes since the build
  return RBL_OK;
}

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
int pc_block_factors(rbl_ctx *c, int b0, int b1)
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
                               (b1 - q0 < 65535) ? b1 - q0 : 65535, Lb + (size_t)(q0 - b0) * (size_t)msz, msz, c->d_err,
                               (c->blk_tile && rbl_tile_cholesky_fits(m)) ? 128 : 0);    // (the tile factorisation never reads above its diagonal tiles)
  const bool want_inv = c->blk_explicit && rbl_block_inverse_large_fits(m) && (c->blk_large == 1 || (c->blk_large == 2 && comm_on(c)));
  c->blk_inv_valid = false; c->blk_f32_valid = false;
  if (c->blk_tile && rbl_tile_cholesky_fits(m)) {
    // large bodies (shell_N_642 / 2562): factor and -- where wanted -- explicit inverse in ONE dataflow launch over 128 x 128 tiles
    double *Xb = nullptr; float *Xf = nullptr;
    if (want_inv) {
      if ((rc = rbl_dev_reserve(c, c->d_blkX, rbl_block_inverse_bytes(m, S.N_bod)))) return rc;
      if (c->blk_f32 && (rc = rbl_dev_reserve(c, c->d_blkXf, rbl_block_inverse_bytes(m, S.N_bod) / 2))) return rc;
      Xb = (double *)c->d_blkX.p + (size_t)b0 * 2 * (size_t)(rbl_block_inverse_ld(m) * m);
      if (c->blk_f32) Xf = (float *)c->d_blkXf.p + (size_t)b0 * 2 * (size_t)(rbl_block_inverse_ld(m) * m);
    }
    if ((rc = rbl_dev_reserve(c, c->d_blkAug, rbl_tile_cholesky_work_bytes(m, b1 - b0)))) return rc;
    rc = rbl_launch_tile_cholesky(c->stream, Lb, m, b1 - b0, msz, c->d_err, (double *)c->d_blkLinv.p + (size_t)b0 * lstride, Xb, Xf,
                                  c->d_blkAug.p, c->n_cu);
    if (rc) return rbl_fail(c, rc, "tile cholesky launch failed");
    if (want_inv) { c->blk_inv_valid = true; c->blk_f32_valid = c->blk_f32; }
  } else {
  rc = rbl_launch_cholesky_batched(c->stream, Lb, m, b1 - b0, msz, c->d_err, (double *)c->d_blkLinv.p + (size_t)b0 * lstride);
  if (rc) return rbl_fail(c, rc, "batched cholesky launch failed");
  if (want_inv) {
    // the round-3 form: explicit inverses through the factorisation's own MFMA kernels on an augmented matrix [L ; I]
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
  }
  if (c->blk_explicit && rbl_block_inverse_fits(m)) {     // small bodies: explicit L^-1, sweeps become matrix-vector products
    if ((rc = rbl_dev_reserve(c, c->d_blkX, rbl_block_inverse_bytes(m, S.N_bod)))) return rc;
    if ((rc = rbl_launch_block_inverse(c->stream, Lb, m, b1 - b0, msz, (const double *)c->d_blkLinv.p + (size_t)b0 * lstride,
                                       (double *)c->d_blkX.p + (size_t)b0 * 2 * (size_t)msz)))
      return rbl_fail(c, rc, "block inverse launch failed");
    c->blk_inv_valid = true;
  }
  c->dev_blk_valid = true; c->blk_b0 = b0; c->blk_b1 = b1; c->blk_age = 0;
  // new per-body factors: the two-level factor's basis Q = orth(L^-1 K_t) follows; its coarse operator may stay (any does) while
  // RBL_OPT_TWO_LEVEL_REFRESH keeps it (sync_bodies did the counting)
  if (c->tl_valid && c->tl_ok && c->tl_refresh > 1) c->tl_q_stale = true;
  else c->tl_valid = false;
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
    return rbl_fail(c, rc, "block-diagonal PC: the per-body factor application failed");
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
  if (shard && comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)(n3 + 6 * S.N_bod), c->stream));
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
                                        (double *)c->d_blkTmp.p, c->shared_gemm ? 1 : 0, c->pc_fold.part ? &c->pc_fold : nullptr)))
        return rbl_fail(c, rc, "body-frame preconditioner launch failed");
      c->pc_fold = RblNormFold();
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
                             b0, nbo, c->pc_fsign, d_out + n3, d_out, ktl, c->pc_fold.part ? &c->pc_fold : nullptr, d_in);
    c->pc_fold = RblNormFold();
    if (ktl) c->ktl_of = d_out;
  }
  (void)f6; (void)off;
  return RBL_OK;
}

static int pc_block_apply(rbl_ctx *c, const double *d_in, double *d_out)
{
  const bool shard = comm_on(c);                         // own bodies only, completed by ONE all-gather of the owners' [lambda | U] segments
  int rc = pc_block_apply_local(c, d_in, d_out, shard);
  if (rc || !shard) return rc;
  const int64_t n3 = (int64_t)3 * c->S.N_bod * c->S.N_blb;
  return comm_allgather_bodies2(c, d_out, 0, 3 * (int64_t)c->S.N_blb, d_out, n3, 6);
}

// may the next rbl_apply_PC_dev be handed an un-normalised Arnoldi vector (rbl_ctx::pc_fold)?  On one GPU, once the preconditioner
// has been built (its first application builds it): the diagonal one, the per-body factors' tail kernel, and the free-space
// body-frame tables in their matrix-vector form
bool pc_can_fold(rbl_ctx *c)
{
  const RblBodyState &S = c->S;
  if (!(c->gmres_fold_norm && c->fused_krylov && c->dev_pc_valid && !comm_on(c) && S.N_bod <= 65535)) return false;
  if (S.block_pc && bf_on(c) && c->bf_tables) return rbl_pc_bodyframe_folds(S.N_bod, c->shared_gemm ? 1 : 0);
  return true;
}

int rbl_apply_PC_dev(rbl_ctx *c, const double *d_in, double *d_out)
{
  struct FoldGuard {                                     // whatever happens, the request to fold a normalisation dies with this call
    rbl_ctx *c;
    explicit FoldGuard(rbl_ctx *c_) : c(c_) {}
    ~FoldGuard() { c->pc_fold = RblNormFold(); }
  } guard(c);
  int rc = sync_bodies(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  if (c->pc_fold.part && !pc_can_fold(c)) { c->pc_fold = RblNormFold(); return rbl_fail(c, RBL_ERR_ARG, "apply_PC: this preconditioner does not normalise its input"); }
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
                           S.N_blb, S.N_bod, d_in, d_out, c->pc_fsign, c->pc_fold.part ? &c->pc_fold : nullptr);
  c->pc_fold = RblNormFold();
  return RBL_OK;
}

// out_v = P^-1 in_v for nv saddle vectors `pitch` doubles apart (in, out and the scratch vectors all laid out alike).  The block
// preconditioner of per-configuration factors on one GPU sends all vectors through the factors together -- three share one pass
// over the 5.9 GB of cfg 3 (rbl_launch_block_solve_multi), so 16 vectors cost six passes, not sixteen; every other
// preconditioner (diagonal, body frame, sharded) is applied vector by vector.  d_scratch: nv vectors of >= n3 doubles.
int apply_PC_multi_dev(rbl_ctx *c, const double *d_in, double *d_out, double *d_scratch, int nv, int64_t pitch)
{
  int rc = sync_bodies(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  if (S.block_pc && !c->dev_pc_valid) {                  // (FIRST: whether the body-frame tables serve this preconditioner is only known once it is built)
    if ((rc = pc_block_build(c))) return rc;
    c->dev_pc_valid = true;
  }
  const bool together = S.block_pc && !comm_on(c) && !(bf_on(c) && c->bf_tables) && nv > 1 && d_scratch;
  if (!together) {
    for (int v = 0; v < nv; ++v)
      if ((rc = rbl_apply_PC_dev(c, d_in + (size_t)v * (size_t)pitch, d_out + (size_t)v * (size_t)pitch))) return rc;
    return RBL_OK;
  }
  RblPhase ph(c, RBL_T_PERBODY);
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  c->ktl_of = nullptr;
  if ((rc = blk_solve(c, 0, S.N_bod, d_in, d_scratch, nv, pitch, 0))) return rc;                              // invM slip, all vectors
  for (int v = 0; v < nv; ++v) {
    const double *in = d_in + (size_t)v * (size_t)pitch;
    double *out = d_out + (size_t)v * (size_t)pitch;
    rbl_launch_pc_block_tail(c->stream, (const double *)c->d_lever.p, d_scratch + (size_t)v * (size_t)pitch, (const double *)c->d_pcMK.p, n3,
                             (const double *)c->d_NL.p, in + n3, S.N_blb, 0, S.N_bod, c->pc_fsign, out + n3, out, nullptr, nullptr, in);
  }
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
  const bool have_ktl = c->ktl_arm && c->ktl_of == d_x;   // GMRES: d_x came out of the block preconditioner together with its K^T Lambda
  c->fuse_done = false;
  if (have_ktl && c->fused_krylov && !comm_on(c)) {
    // one launch fewer per iteration: the slab reduction of the product writes  out = [M lambda - K U ; K^T lambda]  itself
    RblSaddleFuse f;
    f.lever = (const double *)c->d_lever.p; f.U = d_x + n3; f.ktl = (const double *)c->d_ktl.p; f.w = d_out;
    f.N_blb = S.N_blb; f.nb6 = 6 * S.N_bod;
    const int64_t np = (n3 + 63) / 64 + 1;              // one partial per block of the reduction + the body rows
    if (c->fuse_dotK > 0 && c->fuse_dotV && c->fuse_dotPart && np <= rbl_gmres_p1_capacity()) {
      f.dotV = c->fuse_dotV; f.dotStride = (long)(n3 + 6 * S.N_bod); f.dotK = c->fuse_dotK; f.dotNp = (int)np; f.dotPart = c->fuse_dotPart;
    }
    c->sym_tune.fuse = f;
  }
  c->fuse_dots_np = 0;
  const int dots_np = c->sym_tune.fuse.dotK > 0 ? c->sym_tune.fuse.dotNp : 0;
  rc = apply_M_enqueue(c, S.wall, d_x, (const double *)c->d_pos.p, N, 0, N, (double *)c->d_sad.p);
  c->sym_tune.fuse = RblSaddleFuse();
  if (rc) return rc;
  if (c->fuse_done) { c->fuse_done = false; c->fuse_dots_np = dots_np; return RBL_OK; }
  if (have_ktl) {
    rbl_launch_saddle_tail(c->stream, (const double *)c->d_lever.p, d_x + n3, S.N_blb, N, S.N_bod, d_out,
                           (const double *)c->d_sad.p, (const double *)c->d_ktl.p);
    return RBL_OK;
  }
  rbl_launch_K_x_U(c->stream, (const double *)c->d_lever.p, d_x + n3, S.N_blb, N, d_out, (const double *)c->d_sad.p, -1.0);
  rbl_launch_KT_x_Lam(c->stream, (const double *)c->d_lever.p, d_x, S.N_blb, S.N_bod, d_out + n3);
  return RBL_OK;
}

// [M lambda - K U ; K^T lambda] for host vectors (src/Rigid.py:73-80): one upload, the device operator, one download
int rbl_apply_saddle(rbl_ctx *c, const double *x, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  if (!x || !out) return rbl_fail(c, RBL_ERR_ARG, "apply_saddle: null argument");
  if ((rc = rbl_dev_init(c))) return rc;
  const size_t nv = (size_t)3 * c->S.N_bod * c->S.N_blb + (size_t)6 * c->S.N_bod;
  if ((rc = rbl_dev_reserve(c, c->d_tmp, sizeof(double) * 2 * nv))) return rc;
  double *din = (double *)c->d_tmp.p, *dout = din + nv;
  if ((rc = copy_h2d(c, din, x, sizeof(double) * nv))) return rc;
  if ((rc = rbl_apply_saddle_dev(c, din, dout))) return rc;
  if ((rc = copy_d2h(c, out, dout, sizeof(double) * nv))) return rc;
  return finish_and_check(c);
}
