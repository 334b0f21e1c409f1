// rbl_roots.hip -- square roots M^{1/2} W: dense Cholesky path, Lanczos (plain / block-Jacobi / two-level preconditioned).
// Part of the implementation of the C ABI in include/rbl.h (split from the former rbl_api.hip along its sections);
// shared internals are declared in rbl_api_internal.hpp.  Nothing here falls back to a CPU path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "rbl_api_internal.hpp"

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

// ---- two-level factor of the preconditioned Lanczos root (round 3) --------------------------------------------------
// G = B L H with H = I + Q (L_E - I) Q^T: the block-Jacobi factor L times a low-rank correction that carries the monopole
// far field between the bodies (rbl_body_dev.hip: k_tl_orth for the algebra).  Any invertible G keeps
// x = G (G^-1 M G^-T)^{1/2} W an exact root; this one moves the collective translations of the bodies -- the modes whose
// Euclidean-norm error converges last under block-Jacobi -- into the factor: 27 bodies of shell_N_162 above a wall need
// 3-4 Lanczos iterations to 1e-3 instead of 6-7, 8-10 instead of 14-17 to 1e-6 (tests/experiments/two_level_root.py).
// Built once per configuration: Z = L^-1 K_t (three vectors through the per-body factors), a 3 N_bod-square sphere tensor,
// its Cholesky factor and explicit inverse -- all replicated on every rank of a multi-GPU context (Z by own bodies + sum).
// Not usable (tl_ok = false: plain block-Jacobi) when I + E is not positive definite or the small system does not fit.
int tl_build(rbl_ctx *c)
{
  if (c->tl_valid && !c->tl_q_stale) return RBL_OK;
  const bool keep_E = c->tl_valid && c->tl_ok;          // a configuration change within RBL_OPT_TWO_LEVEL_REFRESH: only Q is rebuilt
  c->tl_q_stale = false;
  if (!keep_E) { c->tl_valid = true; c->tl_ok = false; c->tl_age = 0; }
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
  if ((rc = rbl_dev_reserve(c, c->d_tlZ, sizeof(double) * 3 * (size_t)n3))) return rc;
  double *t = (double *)c->d_tlZ.p;
  if (comm_on(c)) {
    rbl_launch_tl_unit(c->stream, n3, Q);                                // K_t: one vector per direction holds that column of every body
    int b0, b1; comm_body_range(c, &b0, &b1);
    if (comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(t, 0, sizeof(double) * 3 * (size_t)n3, c->stream));
    if ((rc = blk_solve(c, b0, b1 - b0, Q, t, 3, n3, 1, false))) return rc;
    if ((rc = comm_allgather_bodies(c, t, 0, 3 * (int64_t)S.N_blb, 3, n3))) return rc;      // owners' segments, in place
    RBL_HIP(c, hipMemcpyAsync(Q, t, sizeof(double) * 3 * (size_t)n3, hipMemcpyDeviceToDevice, c->stream));
  } else {                                                               // Z = L^-1 K_t (G^-1 K_t with body-frame factors), out of place
    rbl_launch_tl_unit(c->stream, n3, t);
    if ((rc = blk_solve(c, 0, Nb, t, Q, 3, n3, 1, false))) return rc;
  }
  rbl_launch_tl_orth(c->stream, Q, n3, S.N_blb, Nb, (double *)c->d_tlCb.p, keep_E ? c->d_err : c->d_err2);
  if (keep_E) return RBL_OK;                             // (a failed orthonormalisation now surfaces with the step's error word)
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
int tl_apply(rbl_ctx *c, const double *w, double *wo, int nvec, int64_t pitch, int op)
{
  const RblBodyState &S = c->S;
  const int Nb = S.N_bod;
  const int64_t nt = 3 * (int64_t)Nb, n3 = 3 * (int64_t)Nb * S.N_blb;
  const double *Q = (const double *)c->d_tlQ.p;
  double *t = (double *)c->d_tlT.p;                                       // (up to three vectors at a time)
  // the small operator by its rows, in the launch that adds the correction: L_E itself (op 0), or the explicit inverse in the
  // layout that holds the wanted rows contiguously (rbl_block_inverse_*: first half columns of X, second half its rows)
  const int64_t ldx = rbl_block_inverse_ld(nt);
  const double *Op = op == 0 ? (const double *)c->d_tlA.p : (const double *)c->d_tlX.p + (op == 1 ? (size_t)(ldx * nt) : 0);
  for (int v0 = 0; v0 < nvec; v0 += 3) {
    const int g = nvec - v0 >= 3 ? 3 : nvec - v0;
    const double *wv = w + (size_t)v0 * (size_t)pitch;
    rbl_launch_tl_qt(c->stream, Q, n3, S.N_blb, Nb, wv, pitch, g, t, nt);
    rbl_launch_tl_eaddq(c->stream, Q, n3, S.N_blb, Nb, Op, nt, op == 0 ? nt : ldx, op, t, nt, wv, wo + (size_t)v0 * (size_t)pitch, pitch, g);
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
    if (comm_on(c)) {   // every rank substitutes through ITS bodies' factors only; an all-gather of the owners' segments completes the vectors
      const int64_t mb = 3 * (int64_t)c->S.N_blb;
      int b0, b1; comm_body_range(c, &b0, &b1);
      const double *src = d_x;
      if (tl) {                                        // (d_y is free until the product: H^-T x goes there)
        if ((rc = tl_apply(c, d_x, d_y, nvec, n, 2))) return rc;
        src = d_y;
      }
      if (comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(d_tmp, 0, vbytes, c->stream));
      if ((rc = blk_solve(c, b0, b1 - b0, src, d_tmp, nvec, n, 2, lz32)))
        return rbl_fail(c, rc, "preconditioned square root: the per-body factor application failed");
      if ((rc = comm_allgather_bodies(c, d_tmp, 0, mb, nvec, n))) return rc;
      c->no_damp = true;
      rc = apply_M_multi_enqueue(c, c->S.wall, d_tmp, d_r, nbl, nvec, d_y);
      c->no_damp = false;
      if (rc) return rc;
      if (comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(d_tmp, 0, vbytes, c->stream));
      if ((rc = blk_solve(c, b0, b1 - b0, d_y, d_tmp, nvec, n, 1, lz32))) return rc;
      if ((rc = comm_allgather_bodies(c, d_tmp, 0, mb, nvec, n))) return rc;
      RBL_HIP(c, hipMemcpyAsync(d_y, d_tmp, vbytes, hipMemcpyDeviceToDevice, c->stream));
      return tl ? tl_apply(c, d_y, d_y, nvec, n, 1) : RBL_OK;
    }
    if (tl) {                                          // H^-T x staged in d_y (free until the product), then out of place through L^-T
      if ((rc = tl_apply(c, d_x, d_y, nvec, n, 2))) return rc;
      rc = blk_solve(c, 0, c->S.N_bod, d_y, d_tmp, nvec, n, 2, lz32);
    } else rc = blk_solve(c, 0, c->S.N_bod, d_x, d_tmp, nvec, n, 2, lz32);   // both vectors in one pass over L
    if (rc) return rbl_fail(c, rc, "preconditioned square root: the per-body factor application failed");
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
  const size_t ndot = 2 + 2 * 512;                                     // partial sums of the increment-norm test (4 vectors x RBL_SQNORM_BLOCKS)
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
  rbl_launch_lanczos_init(c->stream, n, d_W, d_wn(0), Vp(0, 0), d_part_init, nvec, (int64_t)nsc);   // both recurrences of a pair in the same launches
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
  bool done = false, out_ready = false;
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
      // All of it in one batch: the 2 nvec vectors (estimates, then corrections) go through H, the body factors and B
      // together, their norms come back in one read; and the estimate that passes IS the result (out_ready).
      int b0 = 0, b1 = c->S.N_bod;
      if (comm_on(c)) comm_body_range(c, &b0, &b1);
      const int nz = 2 * nvec;
      std::vector<double> cf((size_t)2 * m);
      for (int v = 0; v < nvec; ++v) {
        const std::vector<double> &yc = y_cur[v];
        std::vector<double> yp;
        const int mv = m_last[v] > 1 ? m_last[v] : m;
        alpha.assign(hs.data() + nsc * v, hs.data() + nsc * v + m);
        beta.assign(hs.data() + nsc * v + maxit, hs.data() + nsc * v + maxit + m);
        if (wnorm[v] > 0.0 && (rc = lanczos_coeffs(c, alpha, beta, mv - 1, wnorm[v], yp))) return rc;
        for (int p_ = 0; p_ < m; ++p_) { cf[p_] = yc[p_]; cf[m + p_] = yc[p_] - (p_ < (int)yp.size() ? yp[p_] : 0.0); }
        if ((rc = upload_coef(c, d_coef(v), cf.data(), 2 * m, v))) return rc;   // (the stream is idle: hs has just been read)
      }
      rbl_launch_lanczos_combine_xd(c->stream, n, V, (int64_t)nvec * n, n, d_coef(0), (int64_t)nsc, m, u, tmp, n, nvec);   // u | tmp: x_0.. d_0..
      if (c->tl_ok && (rc = tl_apply(c, u, u, nz, n, 0))) return rc;
      if (comm_on(c) && comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(ex, 0, sizeof(double) * (size_t)n * nz, c->stream));
      if ((rc = blk_trmv_multi(c, b0, b1 - b0, u, ex, nz, n))) return rc;
      if (comm_on(c) && (rc = comm_allgather_bodies(c, ex, 0, 3 * (int64_t)c->S.N_blb, nz, n))) return rc;
      const int gq = rbl_launch_damp_sqnorm(c->stream, P, d_r, nbl, ex, n, nz, d_dot);
      std::vector<double> hq((size_t)gq * nz);
      if ((rc = read_back(c, hq.data(), d_dot, sizeof(double) * hq.size()))) return rc;
      auto sq = [&](int k) { double t = 0.0; for (int g_ = 0; g_ < gq; ++g_) t += hq[(size_t)k * gq + g_]; return t; };
      double worst = 0.0;
      for (int v = 0; v < nvec; ++v) {
        if (!(wnorm[v] > 0.0)) continue;
        const double hx = sq(v), h2 = sq(nvec + v);
        const double dout = hx > 0.0 ? std::sqrt(h2 / hx) : 0.0;
        // rho of the coefficient sequence (same contraction, other norm); resid[v] = d_m rho / (1 - rho) in the energy norm
        const double ratio = d_last[v] > 0.0 ? resid[v] / d_last[v] : 1.0;
        const double est = dout * ratio;
        if (trace) std::fprintf(stderr, "rbl lanczos (pc): vector %d  m = %d  energy-norm estimate %.3e  increment-norm change %.3e  estimate %.3e\n", v, m, resid[v], dout, est);
        resid[v] = est;
        worst = std::max(worst, est);
      }
      out_ready = true;                                    // ex[0..nvec) = B G V y_m: the increments themselves if the test passes
      if (!(worst < c->lanczos_tol) && m < maxit) { all_conv = false; out_ready = false; }
    }
    if (all_conv) done = true;
  }
  c->lanczos_iters = m;
  c->lanczos_resid = *std::max_element(resid.begin(), resid.end());
  if (out_ready) {                                         // the last test evaluated exactly these vectors
    RBL_HIP(c, hipMemcpyAsync(d_out, ex, vbytes * (size_t)nvec, hipMemcpyDeviceToDevice, c->stream));
    return RBL_OK;
  }
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
      if (comm_on(c) && comm_gather_needs_zero(c)) RBL_HIP(c, hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)n, c->stream));
      if ((rc = blk_trmv(c, b0, b1 - b0, o, tmp))) return rc;
      if (comm_on(c) && (rc = comm_allgather_bodies(c, tmp, 0, 3 * (int64_t)c->S.N_blb, 1, n))) return rc;
      rbl_launch_scale_by_damp(c->stream, P, d_r, nbl, tmp, o);
    }
  }
  return RBL_OK;
}

// nvec noise vectors (columns of d_W, stride n) -> nvec increments.  The dense factorisation is done ONCE
// for all of them (the reference calls M_half_W() once per vector, :927-936, and refactors each time).
int mhalf_dev_multi(rbl_ctx *c, const double *d_r, int64_t nbl, const double *d_W, int nvec, int method,
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
