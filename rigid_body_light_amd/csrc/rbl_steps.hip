// rbl_steps.hip -- whole time steps and the random-finite-difference family (reference C++-only members).
// Part of the implementation of the C ABI in include/rbl.h (split from the former rbl_api.hip along its sections);
// shared internals are declared in rbl_api_internal.hpp.  Nothing here falls back to a CPU path.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "rbl_api_internal.hpp"

// ---- whole time steps in one call (what krylov.py's steppers do, for hosts without a Python driver) ----------

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

// ---- random finite differences (reference C++-only members) ---------------

// d_out = (1/delta)[M(q + delta/2 dq) - M(q - delta/2 dq)] W for a displacement direction dq[6 N_bod] (host): the shared core of
// M_RFD (:776-794, dq = Kinv W) and M_RFD_from_U (:820-842, dq = the caller's U).  d_r: n3 scratch, d_work: 2 n3 scratch.
static int m_rfd_dir(rbl_ctx *c, const double *d_W, const double *dq, double delta, double *d_out, double *d_r, double *d_work)
{
  RblPhase ph_total(c, RBL_T_TOTAL);
  RblBodyState &S = c->S;
  const int64_t N = (int64_t)S.N_bod * S.N_blb, n3 = 3 * N;
  std::vector<double> win((size_t)6 * S.N_bod), Xs, Qs;
  const std::vector<double> X0 = S.X, Q0 = S.Q;
  double *dM[2] = {d_work, d_work + n3};
  int rc = RBL_OK;
  for (int sgn = 0; sgn < 2; ++sgn) {                                 // q +- delta/2 dq (:783-788)
    const double f = (sgn == 0 ? 0.5 : -0.5) * delta;
    for (size_t i = 0; i < win.size(); ++i) win[i] = f * dq[i];
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

// core of M_RFD(), c_rigid_obj.cpp:776-794: dq = Kinv W.  Wh = host copy of W (Kinv is O(N) host work)
int m_rfd_core(rbl_ctx *c, const double *d_W, const double *Wh, double delta, double *d_out, double *d_r, double *d_work)
{
  std::vector<double> uom((size_t)6 * c->S.N_bod);
  rbl_body_Kinv_x_V(c->S, Wh, uom.data());                            // UOM = Kinv W (:776)
  return m_rfd_dir(c, d_W, uom.data(), delta, d_out, d_r, d_work);
}

// uom_v = Kinv V_v = (K^T K)^-1 K^T V_v for nv device vectors: the sums over the blobs on the device (one launch each), the
// 6 x 6 blocks on the host after ONE small read-back -- instead of bringing the 3 N-vectors to the host (reference :408)
static int kinv_dev(rbl_ctx *c, const double *const *d_V, int nv, double *d_t, std::vector<double> *uom)
{
  int rc = sync_bodies(c); if (rc) return rc;
  const RblBodyState &S = c->S;
  const size_t nb6 = (size_t)6 * S.N_bod;
  for (int v = 0; v < nv; ++v) rbl_launch_KT_x_Lam(c->stream, (const double *)c->d_lever.p, d_V[v], S.N_blb, S.N_bod, d_t + (size_t)v * nb6);
  std::vector<double> t(nb6 * (size_t)nv);
  if ((rc = read_back(c, t.data(), d_t, sizeof(double) * t.size()))) return rc;
  for (int v = 0; v < nv; ++v) {
    uom[v].assign(nb6, 0.0);
    for (int b = 0; b < S.N_bod; ++b) {
      const double *B = &S.KTKinv[(size_t)36 * b], *tb = t.data() + (size_t)v * nb6 + 6 * (size_t)b;
      for (int p = 0; p < 6; ++p) {
        double s = 0.0;
        for (int q = 0; q < 6; ++q) s += B[6 * p + q] * tb[q];
        uom[v][6 * (size_t)b + p] = s;
      }
    }
  }
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
  if ((rc = rbl_dev_reserve(c, c->d_bd, 9 * vb + 2 * sizeof(double) * (size_t)nb6))) return rc;
  double *base = (double *)c->d_bd.p;
  double *dWown = base, *dMW = base + 3 * n3 /* 2 vectors */, *dRFD = base + 5 * n3, *dr = base + 6 * n3,
         *dwork = base + 7 * n3, *dt6 = base + 9 * n3;
  if (!d_W) {                                                                          // rand_vector (:730-741)
    rbl_launch_normal(c->stream, seed, 0, 3 * n3, dWown);
    d_W = dWown;
  }
  const int nvec = split_rand ? 2 : 1;
  if ((rc = positions_dev(c, 0, S.N_bod, dr))) return rc;                              // multi_body_pos (:662)
  if ((rc = mhalf_dev_multi(c, dr, N, d_W, nvec, method, dMW))) return rc;             // M_half_W1/2 (:927-936)
  // Kinv of the RFD noise (M_RFD's direction, :776) and of M^{1/2}W1 (the predictor, :955): blob sums on the device, one read-back
  const double *kv[2] = {d_W + 2 * n3, dMW};
  std::vector<double> uoms[2];
  if ((rc = kinv_dev(c, kv, 2, dt6, uoms))) return rc;
  if ((rc = m_rfd_dir(c, d_W + 2 * n3, uoms[0].data(), delta, dRFD, dr, dwork))) return rc;  // M_RFD (:940)
  const double c1 = split_rand ? 2.0 * std::sqrt(S.kBT / S.dt) : std::sqrt(2.0 * S.kBT / S.dt);   // :945-952
  const double c2 = split_rand ? std::sqrt(S.kBT / S.dt) : std::sqrt(2.0 * S.kBT / S.dt);
  // Slip -= kBT M_RFD + BI,  BI = c2 (M^{1/2}W1 - M^{1/2}W2)  or  c2 M^{1/2}W1   (:948,953,963)
  rbl_launch_rhs_combine(c->stream, n3, d_Slip, S.kBT, dRFD, c2, dMW, split_rand ? dMW + n3 : nullptr, d_RHS);
  // predictor: q^{n+1/2} = q^n displaced by (dt/2) Kinv (c1 M^{1/2}W1)   (:955-959)
  std::vector<double> &uom = uoms[1], Xo, Qo;
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

// M_RFD_from_U(U, W), c_rigid_obj.cpp:820-842: the same random finite difference along the caller's displacement U[6 N_bod]
int rbl_M_RFD_from_U(rbl_ctx *c, const double *U, const double *W, double delta, double *out)
{
  int rc = need_config(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!U || !W || !out || !(delta > 0.0)) return rbl_fail(c, RBL_ERR_ARG, "M_RFD_from_U: need U, W, out and delta > 0");
  const int64_t n3 = (int64_t)3 * c->S.N_bod * c->S.N_blb;
  const size_t vb = sizeof(double) * (size_t)n3;
  if ((rc = rbl_dev_reserve(c, c->d_W, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_r, vb))) return rc;
  if ((rc = rbl_dev_reserve(c, c->d_U, 2 * vb))) return rc;
  if ((rc = copy_h2d(c, c->d_W.p, W, vb))) return rc;
  double *dU = (double *)c->d_U.p;
  if ((rc = m_rfd_dir(c, (const double *)c->d_W.p, U, delta, dU, (double *)c->d_r.p, dU))) return rc;
  if ((rc = copy_d2h(c, out, dU, vb))) return rc;
  return finish_and_check(c);
}

// M_RFD_cfgs(U, delta), c_rigid_obj.cpp:798-818: blob positions at q +- (delta/2) U
int rbl_M_RFD_cfgs(rbl_ctx *c, const double *U, double delta, double *r_plus, double *r_minus)
{
  int rc = need_config(c); if (rc) return rc;
  if ((rc = rbl_dev_init(c))) return rc;
  if (!U || !r_plus || !r_minus) return rbl_fail(c, RBL_ERR_ARG, "M_RFD_cfgs: null argument");
  RblBodyState &S = c->S;
  const size_t vb = sizeof(double) * 3 * (size_t)S.N_bod * S.N_blb;
  if ((rc = rbl_dev_reserve(c, c->d_r, vb))) return rc;
  std::vector<double> win((size_t)6 * S.N_bod), Xs, Qs;
  const std::vector<double> X0 = S.X, Q0 = S.Q;
  for (int sgn = 0; sgn < 2; ++sgn) {
    const double f = (sgn == 0 ? 0.5 : -0.5) * delta;                 // :808, :811
    for (size_t i = 0; i < win.size(); ++i) win[i] = f * U[i];
    rbl_body_update_X_Q(S, win.data(), Xs, Qs);
    S.X = Xs; S.Q = Qs; c->dev_xq_valid = false;
    rc = positions_dev(c, 0, S.N_bod, (double *)c->d_r.p);
    if (!rc) rc = copy_d2h(c, sgn == 0 ? r_plus : r_minus, c->d_r.p, vb);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = RBL_ERR_HIP;
    S.X = X0; S.Q = Q0; c->dev_xq_valid = false;
    if (rc) return rc;
  }
  return RBL_OK;
}

// KT_RFD_from_U(U, W), c_rigid_obj.cpp:844-863: (1/delta) [K(q+)^T - K(q-)^T] W, W[3N] -> out[6 N_bod] (O(N) host work)
int rbl_KT_RFD_from_U(rbl_ctx *c, const double *U, const double *W, double delta, double *out)
{
  int rc = need_K(c); if (rc) return rc;
  if (!U || !W || !out || !(delta > 0.0)) return rbl_fail(c, RBL_ERR_ARG, "KT_RFD_from_U: need U, W, out and delta > 0");
  const RblBodyState &S = c->S;
  const size_t nb6 = (size_t)6 * S.N_bod;
  std::vector<double> win(nb6), tmp(nb6);
  for (size_t i = 0; i < nb6; ++i) out[i] = 0.0;
  for (int sgn = 0; sgn < 2; ++sgn) {
    const double f = (sgn == 0 ? 0.5 : -0.5) * delta;
    for (size_t i = 0; i < nb6; ++i) win[i] = f * U[i];
    RblBodyState T = S;                                              // displaced copy (:853-859)
    rbl_body_update_X_Q(S, win.data(), T.X, T.Q);
    if ((rc = rbl_body_set_K(T, c->last_error))) return rc;
    rbl_body_KT_x_Lam(T, W, tmp.data());
    const double w = (sgn == 0 ? 1.0 : -1.0) / delta;
    for (size_t i = 0; i < nb6; ++i) out[i] += w * tmp[i];           // :861
  }
  return RBL_OK;
}

// evolve_X_Q_RFD(U), c_rigid_obj.cpp:880-893: commit q displaced by U (displacement units), rebuild K, KEEP the
// preconditioner (:892 PC_mat_Set = true): the factors of q go on serving q + U, whose size is an RFD delta.
int rbl_evolve_X_Q_RFD(rbl_ctx *c, const double *U)
{
  int rc = need_config(c); if (rc) return rc;
  if (!U) return rbl_fail(c, RBL_ERR_ARG, "evolve_X_Q_RFD: U is NULL");
  RblBodyState &S = c->S;
  std::vector<double> Xo, Qo;
  rbl_body_update_X_Q(S, U, Xo, Qo);                                  // :886 (no dt)
  S.X.swap(Xo);
  S.Q.swap(Qo);
  c->dev_bodies_valid = false; c->dev_xq_valid = false;
  c->pc_keep_once = c->dev_pc_valid;                                  // the next re-synchronisation leaves the device preconditioner alone
  return rbl_body_set_K(S, c->last_error);                            // :891; S.pc_set is left as it is (:892)
}
