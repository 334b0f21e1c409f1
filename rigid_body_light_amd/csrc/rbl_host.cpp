// rbl_host.cpp -- host-side O(N_bod)/O(N) rigid-body bookkeeping of CManyBodies:
// quaternion -> rotation, K / K^T / K^-1 (reference c_rigid_obj.cpp:302-410), the
// preconditioner algebra (:489-616) and the quaternion update (:679-710).  Plain C++
// (no Eigen in this image).  None of this is the north-star hot path (SURVEY.md
// section 8: rows N1/N2 "next"); it exists so the drop-in surface is complete.
#include <cmath>
#include <cstring>

#include "rbl_internal.hpp"

void rbl_quat_to_rot(const double *q, double *R)
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

static bool inv3(const double *D, double *S, double *det_out)
{
  const double c00 = D[4] * D[8] - D[5] * D[7];
  const double c01 = D[5] * D[6] - D[3] * D[8];
  const double c02 = D[3] * D[7] - D[4] * D[6];
  const double det = D[0] * c00 + D[1] * c01 + D[2] * c02;
  if (det_out) *det_out = det;
  if (det == 0.0) return false;
  const double id = 1.0 / det;
  S[0] = c00 * id; S[1] = (D[2] * D[7] - D[1] * D[8]) * id; S[2] = (D[1] * D[5] - D[2] * D[4]) * id;
  S[3] = c01 * id; S[4] = (D[0] * D[8] - D[2] * D[6]) * id; S[5] = (D[2] * D[3] - D[0] * D[5]) * id;
  S[6] = c02 * id; S[7] = (D[1] * D[6] - D[0] * D[7]) * id; S[8] = (D[0] * D[4] - D[1] * D[3]) * id;
  return true;
}

// set_K_mats / Make_K_Kinv, c_rigid_obj.cpp:328-402.  K is kept in factored form:
// lever arms r_k = R(Q_b) c_k  (:374) and the 6x6 blocks of (K^T K)^-1 (:302-326).
int rbl_body_set_K(RblBodyState &S, std::string &err)
{
  const int nb = S.N_bod, nl = S.N_blb;
  S.lever.assign((size_t)3 * nb * nl, 0.0);
  S.KTKinv.assign((size_t)36 * nb, 0.0);
  double sumr2 = 0.0, MOI[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < nl; ++k) {
    const double *c = &S.ref_cfg[3 * k];
    sumr2 += c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) MOI[3 * p + q] += c[p] * c[q];
  }
  for (int b = 0; b < nb; ++b) {
    double R[9];
    rbl_quat_to_rot(&S.Q[4 * b], R);
    for (int k = 0; k < nl; ++k) {
      const double *c = &S.ref_cfg[3 * k];
      double *l = &S.lever[3 * ((size_t)b * nl + k)];
      l[0] = c[0] * R[0] + c[1] * R[1] + c[2] * R[2];
      l[1] = c[0] * R[3] + c[1] * R[4] + c[2] * R[5];
      l[2] = c[0] * R[6] + c[1] * R[7] + c[2] * R[8];
    }
    // D = sumr2 I - R MOI R^T  (:309-310)
    double T[9], D[9];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q)
        T[3 * p + q] = R[3 * p] * MOI[q] + R[3 * p + 1] * MOI[3 + q] + R[3 * p + 2] * MOI[6 + q];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q)
        D[3 * p + q] = (p == q ? sumr2 : 0.0) -
                       (T[3 * p] * R[3 * q] + T[3 * p + 1] * R[3 * q + 1] + T[3 * p + 2] * R[3 * q + 2]);
    double Sm[9], det;
    const bool ok = inv3(D, Sm, &det);
    if (!ok || det < 1.0e-13) {  // :312-316 (the reference exit()s)
      err = "K^T*K is singular (is your rigid body a dimer?)";
      return RBL_ERR_SINGULAR;
    }
    double *B = &S.KTKinv[(size_t)36 * b];
    for (int p = 0; p < 3; ++p) B[6 * p + p] = 1.0 / (1.0 * nl);  // Ainv (:306)
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) B[6 * (3 + p) + 3 + q] = Sm[3 * p + q];
  }
  S.K_set = true;
  return RBL_OK;
}

// K U : u_k = U_b + Omega_b x r_k   (:368-383, :404)
void rbl_body_K_x_U(const RblBodyState &S, const double *U, double *out)
{
  const int nb = S.N_bod, nl = S.N_blb;
  for (int b = 0; b < nb; ++b) {
    const double *u = U + 6 * b, *om = U + 6 * b + 3;
    for (int k = 0; k < nl; ++k) {
      const size_t idx = 3 * ((size_t)b * nl + k);
      const double *l = &S.lever[idx];
      out[idx] = u[0] + l[2] * om[1] - l[1] * om[2];
      out[idx + 1] = u[1] + l[0] * om[2] - l[2] * om[0];
      out[idx + 2] = u[2] + l[1] * om[0] - l[0] * om[1];
    }
  }
}

// K^T lambda : F_b = sum lambda_k, T_b = sum r_k x lambda_k   (:410)
void rbl_body_KT_x_Lam(const RblBodyState &S, const double *lam, double *out)
{
  const int nb = S.N_bod, nl = S.N_blb;
  for (int b = 0; b < nb; ++b) {
    double f[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < nl; ++k) {
      const size_t idx = 3 * ((size_t)b * nl + k);
      const double *l = &S.lever[idx];
      const double *v = lam + idx;
      f[0] += v[0]; f[1] += v[1]; f[2] += v[2];
      f[3] += l[1] * v[2] - l[2] * v[1];
      f[4] += l[2] * v[0] - l[0] * v[2];
      f[5] += l[0] * v[1] - l[1] * v[0];
    }
    for (int c = 0; c < 6; ++c) out[6 * b + c] = f[c];
  }
}

// Kinv V = (K^T K)^-1 K^T V   (:390, :406)
void rbl_body_Kinv_x_V(const RblBodyState &S, const double *V, double *out)
{
  const int nb = S.N_bod;
  std::vector<double> t((size_t)6 * nb);
  rbl_body_KT_x_Lam(S, V, t.data());
  for (int b = 0; b < nb; ++b) {
    const double *B = &S.KTKinv[(size_t)36 * b];
    for (int p = 0; p < 6; ++p) {
      double s = 0.0;
      for (int q = 0; q < 6; ++q) s += B[6 * p + q] * t[6 * b + q];
      out[6 * b + p] = s;
    }
  }
}

// Kinv^T F = K (K^T K)^-T F   (:408)
void rbl_body_KTinv_x_F(const RblBodyState &S, const double *F, double *out)
{
  const int nb = S.N_bod;
  std::vector<double> t((size_t)6 * nb);
  for (int b = 0; b < nb; ++b) {
    const double *B = &S.KTKinv[(size_t)36 * b];
    for (int p = 0; p < 6; ++p) {
      double s = 0.0;
      for (int q = 0; q < 6; ++q) s += B[6 * q + p] * F[6 * b + q];
      t[6 * b + p] = s;
    }
  }
  rbl_body_K_x_U(S, t.data(), out);
}

// update_X_Q with Q_from_Om, c_rigid_obj.cpp:679-710.  U has displacement units.
void rbl_body_update_X_Q(const RblBodyState &S, const double *U, std::vector<double> &Xo,
                         std::vector<double> &Qo)
{
  const int nb = S.N_bod;
  Xo = S.X;
  Qo = S.Q;
  for (int b = 0; b < nb; ++b) {
    const double *om = U + 6 * b + 3;
    const double nrm = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double qw = std::cos(nrm / 2.0), qx = 0.0, qy = 0.0, qz = 0.0;  // :681-683
    if (nrm > 1.0e-10) {                                           // :684-686
      const double s = std::sin(nrm / 2.0) / nrm;
      qx = s * om[0]; qy = s * om[1]; qz = s * om[2];
    }
    double qn = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);   // :687
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    const double *o = &S.Q[4 * b];  // (w,x,y,z)
    double rw = qw * o[0] - qx * o[1] - qy * o[2] - qz * o[3];      // Q_rot * Q  (:704)
    double rx = qw * o[1] + qx * o[0] + qy * o[3] - qz * o[2];
    double ry = qw * o[2] + qy * o[0] + qz * o[1] - qx * o[3];
    double rz = qw * o[3] + qz * o[0] + qx * o[2] - qy * o[1];
    qn = std::sqrt(rw * rw + rx * rx + ry * ry + rz * rz);          // :705
    Qo[4 * b] = rw / qn; Qo[4 * b + 1] = rx / qn; Qo[4 * b + 2] = ry / qn; Qo[4 * b + 3] = rz / qn;
    for (int c = 0; c < 3; ++c) Xo[3 * b + c] = S.X[3 * b + c] + U[6 * b + c];  // :706
  }
}

int rbl_chol6(double *A)
{
  for (int j = 0; j < 6; ++j) {
    double d = A[6 * j + j];
    for (int k = 0; k < j; ++k) d -= A[6 * j + k] * A[6 * j + k];
    if (!(d > 0.0)) return RBL_ERR_NOT_SPD;
    d = std::sqrt(d);
    A[6 * j + j] = d;
    for (int i = j + 1; i < 6; ++i) {
      double s = A[6 * i + j];
      for (int k = 0; k < j; ++k) s -= A[6 * i + k] * A[6 * j + k];
      A[6 * i + j] = s / d;
    }
    for (int i = 0; i < j; ++i) A[6 * i + j] = 0.0;
  }
  return RBL_OK;
}

// invM * v for the cached preconditioner mobility inverse
static void apply_invM(const RblBodyState &S, const double *v, double *out)
{
  // diagonal preconditioner only: the block-diagonal one lives on the GPU (rbl_bodies.hip)
  const int nb = S.N_bod, nl = S.N_blb;
  for (size_t i = 0; i < (size_t)nb * nl; ++i) {
    const double *B = &S.invM_diag[9 * i];
    const double *x = v + 3 * i;
    out[3 * i] = B[0] * x[0] + B[1] * x[1] + B[2] * x[2];
    out[3 * i + 1] = B[3] * x[0] + B[4] * x[1] + B[5] * x[2];
    out[3 * i + 2] = B[6] * x[0] + B[7] * x[1] + B[8] * x[2];
  }
}

// apply_PC, c_rigid_obj.cpp:589-616, diagonal PC.  The caller (rbl_bodies.hip) fills invM_diag
// beforehand when !pc_set (diag_invM :489-543).
int rbl_body_apply_PC(RblBodyState &S, const double *in, double *out, std::string &err)
{
  const int nb = S.N_bod, nl = S.N_blb;
  const size_t n = (size_t)3 * nb * nl;
  if (!S.pc_set) {
    // Ninv = K^T invM K, 6x6 per body (:593-594), then its Cholesky (:554-567)
    S.Ninv_chol.assign((size_t)36 * nb, 0.0);
    std::vector<double> col(n), ic(n), kt((size_t)6 * nb), e((size_t)6 * nb, 0.0);
    for (int c = 0; c < 6; ++c) {
      std::fill(e.begin(), e.end(), 0.0);
      for (int b = 0; b < nb; ++b) e[6 * b + c] = 1.0;   // bodies do not couple in K
      rbl_body_K_x_U(S, e.data(), col.data());
      apply_invM(S, col.data(), ic.data());
      rbl_body_KT_x_Lam(S, ic.data(), kt.data());
      for (int b = 0; b < nb; ++b)
        for (int p = 0; p < 6; ++p) S.Ninv_chol[(size_t)36 * b + 6 * p + c] = kt[6 * b + p];
    }
    for (int b = 0; b < nb; ++b) {
      if (rbl_chol6(&S.Ninv_chol[(size_t)36 * b])) {
        err = "preconditioner block K^T invM K is not positive definite";
        return RBL_ERR_NOT_SPD;
      }
    }
    S.pc_set = true;
  }
  const double *slip = in, *F = in + n;
  std::vector<double> t(n), rhs((size_t)6 * nb), U((size_t)6 * nb), ku(n);
  apply_invM(S, slip, t.data());
  rbl_body_KT_x_Lam(S, t.data(), rhs.data());
  for (size_t i = 0; i < (size_t)6 * nb; ++i) rhs[i] = -F[i] - rhs[i];      // :601
  for (int b = 0; b < nb; ++b) {                                          // :605-608 LLT solve
    const double *L = &S.Ninv_chol[(size_t)36 * b];
    double y[6];
    for (int p = 0; p < 6; ++p) {
      double s = rhs[6 * b + p];
      for (int q = 0; q < p; ++q) s -= L[6 * p + q] * y[q];
      y[p] = s / L[6 * p + p];
    }
    for (int p = 5; p >= 0; --p) {
      double s = y[p];
      for (int q = p + 1; q < 6; ++q) s -= L[6 * q + p] * U[6 * b + q];
      U[6 * b + p] = s / L[6 * p + p];
    }
  }
  rbl_body_K_x_U(S, U.data(), ku.data());
  for (size_t i = 0; i < n; ++i) ku[i] += slip[i];
  apply_invM(S, ku.data(), out);                                          // :610
  for (size_t i = 0; i < n; ++i) out[i] *= S.M_scale;
  std::memcpy(out + n, U.data(), sizeof(double) * 6 * nb);
  return RBL_OK;
}
