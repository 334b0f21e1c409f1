// rbl_body_dev.hip -- device versions of the per-body geometric operators around the hot path
// (SURVEY.md section 8f rows N1/N2): K U, K^T lambda (reference c_rigid_obj.cpp:328-410), the
// diagonal preconditioner apply_PC (:489-543, :589-616) and the saddle-operator epilogue
// (src/Rigid.py:73-80).  They exist so that a Krylov iteration needs no host round trip:
// everything a GMRES step touches stays in HBM.  All are O(N) and latency-bound; one
// workgroup per rigid body, deterministic LDS tree reductions (no atomics).
#include "rbl_internal.hpp"

namespace {

constexpr int BT = 256;

__device__ __forceinline__ void quat_rot(const double *q, double *R)
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

// lever arms l_k = R(Q_b) c_k (:374) and positions r_k = l_k + X_b (:257-265)
__global__ void k_body_geom(const double *__restrict__ X, const double *__restrict__ Q,
                            const double *__restrict__ cfg, int N_blb, long N,
                            double *__restrict__ lever, double *__restrict__ pos)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N) return;
  const int b = (int)(idx / N_blb), k = (int)(idx % N_blb);
  double R[9];
  quat_rot(Q + 4 * b, R);
  const double c0 = cfg[3 * k], c1 = cfg[3 * k + 1], c2 = cfg[3 * k + 2];
  {
#pragma clang fp contract(off)
    const double l0 = c0 * R[0] + c1 * R[1] + c2 * R[2];
    const double l1 = c0 * R[3] + c1 * R[4] + c2 * R[5];
    const double l2 = c0 * R[6] + c1 * R[7] + c2 * R[8];
    lever[3 * idx] = l0; lever[3 * idx + 1] = l1; lever[3 * idx + 2] = l2;
    pos[3 * idx] = l0 + X[3 * b]; pos[3 * idx + 1] = l1 + X[3 * b + 1]; pos[3 * idx + 2] = l2 + X[3 * b + 2];
  }
}

// out_k = U_b + Omega_b x l_k     (:368-383, :404)
__global__ void k_K_x_U(const double *__restrict__ lever, const double *__restrict__ U, int N_blb,
                        long N, double *__restrict__ out, const double *__restrict__ sub, double alpha)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N) return;
  const int b = (int)(idx / N_blb);
  const double *u = U + 6 * b, *om = u + 3, *l = lever + 3 * idx;
  const double k0 = u[0] + l[2] * om[1] - l[1] * om[2];
  const double k1 = u[1] + l[0] * om[2] - l[2] * om[0];
  const double k2 = u[2] + l[1] * om[0] - l[0] * om[1];
  if (sub) {  // out = sub + alpha * K U   (saddle epilogue: slip = M lambda - K U)
    out[3 * idx] = sub[3 * idx] + alpha * k0; out[3 * idx + 1] = sub[3 * idx + 1] + alpha * k1;
    out[3 * idx + 2] = sub[3 * idx + 2] + alpha * k2;
  } else {
    out[3 * idx] = k0; out[3 * idx + 1] = k1; out[3 * idx + 2] = k2;
  }
}

template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double (*s)[BT], int t)
{
#pragma unroll
  for (int q = 0; q < NV; ++q) s[q][t] = v[q];
  __syncthreads();
  for (int st = BT / 2; st > 0; st >>= 1) {
    if (t < st) {
#pragma unroll
      for (int q = 0; q < NV; ++q) s[q][t] += s[q][t + st];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = s[q][0];
  __syncthreads();
}

// RblNormFold: |w| and 1 / |w| (0 for a vanishing vector, like k_lz_c) from the partial sums of |w|^2: every wave of every workgroup
// adds them in the same order (lanes strided, 16-lane rows by DPP, the four rows by scalar reads) -- one value everywhere, no barrier
__device__ __forceinline__ void bf_fold_norm(const RblNormFold &nf, int lane, double &nrm, double &inv)
{
  double a = 0.0;
  for (int i = lane; i < nf.np; i += 64) a += nf.part[i];
  nrm = sqrt(rbl_wave_sum64(a));                     // (the ONE order every wave adds in: rbl_internal.hpp)
  inv = nrm > 1e-300 ? 1.0 / nrm : 0.0;
}

// F_b = sum lambda_k, T_b = sum l_k x lambda_k   (:410); one workgroup per body
__global__ __launch_bounds__(BT) void k_KT_x_Lam(const double *__restrict__ lever,
                                                 const double *__restrict__ lam, int N_blb,
                                                 double *__restrict__ out)
{
  __shared__ double s[6][BT];
  const int b = blockIdx.x, t = threadIdx.x;
  double f[6] = {0, 0, 0, 0, 0, 0};
  for (int k = t; k < N_blb; k += BT) {
    const size_t idx = 3 * ((size_t)b * N_blb + k);
    const double *l = lever + idx, *v = lam + idx;
    f[0] += v[0]; f[1] += v[1]; f[2] += v[2];
    f[3] += l[1] * v[2] - l[2] * v[1];
    f[4] += l[2] * v[0] - l[0] * v[2];
    f[5] += l[0] * v[1] - l[1] * v[0];
  }
  block_reduce<6>(f, s, t);
  if (t < 6) out[6 * b + t] = f[t];
}

// diag_invM (:489-543) + Ninv = K^T invM K per body and its Cholesky (:593-594, :554-567)
template <bool WALL>
__global__ __launch_bounds__(BT) void k_pc_diag_build(const double *__restrict__ lever,
                                                      const double *__restrict__ pos, int N_blb,
                                                      RblParams P, double *__restrict__ invM2,
                                                      double *__restrict__ NL, unsigned *err)
{
  __shared__ double s[21][BT];
  const int b = blockIdx.x, t = threadIdx.x;
  const double scale = 1.0 / P.nf;  // 8 pi eta a
  double acc[21];
#pragma unroll
  for (int q = 0; q < 21; ++q) acc[q] = 0.0;
  for (int k = t; k < N_blb; k += BT) {
    const size_t i = (size_t)b * N_blb + k;
    double dxx = 4.0 / 3.0, dzz = 4.0 / 3.0;
    if (WALL) {  // self wall term (:98-104), h = z/a
      const double h = pos[3 * i + 2] / P.a;
      if (h < 0.0) atomicOr(err, (unsigned)RBL_FLAG_BELOW_WALL);
      const double iz = 1.0 / h, iz3 = iz * iz * iz, iz5 = iz3 * iz * iz;
      dxx += -(9 * iz - 2 * iz3 + iz5) / 12.0;
      dzz += -(9 * iz - 4 * iz3 + iz5) / 6.0;
    }
    const double px = scale / dxx, pz = scale / dzz;  // invM = diag(px, px, pz)
    invM2[2 * i] = px; invM2[2 * i + 1] = pz;
    const double lx = lever[3 * i], ly = lever[3 * i + 1], lz = lever[3 * i + 2];
    // K_k = [I | G], G = [[0, lz, -ly], [-lz, 0, lx], [ly, -lx, 0]];  D = diag(px,px,pz)
    const double Kk[3][6] = {{1, 0, 0, 0, lz, -ly}, {0, 1, 0, -lz, 0, lx}, {0, 0, 1, ly, -lx, 0}};
    const double D[3] = {px, px, pz};
    int q = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        acc[q] += Kk[0][r] * D[0] * Kk[0][c] + Kk[1][r] * D[1] * Kk[1][c] + Kk[2][r] * D[2] * Kk[2][c];
        ++q;
      }
  }
  block_reduce<21>(acc, s, t);
  if (t == 0) {  // 6x6 lower Cholesky, row-major L
    double L[36];
    int q = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c <= r; ++c) { L[6 * r + c] = acc[q]; if (c < r) L[6 * c + r] = 0.0; ++q; }
    bool ok = true;
    for (int j = 0; j < 6; ++j) {
      double d = L[6 * j + j];
      for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
      if (!(d > 0.0)) ok = false;
      d = sqrt(d);
      L[6 * j + j] = d;
      for (int i = j + 1; i < 6; ++i) {
        double v = L[6 * i + j];
        for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k];
        L[6 * i + j] = v / d;
      }
    }
    if (!ok) atomicOr(err, (unsigned)RBL_FLAG_NOT_SPD);
    for (int e = 0; e < 36; ++e) NL[36 * (size_t)b + e] = L[e];
  }
}

// apply_PC (:589-616) with the diagonal invM: one workgroup per body
__global__ __launch_bounds__(BT) void k_pc_diag_apply(const double *__restrict__ lever,
                                                      const double *__restrict__ invM2,
                                                      const double *__restrict__ NL, int N_blb,
                                                      long n3, const double *__restrict__ in,
                                                      double *__restrict__ out, double fsign, RblNormFold nf)
{
  __shared__ double s[6][BT];
  __shared__ double Ush[6];
  const int b = blockIdx.x, t = threadIdx.x;
  const double *slip = in, *F = in + n3;
  double nrm = 1.0, inv = 1.0;                           // RblNormFold: `in` is an un-normalised Arnoldi vector
  if (nf.part) {
    bf_fold_norm(nf, t & 63, nrm, inv);
    if (t < 6) nf.vnext[n3 + 6 * (size_t)b + t] = inv * F[6 * (size_t)b + t];
    if (t == 0 && b == 0) *nf.hout = nrm;
  }
  double f[6] = {0, 0, 0, 0, 0, 0};
  for (int k = t; k < N_blb; k += BT) {  // K^T (invM slip)
    const size_t i = (size_t)b * N_blb + k;
    const double *l = lever + 3 * i;
    const double v0 = invM2[2 * i] * (inv * slip[3 * i]), v1 = invM2[2 * i] * (inv * slip[3 * i + 1]),
                 v2 = invM2[2 * i + 1] * (inv * slip[3 * i + 2]);
    f[0] += v0; f[1] += v1; f[2] += v2;
    f[3] += l[1] * v2 - l[2] * v1;
    f[4] += l[2] * v0 - l[0] * v2;
    f[5] += l[0] * v1 - l[1] * v0;
  }
  block_reduce<6>(f, s, t);
  if (t == 0) {  // U = Ninv^-1 (-F - K^T invM slip) through the 6x6 Cholesky factor (:601-608)
    const double *L = NL + 36 * (size_t)b;
    double y[6], u[6];
    for (int p = 0; p < 6; ++p) {
      double v = fsign * (inv * F[6 * b + p]) - f[p];
      for (int q = 0; q < p; ++q) v -= L[6 * p + q] * y[q];
      y[p] = v / L[6 * p + p];
    }
    for (int p = 5; p >= 0; --p) {
      double v = y[p];
      for (int q = p + 1; q < 6; ++q) v -= L[6 * q + p] * u[q];
      u[p] = v / L[6 * p + p];
    }
    for (int p = 0; p < 6; ++p) { Ush[p] = u[p]; out[n3 + 6 * b + p] = u[p]; }
  }
  __syncthreads();
  const double u0 = Ush[0], u1 = Ush[1], u2 = Ush[2], o0 = Ush[3], o1 = Ush[4], o2 = Ush[5];
  for (int k = t; k < N_blb; k += BT) {  // Lambda = invM (slip + K U)   (:610, M_scale = 1)
    const size_t i = (size_t)b * N_blb + k;
    const double *l = lever + 3 * i;
    const double s0 = inv * slip[3 * i], s1 = inv * slip[3 * i + 1], s2 = inv * slip[3 * i + 2];
    if (nf.part) { nf.vnext[3 * i] = s0; nf.vnext[3 * i + 1] = s1; nf.vnext[3 * i + 2] = s2; }
    out[3 * i] = invM2[2 * i] * (s0 + u0 + l[2] * o1 - l[1] * o2);
    out[3 * i + 1] = invM2[2 * i] * (s1 + u1 + l[0] * o2 - l[2] * o0);
    out[3 * i + 2] = invM2[2 * i + 1] * (s2 + u2 + l[1] * o0 - l[0] * o1);
  }
}

// block-diagonal PC: assemble Ninv_b = K_b^T invM_b K_b from its six columns and factor it (6x6)
__global__ void k_pc_block_ninv(const double *__restrict__ cols /* [6][6*N_bod]: column c of every body */,
                                int N_bod, int b_begin, int b_end, double *__restrict__ NL, unsigned *err)
{
  const int b = b_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x);   // bodies [b_begin, b_end) of N_bod
  if (b >= b_end) return;
  double L[36];
  for (int c = 0; c < 6; ++c)
    for (int p = 0; p < 6; ++p) L[6 * p + c] = cols[(size_t)c * 6 * N_bod + 6 * b + p];
  bool ok = true;
  for (int j = 0; j < 6; ++j) {
    double d = L[6 * j + j];
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
    if (!(d > 0.0)) ok = false;
    d = sqrt(d);
    L[6 * j + j] = d;
    for (int i = j + 1; i < 6; ++i) {
      double v = L[6 * i + j];
      for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k];
      L[6 * i + j] = v / d;
    }
    for (int i = 0; i < j; ++i) L[6 * i + j] = 0.0;
  }
  if (!ok) atomicOr(err, (unsigned)RBL_FLAG_NOT_SPD);
  for (int e = 0; e < 36; ++e) NL[36 * (size_t)b + e] = L[e];
}

// The three steps of a block-preconditioner application after invM slip (y1) in ONE launch, one workgroup per body:
// f = K^T y1 (k_KT_x_Lam), U = Ninv^-1 (fsign F - f) through the 6x6 Cholesky factor (:601-608), Lambda = invM (slip + K U)
// = y1 + (invM K) U (:610; invM K: six columns per body kept from the build, no second pass over the factors) --
// and, for the saddle product that follows inside GMRES, K^T Lambda (ktl, may be NULL).  Bodies b_begin .. b_begin + gridDim.x.
__global__ __launch_bounds__(BT) void k_pc_block_tail(const double *__restrict__ lever, const double *__restrict__ y1,
                                                      const double *__restrict__ MK, long stride,
                                                      const double *__restrict__ NL, const double *__restrict__ F, int N_blb,
                                                      int b_begin, double fsign, double *__restrict__ U,
                                                      double *__restrict__ lam, double *__restrict__ ktl, RblNormFold nf,
                                                      const double *__restrict__ win)
{
  __shared__ double s[6][BT];
  __shared__ double us[6];
  const int b = b_begin + blockIdx.x, t = threadIdx.x;
  // RblNormFold: y1 = invM w for an un-normalised Arnoldi vector w (win: its blob part, F: its body part) -- every use of y1 and F
  // below is linear, so they are scaled by 1 / |w| where they are read, and w / |w| goes to nf.vnext
  double nrm = 1.0, inv = 1.0;
  if (nf.part) {
    bf_fold_norm(nf, t & 63, nrm, inv);
    if (t < 6) nf.vnext[3 * (size_t)gridDim.x * N_blb + 6 * (size_t)b + t] = inv * F[6 * (size_t)b + t];   // (all bodies: b_begin = 0)
    if (t == 0 && b == 0) *nf.hout = nrm;
  }
  double f[6] = {0, 0, 0, 0, 0, 0};
  for (int k = t; k < N_blb; k += BT) {
    const size_t idx = 3 * ((size_t)b * N_blb + k);
    const double *l = lever + idx, *v = y1 + idx;
    f[0] += v[0]; f[1] += v[1]; f[2] += v[2];
    f[3] += l[1] * v[2] - l[2] * v[1];
    f[4] += l[2] * v[0] - l[0] * v[2];
    f[5] += l[0] * v[1] - l[1] * v[0];
  }
  block_reduce<6>(f, s, t);
  if (t == 0) {
    const double *L = NL + 36 * (size_t)b;
    double y[6], u[6];
    for (int p = 0; p < 6; ++p) {
      double v = inv * (fsign * F[6 * b + p] - f[p]);
      for (int q = 0; q < p; ++q) v -= L[6 * p + q] * y[q];
      y[p] = v / L[6 * p + p];
    }
    for (int p = 5; p >= 0; --p) {
      double v = y[p];
      for (int q = p + 1; q < 6; ++q) v -= L[6 * q + p] * u[q];
      u[p] = v / L[6 * p + p];
    }
    for (int p = 0; p < 6; ++p) { U[6 * b + p] = u[p]; us[p] = u[p]; }
  }
  __syncthreads();
  double g[6] = {0, 0, 0, 0, 0, 0};
  for (int k = t; k < N_blb; k += BT) {
    const size_t idx = 3 * ((size_t)b * N_blb + k);
    double v[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (nf.part) nf.vnext[idx + d] = inv * win[idx + d];
      double acc = inv * y1[idx + d];
#pragma unroll
      for (int c = 0; c < 6; ++c) acc = __builtin_fma(MK[(size_t)c * stride + idx + d], us[c], acc);
      v[d] = acc;
      lam[idx + d] = acc;
    }
    const double *l = lever + idx;
    g[0] += v[0]; g[1] += v[1]; g[2] += v[2];
    g[3] += l[1] * v[2] - l[2] * v[1];
    g[4] += l[2] * v[0] - l[0] * v[2];
    g[5] += l[0] * v[1] - l[1] * v[0];
  }
  if (ktl) {
    block_reduce<6>(g, s, t);
    if (t < 6) ktl[6 * b + t] = g[t];
  }
}

// saddle epilogue in one launch: out_top = sub - K U (k_K_x_U with alpha = -1) and out_bot = ktl (K^T lambda, computed
// by k_pc_block_tail when the vector came out of the preconditioner)
__global__ void k_saddle_tail(const double *__restrict__ lever, const double *__restrict__ U, int N_blb, long N,
                              double *__restrict__ out, const double *__restrict__ sub, const double *__restrict__ ktl,
                              int nb6)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < nb6) out[3 * N + idx] = ktl[idx];
  if (idx >= N) return;
  const int b = (int)(idx / N_blb);
  const double *u = U + 6 * b, *om = u + 3, *l = lever + 3 * idx;
  const double k0 = u[0] + l[2] * om[1] - l[1] * om[2];
  const double k1 = u[1] + l[0] * om[2] - l[2] * om[0];
  const double k2 = u[2] + l[1] * om[0] - l[0] * om[1];
  out[3 * idx] = sub[3 * idx] - k0; out[3 * idx + 1] = sub[3 * idx + 1] - k1; out[3 * idx + 2] = sub[3 * idx + 2] - k2;
}

// ---- free space, small bodies: the WHOLE block preconditioner in the body frame (see bf_build in rbl_bodies.hip) ----------
// M_b = (I x R) M_body (I x R)^T and K_b = (I x R) K_body blkdiag(R^T, R^T), so with s' = R^T slip, F' = (R^T F, R^T T):
//   y1' = M_body^-1 s',  f' = K_body^T y1',  U' = N_body^-1 (fsign F' - f'),  Lambda' = y1' + (M_body^-1 K_body) U',
//   Lambda = R Lambda',  U = (R U'_lin, R U'_ang),  K^T Lambda = (R, R) K_body^T Lambda'
// with M_body^-1 (n x n), M_body^-1 K_body (n x 6) and chol(N_body) built ONCE per rbl_set_parameters:
// one launch per application, nothing to build per configuration.
__global__ void k_bf_minv(const double *__restrict__ XU, long n, double *__restrict__ Minv)   // Minv = X^T X, X row-major lower
{
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (e >= n) return;
  double acc = 0.0;
  for (long r = (e > q ? e : q); r < n; ++r) acc = __builtin_fma(XU[(size_t)r * n + e], XU[(size_t)r * n + q], acc);
  Minv[(size_t)q * n + e] = acc;
}

__global__ void k_bf_mk(const double *__restrict__ Minv, const double *__restrict__ cfg, long n, double *__restrict__ MK)
{
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (long k = 0; 3 * k < n; ++k) {
    const double c0 = cfg[3 * k], c1 = cfg[3 * k + 1], c2 = cfg[3 * k + 2];
    const double m0 = Minv[(size_t)(3 * k) * n + e], m1 = Minv[(size_t)(3 * k + 1) * n + e], m2 = Minv[(size_t)(3 * k + 2) * n + e];
    acc[0] += m0; acc[1] += m1; acc[2] += m2;                 // K U = U_lin + om x c_k
    acc[3] += m2 * c1 - m1 * c2;                               // om_x: (0, -c2, c1)
    acc[4] += m0 * c2 - m2 * c0;                               // om_y: (c2, 0, -c0)
    acc[5] += m1 * c0 - m0 * c1;                               // om_z: (-c1, c0, 0)
  }
  for (int c = 0; c < 6; ++c) MK[(size_t)c * n + e] = acc[c];
}

__global__ void k_bf_nl(const double *__restrict__ MK, const double *__restrict__ cfg, long n, double *__restrict__ NL, unsigned *err)
{
  __shared__ double N[36];
  const int t = threadIdx.x;
  if (t < 36) {                                                // N[p][c] = sum_rows K_body[row][p] MK[c][row]
    const int p = t / 6, c = t % 6;
    const double *col = MK + (size_t)c * n;
    double acc = 0.0;
    for (long k = 0; 3 * k < n; ++k) {
      const double c0 = cfg[3 * k], c1 = cfg[3 * k + 1], c2 = cfg[3 * k + 2];
      const double v0 = col[3 * k], v1 = col[3 * k + 1], v2 = col[3 * k + 2];
      acc += p == 0 ? v0 : p == 1 ? v1 : p == 2 ? v2 : p == 3 ? c1 * v2 - c2 * v1 : p == 4 ? c2 * v0 - c0 * v2 : c0 * v1 - c1 * v0;
    }
    N[6 * p + c] = acc;
  }
  __syncthreads();
  if (t == 0) {
    double L[36];
    for (int e = 0; e < 36; ++e) L[e] = N[e];
    bool ok = true;
    for (int j = 0; j < 6; ++j) {
      double d = L[6 * j + j];
      for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
      if (!(d > 0.0)) ok = false;
      d = sqrt(d);
      L[6 * j + j] = d;
      for (int i = j + 1; i < 6; ++i) {
        double v = L[6 * i + j];
        for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k];
        L[6 * i + j] = v / d;
      }
      for (int i = 0; i < j; ++i) L[6 * i + j] = 0.0;
    }
    if (!ok) atomicOr(err, (unsigned)RBL_FLAG_NOT_SPD);
    for (int e = 0; e < 36; ++e) NL[e] = L[e];
  }
}

constexpr int BFT = 512;
template <int CTRL>
__device__ __forceinline__ double bf_dpp(double v)       // DPP move of both halves of a double inside a 16-lane row
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void bf_reduce6(double (&v)[6], double (*red)[6], double *res, int t)
{
  // row sums by DPP (a __shfl_xor butterfly on doubles is twelve dependent ds_bpermute round trips per value), the four
  // rows of a wave and the waves of the block through LDS
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    v[q] += bf_dpp<0xB1>(v[q]);      // quad_perm [1,0,3,2]
    v[q] += bf_dpp<0x4E>(v[q]);      // quad_perm [2,3,0,1]
    v[q] += bf_dpp<0x141>(v[q]);     // row_half_mirror
    v[q] += bf_dpp<0x140>(v[q]);     // row_mirror
  }
  __shared__ double rows[BFT / 16][6];
  if ((t & 15) == 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) rows[t >> 4][q] = v[q];
  }
  __syncthreads();
  if ((t & 63) == 0) {
    const int r0 = t >> 4;
#pragma unroll
    for (int q = 0; q < 6; ++q) red[t >> 6][q] = (rows[r0][q] + rows[r0 + 1][q]) + (rows[r0 + 2][q] + rows[r0 + 3][q]);
  }
  __syncthreads();
  if (t < 6) {
    double a = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += red[w][t];
    res[t] = a;
  }
  __syncthreads();
}

// y1' = M_body^-1 R^T slip, 128 outputs per workgroup: a body's product spread over ceil(n / 128) CUs (one workgroup per
// body pulled the whole 1.9 MB table through ONE CU: 14 of that kernel's 35 us at n = 486).  Four waves split the sum
// (interleaved, fixed-order LDS reduction), a lane owns TWO adjacent outputs and reads them with one 16-byte load: the
// vector-memory pipe takes 16 cycles per wave instruction whatever its width, and this kernel is nothing but loads.
constexpr int BFG = 128;                             // outputs per workgroup
constexpr int BFW = 4;                               // waves per workgroup
template <bool PAIR>
// nf (RblNormFold): `in` is an un-normalised Arnoldi vector -- the product is linear, so the OUTPUT is scaled by 1 / |w| (the sum
// of the partials runs beside the main loop), and the first row block of every body stores its blobs of w / |w|
__global__ __launch_bounds__(64 * BFW) void k_bf_gemv(const double *__restrict__ Minv, const double *__restrict__ Q, long n,
                                                      int b_begin, const double *__restrict__ in, double *__restrict__ y1, RblNormFold nf)
{
  extern __shared__ double sv[];                     // s'[n] | partial sums [BFW][128]
  double *red = sv + n;
  const int b = b_begin + blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  double R[9];
  quat_rot(Q + 4 * (size_t)b, R);
  const double *slip = in + (size_t)b * (size_t)n;
  for (long k = t; 3 * k < n; k += 64 * BFW) {       // s' = R^T slip, one blob per thread and pass
    const double a0 = slip[3 * k], a1 = slip[3 * k + 1], a2 = slip[3 * k + 2];
    sv[3 * k] = R[0] * a0 + R[3] * a1 + R[6] * a2;
    sv[3 * k + 1] = R[1] * a0 + R[4] * a1 + R[7] * a2;
    sv[3 * k + 2] = R[2] * a0 + R[5] * a1 + R[8] * a2;
  }
  __syncthreads();
  // symmetric table: row q holds the q-th term of every output, consecutive outputs are adjacent in memory
  const long e0 = (long)blockIdx.x * BFG + 2 * lane;                 // outputs e0, e0 + 1
  const long ec = e0 + 1 < n ? e0 : (n >= 2 ? n - 2 : 0);
  double a0 = 0.0, a1 = 0.0;
  constexpr int UQ = 16;                             // loads in flight per lane
  for (long q0 = w; q0 < n; q0 += (long)BFW * UQ) {
    double m0[UQ], m1[UQ];
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const long q = q0 + (long)BFW * u;
      if (q < n) {
        const double *p = Minv + (size_t)q * (size_t)n + ec;
        if (PAIR) { const double2 mm = *reinterpret_cast<const double2 *>(p); m0[u] = mm.x; m1[u] = mm.y; }
        else { m0[u] = p[0]; m1[u] = p[1]; }
      } else { m0[u] = 0.0; m1[u] = 0.0; }
    }
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const long q = q0 + (long)BFW * u;
      const double s = sv[q < n ? q : n - 1];
      a0 = __builtin_fma(m0[u], s, a0);
      a1 = __builtin_fma(m1[u], s, a1);
    }
  }
  double nrm = 1.0, inv = 1.0;
  if (nf.part) {
    bf_fold_norm(nf, lane, nrm, inv);
    if (blockIdx.x == 0)
      for (long k = t; k < n; k += 64 * BFW) nf.vnext[(size_t)b * (size_t)n + k] = inv * slip[k];
  }
  red[(w * 64 + lane) * 2] = a0; red[(w * 64 + lane) * 2 + 1] = a1;
  __syncthreads();
  if (w == 0) {
    double r0 = red[lane * 2], r1 = red[lane * 2 + 1];
#pragma unroll
    for (int ww = 1; ww < BFW; ++ww) { r0 += red[(ww * 64 + lane) * 2]; r1 += red[(ww * 64 + lane) * 2 + 1]; }
    r0 *= inv; r1 *= inv;
    double *o = y1 + (size_t)b * (size_t)n;
    if (ec == e0) { if (e0 < n) o[e0] = r0; if (e0 + 1 < n) o[e0 + 1] = r1; }
    else if (e0 < n) o[e0] = (e0 == ec + 1) ? r1 : r0;      // last odd entry: the clamped pair (n - 2, n - 1) holds it second
  }
}

__global__ __launch_bounds__(BFT) void k_pc_bodyframe(const double *__restrict__ y1, const double *__restrict__ MK,
                                                      const double *__restrict__ NL, const double *__restrict__ cfg,
                                                      const double *__restrict__ Q, long n, int b_begin, const double *in,
                                                      long n3, double fsign, double *out, double *__restrict__ ktl, RblNormFold nf)
{
  extern __shared__ double sm[];                     // Lambda'[n] | y1'[n]
  __shared__ double red[BFT / 64][6], f6[6], us[6];
  double *sv = sm, *yv = sm + n;
  const int b = b_begin + blockIdx.x, t = threadIdx.x;
  double R[9];
  quat_rot(Q + 4 * (size_t)b, R);
  double nrm = 1.0, inv = 1.0;                       // RblNormFold: y1 arrives scaled (k_bf_gemv), the body rows of `in` are scaled here
  if (nf.part) {
    bf_fold_norm(nf, t & 63, nrm, inv);
    if (t < 6) nf.vnext[n3 + 6 * (size_t)b + t] = inv * in[n3 + 6 * (size_t)b + t];
    if (t == 0 && blockIdx.x == 0) *nf.hout = nrm;
  }
  if (t < n) yv[t] = y1[(size_t)b * (size_t)n + t];  // y1' = M_body^-1 R^T slip (k_bf_gemv)
  __syncthreads();
  double f[6] = {0, 0, 0, 0, 0, 0};
  if (3 * t < n) {                                   // f' = K_body^T y1'
    const double c0 = cfg[3 * t], c1 = cfg[3 * t + 1], c2 = cfg[3 * t + 2];
    const double v0 = yv[3 * t], v1 = yv[3 * t + 1], v2 = yv[3 * t + 2];
    f[0] = v0; f[1] = v1; f[2] = v2;
    f[3] = c1 * v2 - c2 * v1; f[4] = c2 * v0 - c0 * v2; f[5] = c0 * v1 - c1 * v0;
  }
  bf_reduce6(f, red, f6, t);
  if (t == 0) {                                      // U' = N_body^-1 (fsign F' - f'), U = (R U'_lin, R U'_ang)
    const double *F = in + n3 + 6 * (size_t)b;
    double Fb[6], y[6], u[6];
    for (int h = 0; h < 2; ++h)
      for (int d = 0; d < 3; ++d) Fb[3 * h + d] = inv * (R[d] * F[3 * h] + R[3 + d] * F[3 * h + 1] + R[6 + d] * F[3 * h + 2]);   // R^T
    for (int p = 0; p < 6; ++p) {
      double v = fsign * Fb[p] - f6[p];
      for (int q = 0; q < p; ++q) v -= NL[6 * p + q] * y[q];
      y[p] = v / NL[6 * p + p];
    }
    for (int p = 5; p >= 0; --p) {
      double v = y[p];
      for (int q = p + 1; q < 6; ++q) v -= NL[6 * q + p] * u[q];
      u[p] = v / NL[6 * p + p];
    }
    for (int p = 0; p < 6; ++p) us[p] = u[p];
    double *Uo = out + n3 + 6 * (size_t)b;
    for (int h = 0; h < 2; ++h)
      for (int d = 0; d < 3; ++d) Uo[3 * h + d] = R[3 * d] * u[3 * h] + R[3 * d + 1] * u[3 * h + 1] + R[3 * d + 2] * u[3 * h + 2];
  }
  __syncthreads();
  if (t < n) {                                       // Lambda' = y1' + (M_body^-1 K_body) U'
    double acc = yv[t];
#pragma unroll
    for (int c = 0; c < 6; ++c) acc = __builtin_fma(MK[(size_t)c * n + t], us[c], acc);
    sv[t] = acc;
  }
  __syncthreads();
  double g[6] = {0, 0, 0, 0, 0, 0};
  if (3 * t < n) {                                   // Lambda = R Lambda';  K_body^T Lambda'
    const double v0 = sv[3 * t], v1 = sv[3 * t + 1], v2 = sv[3 * t + 2];
    double *lam = out + (size_t)b * (size_t)n + 3 * t;
    lam[0] = R[0] * v0 + R[1] * v1 + R[2] * v2;
    lam[1] = R[3] * v0 + R[4] * v1 + R[5] * v2;
    lam[2] = R[6] * v0 + R[7] * v1 + R[8] * v2;
    const double c0 = cfg[3 * t], c1 = cfg[3 * t + 1], c2 = cfg[3 * t + 2];
    g[0] = v0; g[1] = v1; g[2] = v2;
    g[3] = c1 * v2 - c2 * v1; g[4] = c2 * v0 - c0 * v2; g[5] = c0 * v1 - c1 * v0;
  }
  if (ktl) {
    bf_reduce6(g, red, f6, t);
    if (t < 6) {
      const int h = t / 3, d = t % 3;
      ktl[6 * (size_t)b + t] = R[3 * d] * f6[3 * h] + R[3 * d + 1] * f6[3 * h + 1] + R[3 * d + 2] * f6[3 * h + 2];
    }
  }
}

// ---- two-level factor of the preconditioned Lanczos root (round 3; rbl_roots.hip: tl_build) ----------------------------
// The block-Jacobi factor L leaves the body-body far field to the Krylov iteration; in the Euclidean norm of the increment
// what converges last are the collective translations of the bodies.  Monopole model of the far field:
//     M~ = D + K_t C K_t^T = L (I + Q E Q^T) L^T,    Z_b = L_b^-1 K_t,b  (3 columns per body),  R_b = Z_b^T Z_b = C_b C_b^T,
//     Q_b = Z_b C_b^-T  (orthonormal),   E = blockdiag(C_b^T) C blockdiag(C_b),   C[b,b'] = pair tensor of spheres of the
//     bodies' radius at the body centres (b != b').
// k_tl_orth : per body, R_b, its 3 x 3 Cholesky factor C_b (kept) and Q_b = Z_b C_b^-T in place (Z: [3][n3], column d of
//             every body in vector d -- the bodies do not couple, one vector holds one column of all of them)
// k_tl_E    : A = I + E from the dense sphere tensor (3 N_bod square, column-major), diagonal blocks dropped
// k_tl_qt   : t[3 b + c] = Q_c,b . w_b      (nvec vectors)
// k_tl_addq : wo_b = w_b + sum_c Q_c,b (s[3 b + c] - t[3 b + c])   -- (Op - I) t with s = Op t; wo == w allowed
__global__ __launch_bounds__(BT) void k_tl_orth(double *__restrict__ Z, long n3, int N_blb, double *__restrict__ Cb, unsigned *err)
{
  __shared__ double s[6][BT];
  __shared__ double Ci[9];
  const int b = blockIdx.x, t = threadIdx.x;
  const long o = 3L * b * N_blb, m = 3L * N_blb;
  double *z0 = Z + o, *z1 = Z + n3 + o, *z2 = Z + 2 * n3 + o;
  double f[6] = {0, 0, 0, 0, 0, 0};   // R00 R10 R11 R20 R21 R22
  for (long k = t; k < m; k += BT) {
    const double a = z0[k], bb = z1[k], c = z2[k];
    f[0] += a * a; f[1] += bb * a; f[2] += bb * bb; f[3] += c * a; f[4] += c * bb; f[5] += c * c;
  }
  block_reduce<6>(f, s, t);
  if (t == 0) {
    // lower Cholesky of R (row-major 3 x 3 in Cb) and the inverse of its transpose's action: Q = Z C^-T
    const double c00 = sqrt(f[0]);
    const double c10 = f[1] / c00, c11 = sqrt(f[2] - c10 * c10);
    const double c20 = f[3] / c00, c21 = (f[4] - c20 * c10) / c11, c22 = sqrt(f[5] - c20 * c20 - c21 * c21);
    if (!(f[0] > 0.0) || !(c11 > 0.0) || !(c22 > 0.0)) atomicOr(err, (unsigned)RBL_FLAG_NOT_SPD);
    double *C = Cb + 9 * (size_t)b;
    C[0] = c00; C[1] = 0; C[2] = 0; C[3] = c10; C[4] = c11; C[5] = 0; C[6] = c20; C[7] = c21; C[8] = c22;
    // X = C^-1 (lower): Q = Z X^T, i.e. q_j = sum_{i <= j} X[j][i] z_i
    const double x00 = 1.0 / c00, x11 = 1.0 / c11, x22 = 1.0 / c22;
    const double x10 = -c10 * x00 * x11, x21 = -c21 * x11 * x22, x20 = -(c20 * x00 + c21 * x10) * x22;
    Ci[0] = x00; Ci[1] = 0; Ci[2] = 0; Ci[3] = x10; Ci[4] = x11; Ci[5] = 0; Ci[6] = x20; Ci[7] = x21; Ci[8] = x22;
  }
  __syncthreads();
  for (long k = t; k < m; k += BT) {
    const double a = z0[k], bb = z1[k], c = z2[k];
    z0[k] = Ci[0] * a;
    z1[k] = Ci[3] * a + Ci[4] * bb;
    z2[k] = Ci[6] * a + Ci[7] * bb + Ci[8] * c;
  }
}

__global__ void k_tl_E(const double *__restrict__ Cs /* sphere tensor, n x n */, const double *__restrict__ Cb, int N_bod,
                       double *__restrict__ A)
{
  const int bj = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;   // block (bi, bj) of A
  if (bj >= N_bod) return;
  const long n = 3L * N_bod;
  double Eb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (bi != bj) {
    const double *Ci = Cb + 9 * (size_t)bi, *Cj = Cb + 9 * (size_t)bj;
    double T[9], U[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) T[3 * r + c] = Cs[(size_t)(3 * bj + c) * n + 3 * bi + r];
    for (int r = 0; r < 3; ++r)                       // U = T C_j
      for (int c = 0; c < 3; ++c) U[3 * r + c] = T[3 * r] * Cj[c] + T[3 * r + 1] * Cj[3 + c] + T[3 * r + 2] * Cj[6 + c];
    for (int r = 0; r < 3; ++r)                       // E_b = C_i^T U
      for (int c = 0; c < 3; ++c) Eb[3 * r + c] = Ci[r] * U[c] + Ci[3 + r] * U[3 + c] + Ci[6 + r] * U[6 + c];
  } else {
    Eb[0] = Eb[4] = Eb[8] = 1.0;
  }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) A[(size_t)(3 * bj + c) * n + 3 * bi + r] = Eb[3 * r + c];
}

__global__ __launch_bounds__(BT) void k_tl_qt(const double *__restrict__ Q, long n3, int N_blb, const double *__restrict__ w,
                                              long wpitch, double *__restrict__ tt, long tpitch)
{
  __shared__ double s[3][BT];
  const int b = blockIdx.x, v = blockIdx.y, t = threadIdx.x;
  const long o = 3L * b * N_blb, m = 3L * N_blb;
  const double *wv = w + (size_t)v * wpitch + o;
  double f[3] = {0, 0, 0};
  for (long k = t; k < m; k += BT) {
    const double x = wv[k];
    f[0] += Q[o + k] * x; f[1] += Q[n3 + o + k] * x; f[2] += Q[2 * n3 + o + k] * x;
  }
  block_reduce<3>(f, s, t);
  if (t < 3) tt[(size_t)v * tpitch + 3 * b + t] = f[t];
}

__global__ void k_tl_addq(const double *__restrict__ Q, long n3, int N_blb, const double *__restrict__ sv,
                          const double *__restrict__ tt, long tpitch, const double *w, double *wo, long wpitch)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (k >= n3) return;
  const long b = k / (3L * N_blb);
  const double *s = sv + (size_t)v * tpitch + 3 * b, *t = tt + (size_t)v * tpitch + 3 * b;
  wo[(size_t)v * wpitch + k] = w[(size_t)v * wpitch + k] + Q[k] * (s[0] - t[0]) + Q[n3 + k] * (s[1] - t[1]) + Q[2 * n3 + k] * (s[2] - t[2]);
}

// k_tl_addq with the small triangular operator applied in the same launch (one workgroup per body and vector): rows
// e = 3 b + c of s = Op t, then wo_b = w_b + sum_c Q_c,b (s_e - t_e).   Op (nt x nt, `ld` doubles between columns):
//   kind 0: s = A t,   A lower triangular, column-major (the Cholesky factor L_E):  s_e = sum_{q <= e} A[q ld + e] t_q
//   kind 1: s = X t    from the layout whose COLUMN e holds row e of X (second half of an explicit inverse): q <= e
//   kind 2: s = X^T t  from the layout whose column e holds column e of X (first half):                    q >= e
__global__ __launch_bounds__(BT) void k_tl_eaddq(const double *__restrict__ Q, long n3, int N_blb, const double *__restrict__ Op, long nt,
                                                 long ld, int kind, const double *__restrict__ tt, long tpitch, const double *w, double *wo,
                                                 long wpitch)
{
  __shared__ double s[3][BT];
  const int b = blockIdx.x, v = blockIdx.y, t = threadIdx.x;
  const double *tv = tt + (size_t)v * tpitch;
  double f[3] = {0, 0, 0};
  if (kind == 0) {
    for (long q = t; q <= 3L * b + 2; q += BT) {
      const double x = tv[q];
      const double *a = Op + (size_t)q * ld + 3 * b;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (q <= 3L * b + c) f[c] += a[c] * x;
    }
  } else {
    const long lo = kind == 1 ? 0 : 3L * b, hi = kind == 1 ? 3L * b + 3 : nt;
    for (long q = lo + t; q < hi; q += BT) {
      const double x = tv[q];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long e = 3L * b + c;
        if (kind == 1 ? q <= e : q >= e) f[c] += Op[(size_t)e * ld + q] * x;
      }
    }
  }
  block_reduce<3>(f, s, t);
  const double d0 = f[0] - tv[3 * b], d1 = f[1] - tv[3 * b + 1], d2 = f[2] - tv[3 * b + 2];
  const long o = 3L * b * N_blb, m = 3L * N_blb;
  const double *wv = w + (size_t)v * wpitch + o;
  double *wov = wo + (size_t)v * wpitch + o;
  for (long k = t; k < m; k += BT) wov[k] = wv[k] + Q[o + k] * d0 + Q[n3 + o + k] * d1 + Q[2 * n3 + o + k] * d2;
}

// K_t: unit translation of every body in direction d -> vector d of out ([3][n3])
__global__ void k_tl_unit(long n3, double *__restrict__ out)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n3) return;
  const int d = (int)(k % 3);
  out[k] = d == 0 ? 1.0 : 0.0; out[n3 + k] = d == 1 ? 1.0 : 0.0; out[2 * n3 + k] = d == 2 ? 1.0 : 0.0;
}

__global__ void k_unit_U(int N_bod, int c, double *__restrict__ U)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * N_bod) U[i] = (i % 6 == c) ? 1.0 : 0.0;
}

}  // namespace

void rbl_launch_pc_block_ninv(hipStream_t st, const double *d_cols, int N_bod, double *d_NL, unsigned *d_err, int b_begin,
                              int b_end)
{
  if (b_end < 0) b_end = N_bod;
  if (b_end <= b_begin) return;
  hipLaunchKernelGGL(k_pc_block_ninv, dim3((b_end - b_begin + 63) / 64), dim3(64), 0, st, d_cols, N_bod, b_begin, b_end, d_NL,
                     d_err);
}

void rbl_launch_pc_block_tail(hipStream_t st, const double *d_lever, const double *d_y1, const double *d_MK, int64_t stride,
                              const double *d_NL, const double *d_F, int N_blb, int b_begin, int b_count, double fsign,
                              double *d_U, double *d_lam, double *d_ktl, const RblNormFold *fold, const double *d_win)
{
  if (b_count <= 0) return;
  hipLaunchKernelGGL(k_pc_block_tail, dim3(b_count), dim3(BT), 0, st, d_lever, d_y1, d_MK, (long)stride, d_NL, d_F, N_blb,
                     b_begin, fsign, d_U, d_lam, d_ktl, (fold && b_begin == 0) ? *fold : RblNormFold(), d_win);
}

void rbl_launch_saddle_tail(hipStream_t st, const double *d_lever, const double *d_U, int N_blb, int64_t N, int N_bod,
                            double *d_out, const double *d_sub, const double *d_ktl)
{
  if (N <= 0) return;
  hipLaunchKernelGGL(k_saddle_tail, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_lever, d_U, N_blb, (long)N, d_out,
                     d_sub, d_ktl, 6 * N_bod);
}

// body-frame preconditioner of small bodies: one-time tables from X = L_body^-1 (row-major copy d_XU) ...
void rbl_launch_bf_tables(hipStream_t st, const double *d_XU, const double *d_cfg, int64_t n, double *d_Minv, double *d_MK,
                          double *d_NL, unsigned *d_err)
{
  hipLaunchKernelGGL(k_bf_minv, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, st, d_XU, (long)n, d_Minv);
  hipLaunchKernelGGL(k_bf_mk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double *)d_Minv, d_cfg, (long)n, d_MK);
  hipLaunchKernelGGL(k_bf_nl, dim3(1), dim3(64), 0, st, (const double *)d_MK, d_cfg, (long)n, d_NL, d_err);
}

// ... and the application to bodies [b_begin, b_begin + b_count): d_in = [slip ; F] -> d_out = [Lambda ; U] (full vectors)
// d_y1: scratch, n doubles per body (laid out like the blob vector)
int rbl_launch_pc_bodyframe(hipStream_t st, const double *d_Minv, const double *d_MK, const double *d_NL, const double *d_cfg,
                            const double *d_Q, int64_t n, int b_begin, int b_count, const double *d_in, int64_t n3, double fsign,
                            double *d_out, double *d_ktl, double *d_y1, int gemm, const RblNormFold *fold)
{
  if (n > BFT || !d_y1) return RBL_ERR_SIZE;
  if (b_count <= 0) return RBL_OK;
  const RblNormFold nf = fold ? *fold : RblNormFold();
  if (nf.part && (b_begin != 0 || !rbl_pc_bodyframe_folds(b_count, gemm))) return RBL_ERR_ARG;   // (all bodies, matrix-vector form: see there)
  // y1' = M_body^-1 R^T slip for every body: ONE table, many vectors -- a matrix-matrix product on the fp64 matrix cores
  // (rbl_launch_shared_gemm; round 4) or, RBL_OPT_SHARED_GEMM = 0, a matrix-vector product per body that re-reads the table
  // (measured, tools/bench_shared_gemm.py: 14.9 -> 16.0 us at 50 bodies of 162 blobs, 28.7 -> 16.1 at 200: the full table is twice the
  // triangular factor's work per row tile, so the product only pays once there are enough columns to share it)
  const bool use_gemm = gemm && rbl_shared_gemm_fits(n) && b_count >= 64 && b_count <= 65535;
  if (use_gemm) {
    const int rc = rbl_launch_shared_gemm(st, d_Minv, n, n, 0, d_in + (size_t)b_begin * (size_t)n, d_y1 + (size_t)b_begin * (size_t)n, n, 0,
                                          b_count, 1, d_Q + 4 * (size_t)b_begin, 1);
    if (rc) return rc;
  }
  for (int q0 = 0; q0 < b_count && !use_gemm; q0 += 65535) {      // bodies ride in gridDim.y
    const int nb = b_count - q0 < 65535 ? b_count - q0 : 65535;
    const dim3 grid((unsigned)((n + BFG - 1) / BFG), nb);
    const size_t lds = sizeof(double) * ((size_t)n + 2 * 64 * BFW);
    if (n % 2 == 0 && (reinterpret_cast<uintptr_t>(d_Minv) & 15) == 0)     // rows start 16-byte aligned: one load per output pair
      hipLaunchKernelGGL(k_bf_gemv<true>, grid, dim3(64 * BFW), lds, st, d_Minv, d_Q, (long)n, b_begin + q0, d_in, d_y1, nf);
    else
      hipLaunchKernelGGL(k_bf_gemv<false>, grid, dim3(64 * BFW), lds, st, d_Minv, d_Q, (long)n, b_begin + q0, d_in, d_y1, nf);
  }
  const int th = (int)(n <= 64 ? 64 : ((n + 63) / 64) * 64);
  hipLaunchKernelGGL(k_pc_bodyframe, dim3(b_count), dim3(th), sizeof(double) * 2 * (size_t)n, st, (const double *)d_y1, d_MK, d_NL,
                     d_cfg, d_Q, (long)n, b_begin, d_in, (long)n3, fsign, d_out, d_ktl, nf);
  return RBL_OK;
}

// the one-launch-pair form above can take an un-normalised input (RblNormFold) when it runs the matrix-vector kernel over all bodies
bool rbl_pc_bodyframe_folds(int b_count, int gemm) { return b_count > 0 && b_count <= 65535 && !(gemm && b_count >= 64); }

void rbl_launch_unit_U(hipStream_t st, int N_bod, int c, double *d_U)
{
  if (N_bod <= 0) return;
  hipLaunchKernelGGL(k_unit_U, dim3((6 * N_bod + 255) / 256), dim3(256), 0, st, N_bod, c, d_U);
}

void rbl_launch_body_geom(hipStream_t st, const double *dX, const double *dQ, const double *dcfg,
                          int N_blb, int64_t N, double *d_lever, double *d_pos)
{
  if (N <= 0) return;
  hipLaunchKernelGGL(k_body_geom, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dX, dQ, dcfg, N_blb,
                     (long)N, d_lever, d_pos);
}

void rbl_launch_K_x_U(hipStream_t st, const double *d_lever, const double *d_U, int N_blb, int64_t N,
                      double *d_out, const double *d_sub, double alpha)
{
  if (N <= 0) return;
  hipLaunchKernelGGL(k_K_x_U, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_lever, d_U, N_blb,
                     (long)N, d_out, d_sub, alpha);
}

void rbl_launch_KT_x_Lam(hipStream_t st, const double *d_lever, const double *d_lam, int N_blb, int N_bod,
                         double *d_out)
{
  if (N_bod <= 0) return;
  hipLaunchKernelGGL(k_KT_x_Lam, dim3(N_bod), dim3(BT), 0, st, d_lever, d_lam, N_blb, d_out);
}

void rbl_launch_pc_diag_build(hipStream_t st, const RblParams &P, bool wall, const double *d_lever,
                              const double *d_pos, int N_blb, int N_bod, double *d_invM2, double *d_NL,
                              unsigned *d_err)
{
  if (N_bod <= 0) return;
  if (wall)
    hipLaunchKernelGGL(k_pc_diag_build<true>, dim3(N_bod), dim3(BT), 0, st, d_lever, d_pos, N_blb, P, d_invM2, d_NL, d_err);
  else
    hipLaunchKernelGGL(k_pc_diag_build<false>, dim3(N_bod), dim3(BT), 0, st, d_lever, d_pos, N_blb, P, d_invM2, d_NL, d_err);
}

void rbl_launch_pc_diag_apply(hipStream_t st, const double *d_lever, const double *d_invM2, const double *d_NL,
                              int N_blb, int N_bod, const double *d_in, double *d_out, double fsign, const RblNormFold *fold)
{
  if (N_bod <= 0) return;
  hipLaunchKernelGGL(k_pc_diag_apply, dim3(N_bod), dim3(BT), 0, st, d_lever, d_invM2, d_NL, N_blb,
                     (long)3 * N_blb * N_bod, d_in, d_out, fsign, fold ? *fold : RblNormFold());
}

// ---- two-level factor of the preconditioned Lanczos root (see k_tl_orth) -------------------------------------------------
void rbl_launch_tl_unit(hipStream_t st, int64_t n3, double *d_out)
{
  hipLaunchKernelGGL(k_tl_unit, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, (long)n3, d_out);
}
void rbl_launch_tl_orth(hipStream_t st, double *d_Z, int64_t n3, int N_blb, int N_bod, double *d_Cb, unsigned *d_err)
{
  hipLaunchKernelGGL(k_tl_orth, dim3(N_bod), dim3(BT), 0, st, d_Z, (long)n3, N_blb, d_Cb, d_err);
}
void rbl_launch_tl_E(hipStream_t st, const double *d_Cs, const double *d_Cb, int N_bod, double *d_A)
{
  hipLaunchKernelGGL(k_tl_E, dim3((unsigned)((N_bod + 63) / 64), (unsigned)N_bod), dim3(64), 0, st, d_Cs, d_Cb, N_bod, d_A);
}
void rbl_launch_tl_qt(hipStream_t st, const double *d_Q, int64_t n3, int N_blb, int N_bod, const double *d_w, int64_t wpitch, int nvec,
                      double *d_t, int64_t tpitch)
{
  hipLaunchKernelGGL(k_tl_qt, dim3(N_bod, nvec), dim3(BT), 0, st, d_Q, (long)n3, N_blb, d_w, (long)wpitch, d_t, (long)tpitch);
}
void rbl_launch_tl_eaddq(hipStream_t st, const double *d_Q, int64_t n3, int N_blb, int N_bod, const double *d_Op, int64_t nt, int64_t ld,
                         int kind, const double *d_t, int64_t tpitch, const double *d_w, double *d_wo, int64_t wpitch, int nvec)
{
  hipLaunchKernelGGL(k_tl_eaddq, dim3(N_bod, nvec), dim3(BT), 0, st, d_Q, (long)n3, N_blb, d_Op, (long)nt, (long)ld, kind, d_t, (long)tpitch,
                     d_w, d_wo, (long)wpitch);
}
void rbl_launch_tl_addq(hipStream_t st, const double *d_Q, int64_t n3, int N_blb, const double *d_s, const double *d_t, int64_t tpitch,
                        const double *d_w, double *d_wo, int64_t wpitch, int nvec)
{
  hipLaunchKernelGGL(k_tl_addq, dim3((unsigned)((n3 + 255) / 256), nvec), dim3(256), 0, st, d_Q, (long)n3, N_blb, d_s, d_t, (long)tpitch,
                     d_w, d_wo, (long)wpitch);
}
