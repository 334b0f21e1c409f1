// rbl_small.hip -- the whole right-preconditioned GMRES solve of a SMALL system in ONE kernel launch.
//
// moving the Krylov loop onto the device "removes per-iteration latency at small N".  At
// BASELINE cfg 1 (10 x shell_N_12 = 120 blobs, 360 + 60 unknowns) one iteration of rbl_gmres_saddle_dev is ~6 kernel
// launches of a few microseconds each: the solve is launch-bound (1.0 ms for 21 products).  Here one workgroup of
// 1024 threads on one CU does everything between the right-hand side and the solution with workgroup barriers
// instead of kernel boundaries:
//
//   geometry (lever arms, positions: reference :257-265, :374) -> diagonal preconditioner (diag_invM :489-543,
//   Ninv = K^T invM K and its 6x6 Cholesky :593-594) -> [ z = P^-1 v_j  (:589-616) ; w = A z  (src/Rigid.py:73-80:
//   M lambda - K U | K^T lambda, M by ordered pair evaluation with rbl_pair_accum, reference :641-659) ;
//   classical Gram-Schmidt twice ; Givens update and convergence test ]* -> x = P^-1 V y.
//
// Vectors of the iteration, the new Hessenberg column, the Givens rotations and (up to 64 iterations) the triangular
// factor live in LDS; so does the Krylov basis when it fits beside them in the CU's 160 KB (cfg 1: 71 KB for 21 vectors),
// otherwise it goes to a global workspace (L2-resident).  A first version kept basis and Hessenberg matrix in global
// memory: thread 0's Givens recurrence then was a chain of dependent L2 round trips, ~10 us per iteration.
// Every sum has a fixed order: results are bitwise reproducible, like the rest of the library.
// Limits: N <= 256 blobs, N_bod <= 64, diagonal preconditioner, max_iter <= 255 (checked by the launcher).
#include "rbl_internal.hpp"

namespace {

constexpr int SGT = 1024;          // threads of the one workgroup
constexpr int SG_MAXN = 256;       // blobs
constexpr int SG_MAXB = 64;        // bodies
constexpr int SG_MAXIT = 255;
constexpr int SG_RLDS = 64;        // up to this many iterations the triangular factor R stays in LDS (packed columns)
constexpr size_t SG_LDS_MAX = 150 * 1024;

struct SmallArgs {
  const double *X, *Q, *cfg;       // body state (device): 3 Nb, 4 Nb (scalar-first), N_blb x 3 (mean removed)
  const double *rhs;               // nsys
  const double *x0;                // initial guess or nullptr
  double *x;                       // nsys
  double *V, *H;                   // global workspace: (max_iter+1) nsys (unused when the basis is in LDS) | packed R columns (max_iter > 64)
  int *iters_out;                  // device scalars
  double *resid_out;
  unsigned *err;
  RblParams P;
  int N_blb, N_bod, max_iter;
  double rtol, fsign;
};

__device__ __forceinline__ void quat_rot9(const double *q, double *R)
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

// sum over the wavefront, result in every lane, fixed order.  Inside the 16-lane rows by DPP moves (quad_perm, half-row
// and row mirrors: a few cycles each; the __shfl_xor butterfly is six dependent ds_bpermute round trips, ~800 cycles, and
// this kernel pays it on every dot product), across the four rows through v_readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v)
{
  v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);     // row_half_mirror
  v += dpp_move<0x140>(v);     // row_mirror: every lane holds its row's sum
  return ((lane_value(v, 0) + lane_value(v, 16)) + lane_value(v, 32)) + lane_value(v, 48);
}

template <bool WALL, bool VLDS>
__global__ __launch_bounds__(SGT) void k_gmres_small(SmallArgs A)
{
  extern __shared__ double sm[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int NW = SGT / 64;
  const int nbl = A.N_blb, nb = A.N_bod, N = nbl * nb, n3 = 3 * N, nb6 = 6 * nb, nsys = n3 + nb6;
  const int m = A.max_iter;
  double *pos = sm;                             // 3N  positions / a
  double *lev = pos + n3;                       // 3N  lever arms
  double *iM = lev + n3;                        // 2N  diag_invM: (xx = yy, zz)
  double *dmp = iM + 2 * N;                     // N   wall damping d_i (1 without the wall term)
  double *NLs = dmp + N;                        // 36 Nb  (K^T invM K)^-1, row-major 6x6 per body
  double *vz = NLs + 36 * nb;                   // nsys   z = P^-1 v
  double *vw = vz + nsys;                       // nsys   w
  double *vv = vw + nsys;                       // nsys   current basis vector / scratch
  double *fb = vv + nsys;                       // 6 Nb   body sums
  double *hh = fb + nb6;                        // max_iter + 2: Gram-Schmidt coefficients of one pass, later y
  double *hc = hh + (m + 2);                    // max_iter + 2: the new Hessenberg column
  double *gg = hc + (m + 2);                    // max_iter + 2: rotated right-hand side
  double *cs = gg + (m + 2);                    // max_iter: Givens cosines
  double *sn = cs + m;                          // max_iter: Givens sines
  double *sc = sn + m;                          // 8 scalars: [1] stop flag, [2] residual estimate, [3] |w|^2 before Gram-Schmidt
  double *part = sc + 8;                        // NW x 3N: one accumulator set per wavefront for the symmetric pair sweep (also scratch)
  double *su = part + (size_t)NW * n3;          // 3N  self terms
  double *kt = su + n3;                         // 6N  per-blob terms of K^T lambda
  double *Rl = kt + 6 * N;                      // packed upper-triangular R, column j at j (j+1) / 2 (max_iter <= SG_RLDS), else global
  const bool r_lds = m <= SG_RLDS;
  double *Rm = r_lds ? Rl : A.H;
  double *Vb = VLDS ? Rl + (r_lds ? (size_t)m * (m + 1) / 2 : 0) : A.V;   // Krylov basis
  const RblParams P = A.P;
  const RblParams Pu = {1.0, 1.0, P.nf, 4.0, 1e-24, -0.375, 0.125, 0};
  unsigned flags = 0;

  // ---- geometry + diagonal preconditioner ---------------------------------------------------------------
  if (t < N) {
    const int b = t / nbl, k = t - b * nbl;
    double R[9];
    quat_rot9(A.Q + 4 * b, R);
    const double c0 = A.cfg[3 * k], c1 = A.cfg[3 * k + 1], c2 = A.cfg[3 * k + 2];
    double l0, l1, l2, p2;
    {
#pragma clang fp contract(off)
      l0 = c0 * R[0] + c1 * R[1] + c2 * R[2];
      l1 = c0 * R[3] + c1 * R[4] + c2 * R[5];
      l2 = c0 * R[6] + c1 * R[7] + c2 * R[8];
      lev[3 * t] = l0; lev[3 * t + 1] = l1; lev[3 * t + 2] = l2;
      p2 = l2 + A.X[3 * b + 2];
      pos[3 * t] = (l0 + A.X[3 * b]) * P.inv_a; pos[3 * t + 1] = (l1 + A.X[3 * b + 1]) * P.inv_a; pos[3 * t + 2] = p2 * P.inv_a;
    }
    double dxx = 4.0 / 3.0, dzz = 4.0 / 3.0, d = 1.0;
    if (WALL) {  // self wall term (:98-104), h = z/a; damping (:629-633)
      const double h = p2 / P.a;
      if (h < 0.0) flags |= RBL_FLAG_BELOW_WALL;
      const double iz = 1.0 / h, iz3 = iz * iz * iz, iz5 = iz3 * iz * iz;
      dxx += -(9 * iz - 2 * iz3 + iz5) / 12.0;
      dzz += -(9 * iz - 4 * iz3 + iz5) / 6.0;
      d = (p2 >= P.a) ? 1.0 : p2 / P.a;
    }
    const double scale = 1.0 / P.nf;
    iM[2 * t] = scale / dxx; iM[2 * t + 1] = scale / dzz;
    dmp[t] = d;
  }
  __syncthreads();
  if (t < nb) {   // Ninv_b = sum_k K_k^T D_k K_k (lower triangle), then its 6x6 Cholesky, row-major L
    double acc[21];
    for (int q = 0; q < 21; ++q) acc[q] = 0.0;
    for (int k = 0; k < nbl; ++k) {
      const int i = t * nbl + k;
      const double lx = lev[3 * i], ly = lev[3 * i + 1], lz = lev[3 * i + 2];
      const double Kk[3][6] = {{1, 0, 0, 0, lz, -ly}, {0, 1, 0, -lz, 0, lx}, {0, 0, 1, ly, -lx, 0}};
      const double D[3] = {iM[2 * i], iM[2 * i], iM[2 * i + 1]};
      int q = 0;
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c <= r; ++c) {
          acc[q] += Kk[0][r] * D[0] * Kk[0][c] + Kk[1][r] * D[1] * Kk[1][c] + Kk[2][r] * D[2] * Kk[2][c];
          ++q;
        }
    }
    double L[36];
    int q = 0;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c <= r; ++c) { L[6 * r + c] = acc[q]; if (c < r) L[6 * c + r] = 0.0; ++q; }
    bool ok = true;
    for (int j = 0; j < 6; ++j) {
      double dd = L[6 * j + j];
      for (int k = 0; k < j; ++k) dd -= L[6 * j + k] * L[6 * j + k];
      if (!(dd > 0.0)) ok = false;
      dd = sqrt(dd);
      L[6 * j + j] = dd;
      for (int i = j + 1; i < 6; ++i) {
        double v = L[6 * i + j];
        for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k];
        L[6 * i + j] = v / dd;
      }
    }
    if (!ok) flags |= RBL_FLAG_NOT_SPD;
    // the explicit inverse (L L^T)^-1, column by column: every application is then a 6x6 product spread over six
    // threads instead of one thread's chain of two triangular solves with twelve divisions
    for (int col = 0; col < 6; ++col) {
      double y[6], u[6];
      for (int p = 0; p < 6; ++p) {
        double v = (p == col) ? 1.0 : 0.0;
        for (int q2 = 0; q2 < p; ++q2) v -= L[6 * p + q2] * y[q2];
        y[p] = v / L[6 * p + p];
      }
      for (int p = 5; p >= 0; --p) {
        double v = y[p];
        for (int q2 = p + 1; q2 < 6; ++q2) v -= L[6 * q2 + p] * u[q2];
        u[p] = v / L[6 * p + p];
      }
      for (int p = 0; p < 6; ++p) NLs[36 * t + 6 * p + col] = u[p];
    }
  }
  __syncthreads();

  // ---- operators on LDS vectors (every thread calls them: they contain barriers) ----------------------------
  // out = P^-1 in   (apply_PC with the diagonal invM, :589-616; fsign: see rbl_ctx::pc_fsign)
  auto apply_PC = [&](const double *in, double *out) {
    if (t < N) {                             // per-blob terms of K^T (invM slip) -> part[c][blob]
      const double v0 = iM[2 * t] * in[3 * t], v1 = iM[2 * t] * in[3 * t + 1], v2 = iM[2 * t + 1] * in[3 * t + 2];
      const double l0 = lev[3 * t], l1 = lev[3 * t + 1], l2 = lev[3 * t + 2];
      part[t] = v0; part[N + t] = v1; part[2 * N + t] = v2;
      part[3 * N + t] = l1 * v2 - l2 * v1; part[4 * N + t] = l2 * v0 - l0 * v2; part[5 * N + t] = l0 * v1 - l1 * v0;
    }
    __syncthreads();
    if (t < nb6) {                           // fsign F - K^T (invM slip), component c of body b (blobs added in order)
      const int b = t / 6, c = t - 6 * b;
      const double *p = part + (size_t)c * N + (size_t)b * nbl;
      double f = 0.0;
      for (int k = 0; k < nbl; ++k) f += p[k];
      fb[t] = A.fsign * in[n3 + t] - f;
    }
    __syncthreads();
    if (t < nb6) {                           // U = Ninv^-1 (fsign F - f)   (:601-608), explicit 6x6 inverse
      const int b = t / 6, c = t - 6 * b;
      const double *Nm = NLs + 36 * b + 6 * c, *rh = fb + 6 * b;
      double u = 0.0;
#pragma unroll
      for (int d = 0; d < 6; ++d) u = __builtin_fma(Nm[d], rh[d], u);
      out[n3 + t] = u;
    }
    __syncthreads();
    if (t < N) {                             // Lambda = invM (slip + K U)   (:610)
      const int b = t / nbl;
      const double *u = out + n3 + 6 * b;
      const double l0 = lev[3 * t], l1 = lev[3 * t + 1], l2 = lev[3 * t + 2];
      out[3 * t] = iM[2 * t] * (in[3 * t] + u[0] + l2 * u[4] - l1 * u[5]);
      out[3 * t + 1] = iM[2 * t] * (in[3 * t + 1] + u[1] + l0 * u[5] - l2 * u[3]);
      out[3 * t + 2] = iM[2 * t + 1] * (in[3 * t + 2] + u[2] + l1 * u[3] - l0 * u[4]);
    }
    __syncthreads();
  };

  // out = [M lambda - K U ; K^T lambda]   (src/Rigid.py:73-80; M = B Mob B with the wall term, :641-659)
  // Every unordered pair once (M_ji = M_ij^T, rbl_pair_sym): step (s, rb) pairs the rows i = 64 rb + lane with the columns
  // j = i + s (mod N), s = 1 .. N/2 (for even N the offset N/2 only from the lower half), so the 64 lanes of a wavefront
  // touch 64 different rows and 64 different columns per step; the steps are dealt round-robin to the 16 wavefronts, each
  // adding into its OWN accumulator set (fixed order inside a wave), and the sets are added in wave order afterwards.
  auto apply_A = [&](const double *in, double *out) {
    for (int idx = t; idx < NW * n3; idx += SGT) part[idx] = 0.0;
    if (t < N) {                             // self blocks (:40-46, :98-104) and the per-blob terms of K^T lambda
      const double di = WALL ? dmp[t] : 1.0;
      double ux = 0.0, uy = 0.0, uz = 0.0;
      rbl_pair_accum<WALL, true, true>(Pu, pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], pos[3 * t], pos[3 * t + 1], pos[3 * t + 2],
                                       di * in[3 * t], di * in[3 * t + 1], di * in[3 * t + 2], true, ux, uy, uz, flags);
      su[3 * t] = ux; su[3 * t + 1] = uy; su[3 * t + 2] = uz;
      const double v0 = in[3 * t], v1 = in[3 * t + 1], v2 = in[3 * t + 2];
      const double l0 = lev[3 * t], l1 = lev[3 * t + 1], l2 = lev[3 * t + 2];
      kt[t] = v0; kt[N + t] = v1; kt[2 * N + t] = v2;
      kt[3 * N + t] = l1 * v2 - l2 * v1; kt[4 * N + t] = l2 * v0 - l0 * v2; kt[5 * N + t] = l0 * v1 - l1 * v0;
    }
    __syncthreads();
    {
      const int RB = (N + 63) / 64, nsteps = (N / 2) * RB;
      double *acc = part + (size_t)wave * n3;
      for (int q = wave; q < nsteps; q += NW) {
        const int s_ = q / RB + 1, i = (q - (s_ - 1) * RB) * 64 + lane;
        if (i < N && (2 * s_ != N || 2 * i < N)) {
          int j = i + s_;
          if (j >= N) j -= N;
          const double di = WALL ? dmp[i] : 1.0, dj = WALL ? dmp[j] : 1.0;
          double uix = 0.0, uiy = 0.0, uiz = 0.0, ujx = 0.0, ujy = 0.0, ujz = 0.0;
          rbl_pair_sym<WALL, true, true>(Pu, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], di * in[3 * i], di * in[3 * i + 1],
                                         di * in[3 * i + 2], pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], dj * in[3 * j],
                                         dj * in[3 * j + 1], dj * in[3 * j + 2], uix, uiy, uiz, ujx, ujy, ujz, flags);
          __hip_atomic_fetch_add(&acc[3 * i], uix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&acc[3 * i + 1], uiy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&acc[3 * i + 2], uiz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&acc[3 * j], ujx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&acc[3 * j + 1], ujy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&acc[3 * j + 2], ujz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    __syncthreads();
    if (t < n3) {                            // U_i = nf d_i (self + sum over the waves' sets) - (K U)_i
      double s = su[t];
      for (int w = 0; w < NW; ++w) s += part[(size_t)w * n3 + t];
      const int i = t / 3, c = t - 3 * i, b = i / nbl;
      if (!isfinite(s)) flags |= RBL_FLAG_NONFINITE;
      const double *u = in + n3 + 6 * b;
      const double l0 = lev[3 * i], l1 = lev[3 * i + 1], l2 = lev[3 * i + 2];
      const double ku = c == 0 ? u[0] + l2 * u[4] - l1 * u[5] : c == 1 ? u[1] + l0 * u[5] - l2 * u[3] : u[2] + l1 * u[3] - l0 * u[4];
      out[t] = (WALL ? P.nf * dmp[i] : P.nf) * s - ku;
    }
    if (t < nb6) {                           // K^T lambda: blobs of the body added in order (3N + 6 N_bod may exceed the
      const int b = t / 6, c = t - 6 * b;    // thread count, so this is not an else-branch of the rows above)
      const double *p = kt + (size_t)c * N + (size_t)b * nbl;
      double f = 0.0;
      for (int k = 0; k < nbl; ++k) f += p[k];
      out[n3 + t] = f;
    }
    __syncthreads();
  };

  // squared norm of an LDS vector -> hh[m+1] (every thread gets it): wave partials added in wave order
  auto norm2 = [&](const double *v) -> double {
    double a = 0.0;
    for (int i = t; i < nsys; i += SGT) a = __builtin_fma(v[i], v[i], a);
    a = wave_sum(a);
    if (lane == 0) part[wave] = a;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < NW; ++w) s += part[w];
    __syncthreads();
    return s;
  };

  // ---- r0 = b - A x0 (or b), beta, V_0 --------------------------------------------------------------------
  double bnorm;
  {
    for (int i = t; i < nsys; i += SGT) vw[i] = A.rhs[i];
    __syncthreads();
    bnorm = sqrt(norm2(vw));
    if (A.x0) {
      for (int i = t; i < nsys; i += SGT) vz[i] = A.x0[i];
      __syncthreads();
      apply_A(vz, vv);
      for (int i = t; i < nsys; i += SGT) vw[i] -= vv[i];
      __syncthreads();
    }
  }
  const double beta = A.x0 ? sqrt(norm2(vw)) : bnorm;
  int used = 0;
  const double ibn = (bnorm > 0.0) ? 1.0 / bnorm : 0.0;
  double resid = (bnorm > 0.0) ? beta / bnorm : 0.0;
  const bool trivial = !(beta > 0.0) || (A.rtol > 0.0 && resid < A.rtol);
  if (!trivial) {
    const double ib = 1.0 / beta;
    for (int i = t; i < nsys; i += SGT) { const double v = vw[i] * ib; vv[i] = v; Vb[i] = v; }
    if (t == 0) gg[0] = beta;
    __syncthreads();
#ifdef RBL_SMALL_PROF
    long long tp[6] = {0, 0, 0, 0, 0, 0}, t0 = clock64(), t1;
#define RBL_STAMP(k) do { t1 = clock64(); tp[k] += t1 - t0; t0 = t1; } while (0)
#else
#define RBL_STAMP(k) do { } while (0)
#endif
    for (int j = 0; j < m; ++j) {
      RBL_STAMP(5);
      apply_PC(vv, vz);
      RBL_STAMP(0);
      apply_A(vz, vw);
      RBL_STAMP(1);
      // Classical Gram-Schmidt against V_0..V_j (wave w takes the basis vectors w, w+NW, ...), repeated only when the
      // first pass cancelled more than a factor 10 of |w| (the "twice is enough" test of Daniel-Gragg-Kaufman-Stewart
      // with eta = 0.1: the orthogonality error of a single pass is ~eps |w_before| / |w_after|).  Every update sweep
      // also gathers |w|^2, wave 0's first dot-product sweep gathers |w_before|^2: no separate norm passes.
      double hn2 = 0.0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int k = wave; k <= j; k += NW) {
          const double *vk = Vb + (size_t)k * nsys;
          double a = 0.0, w0 = 0.0;
          for (int i = lane; i < nsys; i += 64) {
            const double wi = vw[i];
            a = __builtin_fma(vk[i], wi, a);
            if (k == 0 && pass == 0) w0 = __builtin_fma(wi, wi, w0);
          }
          a = wave_sum(a);
          if (k == 0 && pass == 0) w0 = wave_sum(w0);
          if (lane == 0) { hh[k] = a; if (k == 0 && pass == 0) sc[3] = w0; }
        }
        __syncthreads();
        double wsq = 0.0;
        for (int i = t; i < nsys; i += SGT) {
          double a = vw[i];
          for (int k = 0; k <= j; ++k) a = __builtin_fma(-hh[k], Vb[(size_t)k * nsys + i], a);
          vw[i] = a;
          wsq = __builtin_fma(a, a, wsq);
        }
        if (t <= j) hc[t] = pass ? hc[t] + hh[t] : hh[t];
        wsq = wave_sum(wsq);
        if (lane == 0) part[wave] = wsq;
        __syncthreads();
        hn2 = 0.0;
        for (int w = 0; w < NW; ++w) hn2 += part[w];
        if (hn2 >= 0.01 * sc[3]) break;      // workgroup-uniform: every thread adds the same 16 numbers in the same order
        __syncthreads();                     // part / hh are rewritten by the second pass
      }
      RBL_STAMP(2);
      const double hn = sqrt(hn2);
      RBL_STAMP(3);
      const double ih = hn > 1e-300 ? 1.0 / hn : 0.0;
      double *vn = Vb + (size_t)(j + 1) * nsys;
      for (int i = t; i < nsys; i += SGT) { const double v = vw[i] * ih; vv[i] = v; vn[i] = v; }
      if (t == 0) {   // Givens update of column j (in LDS) and of the rotated right-hand side; residual estimate
        double cur = hc[0];                  // the running entry stays in a register: the chain is two FMAs per rotation,
        for (int i = 0; i < j; ++i) {        // the LDS reads of cs, sn and the next entry do not depend on it
          const double nxt = hc[i + 1], ci = cs[i], si = sn[i];
          hc[i] = __builtin_fma(ci, cur, si * nxt);
          cur = __builtin_fma(-si, cur, ci * nxt);
        }
        hc[j] = cur; hc[j + 1] = hn;
        const double den = sqrt(__builtin_fma(hc[j], hc[j], hc[j + 1] * hc[j + 1]));   // entries are O(|A P^-1|): no scaling needed
        const double rden = den > 0.0 ? 1.0 / den : 0.0;
        cs[j] = den > 0.0 ? hc[j] * rden : 1.0;
        sn[j] = hc[j + 1] * rden;
        hc[j] = den;
        gg[j + 1] = -sn[j] * gg[j];
        gg[j] = cs[j] * gg[j];
        sc[2] = fabs(gg[j + 1]) * ibn;
        sc[1] = (A.rtol > 0.0 && sc[2] < A.rtol) || !(hn > 1e-300) ? 1.0 : 0.0;
        double *rc = Rm + (size_t)j * (j + 1) / 2;         // column j of R, rows 0..j
        for (int i = 0; i <= j; ++i) rc[i] = hc[i];
      }
      __syncthreads();
      RBL_STAMP(4);
      used = j + 1;
      resid = sc[2];
      if (sc[1] != 0.0) break;               // workgroup-uniform
    }
#ifdef RBL_SMALL_PROF
    if (t == 0) printf("k_gmres_small cycles (s_memtime, 100 MHz ticks?) over %d iterations: PC %lld  A %lld  CGS2 %lld  norm %lld  normalise+Givens %lld  other %lld\n",
                       used, tp[0], tp[1], tp[2], tp[3], tp[4], tp[5]);
#endif
    // y = R^-1 g (thread 0), z = V y, x = P^-1 z (+ x0).  A factor kept in global memory (max_iter > 64) is first staged
    // into the idle partial-sum buffer: the substitution is a chain of dependent reads
    const double *Rs = Rm;
    const int nR = used * (used + 1) / 2;
    if (!r_lds && nR <= NW * n3) {
      for (int i = t; i < nR; i += SGT) part[i] = Rm[i];
      Rs = part;
      __syncthreads();
    }
    if (t == 0) {
      for (int i = used - 1; i >= 0; --i) {
        double v = gg[i];
        for (int k = i + 1; k < used; ++k) v -= Rs[(size_t)k * (k + 1) / 2 + i] * hh[k];
        const double d = Rs[(size_t)i * (i + 1) / 2 + i];
        hh[i] = d != 0.0 ? v / d : 0.0;
        if (!isfinite(hh[i])) flags |= RBL_FLAG_NONFINITE;
      }
    }
    __syncthreads();
    for (int i = t; i < nsys; i += SGT) {
      double a = 0.0;
      for (int k = 0; k < used; ++k) a = __builtin_fma(hh[k], Vb[(size_t)k * nsys + i], a);
      vw[i] = a;
    }
    __syncthreads();
    apply_PC(vw, vz);
    for (int i = t; i < nsys; i += SGT) A.x[i] = A.x0 ? A.x0[i] + vz[i] : vz[i];
  } else {
    for (int i = t; i < nsys; i += SGT) A.x[i] = A.x0 ? A.x0[i] : 0.0;
  }
  if (t == 0) { *A.iters_out = used; *A.resid_out = resid; }
  if (flags) atomicOr(A.err, flags);
}

}  // namespace

static size_t small_lds_base_bytes(int N_blb, int N_bod, int max_iter)
{
  const size_t N = (size_t)N_blb * N_bod, n3 = 3 * N, nb6 = 6 * (size_t)N_bod, nsys = n3 + nb6, m = (size_t)max_iter;
  return sizeof(double) * (2 * n3 + 2 * N + N + 36 * (size_t)N_bod + 3 * nsys + nb6 + 3 * (m + 2) + 2 * m + 8 + (SGT / 64) * n3 + n3 + 6 * N +
                           (max_iter <= SG_RLDS ? m * (m + 1) / 2 : 0));
}
static size_t small_basis_bytes(int N_blb, int N_bod, int max_iter)
{
  return sizeof(double) * (size_t)(max_iter + 1) * ((size_t)3 * N_blb * N_bod + (size_t)6 * N_bod);
}

// does the one-kernel solver cover this system?
bool rbl_gmres_small_fits(int N_blb, int N_bod, int max_iter, bool block_pc)
{
  const long N = (long)N_blb * N_bod;
  if (block_pc || N < 1 || N > SG_MAXN || N_bod > SG_MAXB || max_iter < 1 || max_iter > SG_MAXIT) return false;
  return small_lds_base_bytes(N_blb, N_bod, max_iter) <= SG_LDS_MAX;
}

size_t rbl_gmres_small_work_doubles(int N_blb, int N_bod, int max_iter)
{
  const size_t nsys = (size_t)3 * N_blb * N_bod + (size_t)6 * N_bod;
  return (size_t)(max_iter + 1) * nsys + (size_t)max_iter * (max_iter + 1) / 2 + 8;
}

// d_work: rbl_gmres_small_work_doubles(...) doubles; d_scal: 2 doubles (iterations as an int in the first, residual in the second)
int rbl_launch_gmres_small(hipStream_t st, const RblParams &P, bool wall, const double *dX, const double *dQ, const double *dcfg,
                           int N_blb, int N_bod, const double *d_rhs, const double *d_x0, double *d_x, int max_iter, double rtol,
                           double fsign, double *d_work, double *d_scal, unsigned *d_err)
{
  if (!rbl_gmres_small_fits(N_blb, N_bod, max_iter, false)) return RBL_ERR_SIZE;
  const size_t nsys = (size_t)3 * N_blb * N_bod + (size_t)6 * N_bod;
  size_t lds = small_lds_base_bytes(N_blb, N_bod, max_iter);
  bool vlds = lds + small_basis_bytes(N_blb, N_bod, max_iter) <= SG_LDS_MAX;     // the Krylov basis beside the vectors in LDS?
  SmallArgs A;
  A.X = dX; A.Q = dQ; A.cfg = dcfg; A.rhs = d_rhs; A.x0 = d_x0; A.x = d_x;
  A.V = d_work; A.H = d_work + (size_t)(max_iter + 1) * nsys;
  A.iters_out = (int *)d_scal; A.resid_out = d_scal + 1; A.err = d_err;
  A.P = P; A.N_blb = N_blb; A.N_bod = N_bod; A.max_iter = max_iter; A.rtol = rtol; A.fsign = fsign;
  // more than 64 KB of dynamic LDS needs the attribute (gfx950: 160 KB per CU).  If the runtime refuses the basis goes to
  // global memory; if even the vectors do not fit the caller falls back to the general solver (RBL_ERR_SIZE).
  auto allow = [&](bool v, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    const void *fn = wall ? (v ? (const void *)k_gmres_small<true, true> : (const void *)k_gmres_small<true, false>)
                          : (v ? (const void *)k_gmres_small<false, true> : (const void *)k_gmres_small<false, false>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
  };
  if (vlds && allow(true, lds + small_basis_bytes(N_blb, N_bod, max_iter))) lds += small_basis_bytes(N_blb, N_bod, max_iter);
  else {
    vlds = false;
    if (!allow(false, lds)) return RBL_ERR_SIZE;
  }
  if (vlds) {
    if (wall) hipLaunchKernelGGL((k_gmres_small<true, true>), dim3(1), dim3(SGT), lds, st, A);
    else hipLaunchKernelGGL((k_gmres_small<false, true>), dim3(1), dim3(SGT), lds, st, A);
  } else {
    if (wall) hipLaunchKernelGGL((k_gmres_small<true, false>), dim3(1), dim3(SGT), lds, st, A);
    else hipLaunchKernelGGL((k_gmres_small<false, false>), dim3(1), dim3(SGT), lds, st, A);
  }
  return RBL_OK;
}
