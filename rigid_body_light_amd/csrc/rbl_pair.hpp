// rbl_pair.hpp -- device pair arithmetic for the blob mobility (gfx950, fp64).
//
// Two formulations of the same physics:
//
//  * rbl_block_ref():   the 3x3 block exactly as the reference assembles it
//    (c_rigid_obj.cpp:31-142 and :432-447): same expressions, IEEE division and
//    sqrt, FMA contraction OFF -> entries come out bit-identical to the CPU
//    oracle.  Used by the dense-build kernel, which is HBM-write bound, so the
//    extra ALU work is free.
//
//  * rbl_pair_accum() / rbl_pair_sym() / rbl_pair_block_fast():  the fast form used by the matvec
//    kernels, which are fp64-VALU bound: ONE v_rsq_f64 + a 3rd-order Newton step per distance, no
//    division, h_hat eliminated algebraically ( h_hat ez = z_j/R, (1-h_hat) ez = z_i/R ), the wall
//    "facts" as Horner polynomials (rbl_wall_coeffs) and the block applied in vector form
//        M F = cF F + dl [beta (dl.F) + gxz Fz] + z^ [gzx (dl.F) + (mzz - cF) Fz],   dl = (dx, dy, 0).
//    Agrees with the reference to ~1e-15 relative per pair (tests pin <=1e-12 on apply_M).
#pragma once
#include <hip/hip_runtime.h>

#define RBL_FLAG_OVERLAP 1
#define RBL_FLAG_BELOW_WALL 2
#define RBL_FLAG_NONFINITE 4
#define RBL_FLAG_NOT_SPD 8
#define RBL_FLAG_INTERNAL 16   // a bounded wait between workgroups ran out (rbl_tilechol.hip): never expected, never a hang

struct RblParams {
  double a;        // blob radius
  double inv_a;    // 1/a
  double nf;       // 1/(8 pi eta a)            c_rigid_obj.cpp:415
  double four_a2;  // (2a)^2, far/overlap switch c_rigid_obj.cpp:62
  double tiny2;    // (1e-12 a)^2, overlap abort c_rigid_obj.cpp:53
  double c_near_A; // -(3/8)/a   : 4/3 (1 - 9/32 r/a) = 4/3 + c_near_A r
  double c_near_B; // (1/8)/a    : 4/3 * 3/32 (a/r) / a^2 = c_near_B / r
  int no_damp;     // 1: the matvec kernels skip the damping B (plain wall-corrected M), used by the preconditioned square root
};

// ---------------------------------------------------------------------------
// 1/sqrt(x), full double precision: v_rsq_f64 seed (measured max rel. error 5.2e-8 on
// gfx950, tools/peak_fp64) + one 3rd-order correction -> error ~ e^3 ~ 1e-22 + rounding.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double rbl_rsqrt(double x)
{
  double y = __builtin_amdgcn_rsq(x);
  double h = x * y;
  double e = __builtin_fma(-h, y, 1.0);            // 1 - x y^2
  double p = __builtin_fma(e, 0.375, 0.5);         // 1/2 + 3/8 e
  double ye = y * e;
  y = __builtin_fma(ye, p, y);                     // y (1 + e/2 + 3 e^2/8)
  return y;
}

// ---------------------------------------------------------------------------
// Wall-corrected block of one pair (i <- j, h = z_j) as five scalars:
//     M_ij = cF I + beta dl dl^T + gxz dl z^T + gzx z dl^T + (mzz - cF) z z^T,    dl = (dx, dy, 0)
// (A, Bc: the free-space RPY coefficients; q = dx^2 + dy^2).  With e = (dx, dy, Rz)/R, Rz = z_i + z_j,
// w = a/R, u = w^2, v = e_z^2, g = z_j/R (= h_hat e_z), k = z_i/R (= (1 - h_hat) e_z) every "fact" of the
// reference (c_rigid_obj.cpp:108-130) is  w x (polynomial in u), evaluated in Horner form:
//   fact1 = w b1,  b1 = -1 - 2gk + u[(2v - 2/3) + u(2/3 - 10/3 v)]
//   fact2 = w b2,  b2 = -1 + 6gk + u[(2 - 10v) + u(70/3 v - 10/3)]
//   fact5 = w b5,  b5 = -4g^2 + u[-4v + u(20v - 8/3)]
//   fact2 ez + fact3 = w (T0 - T1),  fact2 ez + fact4 = w (T0 + T1)     with
//   T0 = 2g - ez,  T1 = ez { 6gk + u[(2 - 10v) + u(70/3 v - 10)] }
// Swapping the roles of i and j swaps g <-> k, i.e. T1 -> -T1 only: M_ji = M_ij^T exactly.
// ---------------------------------------------------------------------------
// UNIT: the caller works in coordinates already divided by the blob radius (a = 1), which removes the
// a/R and a/r multiplications from every pair.
// Three of the polynomial coefficients come in pairs of non-inline constants (fma(v, -10/3, 2/3), ...).
// A gfx9 VOP3 instruction reads at most one SGPR/constant, so the compiler re-materialises the second one
// into a VGPR with a v_mov_b64 on EVERY evaluation.  A caller with registers to spare passes them in as
// opaque VGPR-resident values (RblWallK from rbl_wall_k_resident()) and saves those two issue slots per pair.
struct RblWallK {
  double k1, k2, k3, k4;   // -10/3, 70/3, 20, 3/2
};
__device__ __forceinline__ RblWallK rbl_wall_k_literal() { return RblWallK{-10.0 / 3.0, 70.0 / 3.0, 20.0, 1.5}; }
__device__ __forceinline__ RblWallK rbl_wall_k_resident()
{
  RblWallK K = rbl_wall_k_literal();
  asm volatile("" : "+v"(K.k1), "+v"(K.k2), "+v"(K.k3), "+v"(K.k4));   // opaque to the optimiser: stays in VGPRs
  return K;
}

template <bool UNIT = false>
__device__ __forceinline__ void rbl_wall_coeffs(const RblParams &P, double dz, double zi, double zj, double q,
                                                double r2, double A, double Bc, double &cF, double &beta, double &gxz,
                                                double &gzx, double &mzz, const RblWallK &K = rbl_wall_k_literal())
{
  // Further identities that take instructions out (84 -> 75 fp64 instructions per unordered pair):
  //   T0 = 2g - ez = (z_j - z_i)/R = -dz/R             (g never formed)
  //   gk = z_i z_j / R^2 = (1 - r^2/R^2) / 4           (R^2 - r^2 = 4 z_i z_j)
  //   v = ez^2 = 1 - q/R^2                             (ez itself is only needed as Rz/R inside Rz w/R^2)
  //   gk and v only enter O(1) polynomials, so the cancellation when z_i z_j << R^2 or Rz^2 << R^2 costs an
  //   absolute ~1e-16 there, nothing relative to the block
  //   2 ez T0 - 4 g^2 = -T0^2 - v                      (the zz bracket: -T0^2 - v (b2 + 1) + u d1), and its -w T0^2 =
  //   -dz^2 w/R^2 cancels the same term of Bc dz^2 = (Bc - w/R^2) dz^2 + dz^2 w/R^2:  mzz = cF + gm dz + w (u d1 - v b2p)
  //   T1 = ez (b2 + 1 - 20/3 u^2),  b2 + 1 = 6 gk + u t2 =: b2p
  //   fact2 ez -+ T1 enter as  gm -+ T1' (Rz w / R^2):  two FMAs on the shared product, no separate add/sub
  //   Bc dz + T0 w / R = dz (Bc - w / R^2) = gm,  and  Bc + b2 w / R^2 = (Bc - w/R^2) + b2p w / R^2
  const double Rz = zi + zj;
  const double R2 = __builtin_fma(Rz, Rz, q);
  const double invR = rbl_rsqrt(R2);
  const double w = UNIT ? invR : P.a * invR;
  const double iR2 = invR * invR;
  const double u = w * w;
  const double v = __builtin_fma(-q, iR2, 1.0);
  const double rr = r2 * iR2;                                                        // (r/R)^2 = 1 - 4 gk
  const double t1 = __builtin_fma(u, __builtin_fma(v, K.k1, 2.0 / 3.0), __builtin_fma(v, 2.0, -2.0 / 3.0));
  const double b1 = __builtin_fma(u, t1, __builtin_fma(0.5, rr, -K.k4));             // -1 - 2 gk + u t1
  const double t2 = __builtin_fma(u, __builtin_fma(v, K.k2, -10.0 / 3.0), __builtin_fma(v, -10.0, 2.0));
  const double sixgk = __builtin_fma(-rr, K.k4, K.k4);
  const double b2p = __builtin_fma(u, t2, sixgk);                                    // b2 + 1
  const double uu = u * u;
  const double T1p = __builtin_fma(uu, -20.0 / 3.0, b2p);                            // T1 / ez
  const double w3 = w * iR2;                                                         // w / R^2
  const double Bw = Bc - w3;
  cF = __builtin_fma(w, b1, A);
  beta = __builtin_fma(b2p, w3, Bw);                                                 // lateral dyad: (Bc + fact2/R^2) dl dl^T
  const double gm = dz * Bw, gdw = Rz * w3;
  gxz = __builtin_fma(-T1p, gdw, gm);                                                // M_xz = dx gxz, M_yz = dy gxz
  gzx = __builtin_fma(T1p, gdw, gm);                                                 // M_zx = dx gzx, M_zy = dy gzx
  // zz bracket  u d1 - v b2p,  d1 = u (20 v - 8/3) - 4 v:   u^2 (20 v - 8/3) - v (b2p + 4 u)
  const double c = __builtin_fma(uu, __builtin_fma(v, K.k3, -8.0 / 3.0), -(v * __builtin_fma(4.0, u, b2p)));
  mzz = __builtin_fma(w, c, __builtin_fma(gm, dz, cF));
}

// ---------------------------------------------------------------------------
// Fast matrix-free accumulation of one ordered pair (i <- j).
//   dx,dy,dz = r_i - r_j ;  zi, zj heights ; (Fx,Fy,Fz) = (damped) force on j
//   SELF: compile-time "this j-tile may contain i" (index-equality self term)
// Accumulates UNSCALED (units 1/(8 pi eta a)) into ux,uy,uz.
// ---------------------------------------------------------------------------
template <bool WALL, bool SELF, bool UNIT = false>
__device__ __forceinline__ void rbl_pair_accum(const RblParams &P, double xi, double yi,
                                               double zi, double xj, double yj, double zj,
                                               double Fx, double Fy, double Fz, bool is_self,
                                               double &ux, double &uy, double &uz,
                                               unsigned &flags, const RblWallK &K = rbl_wall_k_literal())
{
  const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
  const double q = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, q);

  // ---- free-space RPY (c_rigid_obj.cpp:48-79) in physical units ----------
  const double invr = rbl_rsqrt(r2);
  const double invr2 = invr * invr;
  const double s = UNIT ? invr : P.a * invr;         // a/r
  const double t = s * s;                            // (a/r)^2
  // far:  A = (a/r)(1 + 2/3 (a/r)^2),  Bc = (a/r)(1 - 2 (a/r)^2)/r^2
  double A = __builtin_fma(s * t, 2.0 / 3.0, s);
  double Bc = (s * invr2) * __builtin_fma(-2.0, t, 1.0);
  if (__builtin_expect(__any(r2 < P.four_a2), 0)) {  // wave-uniform: the overlap branch (and i == j) is rare
    // overlap: A = 4/3 - 3/8 r/a,  Bc = (1/8) / (a r)
    const double rr = r2 * invr;                     // r
    const bool far = r2 >= P.four_a2;
    A = far ? A : __builtin_fma(rr, P.c_near_A, 4.0 / 3.0);
    Bc = far ? Bc : invr * P.c_near_B;
    if (r2 < P.tiny2 && !(SELF && is_self)) flags |= RBL_FLAG_OVERLAP;   // :53-58
  }
  if (SELF) {                                        // index equality, :40-46
    A = is_self ? 4.0 / 3.0 : A;
    Bc = is_self ? 0.0 : Bc;
  }
  const double q2 = __builtin_fma(dy, Fy, dx * Fx);

  if (!WALL) {
    const double tB = Bc * __builtin_fma(dz, Fz, q2);
    ux = __builtin_fma(A, Fx, __builtin_fma(tB, dx, ux));
    uy = __builtin_fma(A, Fy, __builtin_fma(tB, dy, uy));
    uz = __builtin_fma(A, Fz, __builtin_fma(tB, dz, uz));
    return;
  }

  // ---- single-wall correction (c_rigid_obj.cpp:98-140), ordered pair, h = z_j
  if (SELF && is_self) {
    // self wall term (:98-104): diag only, args (0,0,2h; h = z_i/a)
    const double iz = (UNIT ? 1.0 : P.a) / zi;       // 1/h  (true division kept: rare path)
    const double iz3 = iz * iz * iz;
    const double iz5 = iz3 * iz * iz;
    const double dpar = -(9.0 * iz - 2.0 * iz3 + iz5) / 12.0;
    const double dper = -(9.0 * iz - 4.0 * iz3 + iz5) / 6.0;
    ux = __builtin_fma(A + dpar, Fx, ux);
    uy = __builtin_fma(A + dpar, Fy, uy);
    uz = __builtin_fma(A + dper, Fz, uz);
    return;
  }
  double cF, beta, gxz, gzx, mzz;
  rbl_wall_coeffs<UNIT>(P, dz, zi, zj, q, r2, A, Bc, cF, beta, gxz, gzx, mzz, K);
  const double lat = __builtin_fma(beta, q2, gxz * Fz);
  ux = __builtin_fma(cF, Fx, __builtin_fma(lat, dx, ux));
  uy = __builtin_fma(cF, Fy, __builtin_fma(lat, dy, uy));
  uz = __builtin_fma(mzz, Fz, __builtin_fma(gzx, q2, uz));
}

// ---------------------------------------------------------------------------
// Symmetric pair (i != j): evaluates the scalar coefficients ONCE and applies the
// block in both directions,
//     U_i += M_ij F_j            (accumulated into uix,uiy,uiz)
//     U_j += M_ji F_i            (accumulated into ujx,ujy,ujz; caller adds it to j)
// M_ji = M_ij^T holds exactly for the RPY part and, for the wall part, through the
// role swap g <-> k (h = z_i instead of z_j): fact1, fact2, the e_z-part of fact3
// and the h-free part of fact5 are shared.  ~78 fp64 instructions per unordered wall pair
// (measured in the ISA of k_apply_M_sym<true,2>) vs 2 x ~68 for two ordered evaluations.
// ---------------------------------------------------------------------------
// NEARCHK = false: the caller has proved (tile bounding boxes) that no pair of this sweep is closer than 2a,
// the overlap branch and its per-pair compare are compiled out.
template <bool WALL, bool UNIT = false, bool NEARCHK = true>
__device__ __forceinline__ void rbl_pair_sym(const RblParams &P, double xi, double yi, double zi,
                                             double Fix, double Fiy, double Fiz, double xj,
                                             double yj, double zj, double Fjx, double Fjy,
                                             double Fjz, double &uix, double &uiy, double &uiz,
                                             double &ujx, double &ujy, double &ujz,
                                             unsigned &flags, const RblWallK &K = rbl_wall_k_literal())
{
  const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
  const double q = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, q);
  const double invr = rbl_rsqrt(r2);
  const double invr2 = invr * invr;
  const double s = UNIT ? invr : P.a * invr;
  const double s3 = (s * s) * s;
  double A = __builtin_fma(s3, 2.0 / 3.0, s);              // (a/r)(1 + 2/3 (a/r)^2)
  double Bc = __builtin_fma(s3, -2.0, s) * invr2;          // (a/r)(1 - 2 (a/r)^2) / r^2
  if (NEARCHK && __builtin_expect(__any(r2 < P.four_a2), 0)) {   // wave-uniform: overlap branch is rare
    const double rr = r2 * invr;
    const double A_near = __builtin_fma(rr, P.c_near_A, 4.0 / 3.0);
    const double B_near = invr * P.c_near_B;
    const bool far = r2 >= P.four_a2;
    A = far ? A : A_near;
    Bc = far ? Bc : B_near;
    if (r2 < P.tiny2) flags |= RBL_FLAG_OVERLAP;
  }
  if (!WALL) {   // free space: the vector form A F + Bc (d.F) d is cheaper than forming the block
    const double q2j = __builtin_fma(dy, Fjy, dx * Fjx);
    const double q2i = __builtin_fma(dy, Fiy, dx * Fix);
    const double tBj = Bc * __builtin_fma(dz, Fjz, q2j);
    const double tBi = Bc * __builtin_fma(dz, Fiz, q2i);
    uix = __builtin_fma(A, Fjx, __builtin_fma(tBj, dx, uix));
    uiy = __builtin_fma(A, Fjy, __builtin_fma(tBj, dy, uiy));
    uiz = __builtin_fma(A, Fjz, __builtin_fma(tBj, dz, uiz));
    ujx = __builtin_fma(A, Fix, __builtin_fma(tBi, dx, ujx));
    ujy = __builtin_fma(A, Fiy, __builtin_fma(tBi, dy, ujy));
    ujz = __builtin_fma(A, Fiz, __builtin_fma(tBi, dz, ujz));
    return;
  }

  // Wall: coefficients for ONE direction (h = z_j); M_ji = M_ij^T exactly, so U_j += M_ij^T F_i reuses
  // them.  Vector form: 20 FMAs for both directions instead of forming nine entries.
  double cF, beta, gxz, gzx, mzz;
  rbl_wall_coeffs<UNIT>(P, dz, zi, zj, q, r2, A, Bc, cF, beta, gxz, gzx, mzz, K);
  // U_i += M F_j
  const double pj = __builtin_fma(dy, Fjy, dx * Fjx);
  const double lj = __builtin_fma(beta, pj, gxz * Fjz);
  uix = __builtin_fma(cF, Fjx, __builtin_fma(lj, dx, uix));
  uiy = __builtin_fma(cF, Fjy, __builtin_fma(lj, dy, uiy));
  uiz = __builtin_fma(mzz, Fjz, __builtin_fma(gzx, pj, uiz));
  // U_j += M^T F_i
  const double pi = __builtin_fma(dy, Fiy, dx * Fix);
  const double li = __builtin_fma(beta, pi, gzx * Fiz);
  ujx = __builtin_fma(cF, Fix, __builtin_fma(li, dx, ujx));
  ujy = __builtin_fma(cF, Fiy, __builtin_fma(li, dy, ujy));
  ujz = __builtin_fma(mzz, Fiz, __builtin_fma(gxz, pi, ujz));
}

// ---------------------------------------------------------------------------
// The same for TWO right-hand sides at once: the scalar coefficients (~60 of the ~84 fp64 instructions of
// a wall pair) are shared, only the 20-FMA application is repeated.  Used for the two Brownian increments
// of the stochastic step (lock-step Lanczos) and for 2-3 simultaneous vectors in general.
// ---------------------------------------------------------------------------
struct RblV3 {
  double x, y, z;
};

template <bool WALL, bool UNIT = false, bool NEARCHK = true>
__device__ __forceinline__ void rbl_pair_sym2(const RblParams &P, double xi, double yi, double zi, const RblV3 &Fi0,
                                              const RblV3 &Fi1, double xj, double yj, double zj, const RblV3 &Fj0,
                                              const RblV3 &Fj1, RblV3 &ui0, RblV3 &ui1, RblV3 &uj0, RblV3 &uj1,
                                              unsigned &flags, const RblWallK &K = rbl_wall_k_literal())
{
  const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
  const double q = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, q);
  const double invr = rbl_rsqrt(r2);
  const double invr2 = invr * invr;
  const double s = UNIT ? invr : P.a * invr;
  const double s3 = (s * s) * s;
  double A = __builtin_fma(s3, 2.0 / 3.0, s);
  double Bc = __builtin_fma(s3, -2.0, s) * invr2;
  if (NEARCHK && __builtin_expect(__any(r2 < P.four_a2), 0)) {
    const double rr = r2 * invr;
    const bool far = r2 >= P.four_a2;
    A = far ? A : __builtin_fma(rr, P.c_near_A, 4.0 / 3.0);
    Bc = far ? Bc : invr * P.c_near_B;
    if (r2 < P.tiny2) flags |= RBL_FLAG_OVERLAP;
  }
  if (!WALL) {
    auto app = [&](const RblV3 &F, RblV3 &u) {
      const double tB = Bc * __builtin_fma(dz, F.z, __builtin_fma(dy, F.y, dx * F.x));
      u.x = __builtin_fma(A, F.x, __builtin_fma(tB, dx, u.x));
      u.y = __builtin_fma(A, F.y, __builtin_fma(tB, dy, u.y));
      u.z = __builtin_fma(A, F.z, __builtin_fma(tB, dz, u.z));
    };
    app(Fj0, ui0); app(Fj1, ui1); app(Fi0, uj0); app(Fi1, uj1);
    return;
  }
  double cF, beta, gxz, gzx, mzz;
  rbl_wall_coeffs<UNIT>(P, dz, zi, zj, q, r2, A, Bc, cF, beta, gxz, gzx, mzz, K);
  auto app = [&](const RblV3 &F, RblV3 &u, double g_lat, double g_z) {   // M F (g_lat = gxz, g_z = gzx) or M^T F (swapped)
    const double p = __builtin_fma(dy, F.y, dx * F.x);
    const double l = __builtin_fma(beta, p, g_lat * F.z);
    u.x = __builtin_fma(cF, F.x, __builtin_fma(l, dx, u.x));
    u.y = __builtin_fma(cF, F.y, __builtin_fma(l, dy, u.y));
    u.z = __builtin_fma(mzz, F.z, __builtin_fma(g_z, p, u.z));
  };
  app(Fj0, ui0, gxz, gzx); app(Fj1, ui1, gxz, gzx);      // U_i += M F_j
  app(Fi0, uj0, gzx, gxz); app(Fi1, uj1, gzx, gxz);      // U_j += M^T F_i
}

// ---------------------------------------------------------------------------
// Full 3x3 block of one ORDERED pair (i <- j, h = z_j) from the fast coefficient
// arithmetic, row-major m[9], unscaled.  Used by the multi-RHS kernel, where the
// block is the A-operand of fp64 MFMAs:  M = cF I + Bc d d^T + f2 e e^T + f3 e z^T
// + f4 z e^T + f5 z z^T.
// ---------------------------------------------------------------------------
template <bool WALL, bool SELF, bool UNIT = false>
__device__ __forceinline__ void rbl_pair_block_fast(const RblParams &P, double xi, double yi,
                                                    double zi, double xj, double yj, double zj,
                                                    bool is_self, double *m, unsigned &flags,
                                                    const RblWallK &K = rbl_wall_k_literal())
{
  const double dx = xi - xj, dy = yi - yj, dz = zi - zj;
  const double q = __builtin_fma(dy, dy, dx * dx);
  const double r2 = __builtin_fma(dz, dz, q);
  const double invr = rbl_rsqrt(r2);
  const double invr2 = invr * invr;
  const double s = UNIT ? invr : P.a * invr;
  const double t = s * s;
  double A = __builtin_fma(s * t, 2.0 / 3.0, s);
  double Bc = (s * invr2) * __builtin_fma(-2.0, t, 1.0);
  if (__builtin_expect(__any(r2 < P.four_a2), 0)) {
    const double rr = r2 * invr;
    const bool far = r2 >= P.four_a2;
    A = far ? A : __builtin_fma(rr, P.c_near_A, 4.0 / 3.0);
    Bc = far ? Bc : invr * P.c_near_B;
    if (r2 < P.tiny2 && !(SELF && is_self)) flags |= RBL_FLAG_OVERLAP;
  }
  if (SELF) {
    A = is_self ? 4.0 / 3.0 : A;
    Bc = is_self ? 0.0 : Bc;
  }
  if (!WALL) {
    const double Bdx = Bc * dx, Bdy = Bc * dy, Bdz = Bc * dz;
    m[0] = __builtin_fma(Bdx, dx, A); m[1] = Bdx * dy; m[2] = Bdx * dz;
    m[3] = m[1]; m[4] = __builtin_fma(Bdy, dy, A); m[5] = Bdy * dz;
    m[6] = m[2]; m[7] = m[5]; m[8] = __builtin_fma(Bdz, dz, A);
    return;
  }
  if (SELF && is_self) {  // self wall term (:98-104): diagonal only
    const double iz = (UNIT ? 1.0 : P.a) / zi;
    const double iz3 = iz * iz * iz;
    const double iz5 = iz3 * iz * iz;
    const double dpar = -(9.0 * iz - 2.0 * iz3 + iz5) / 12.0;
    const double dper = -(9.0 * iz - 4.0 * iz3 + iz5) / 6.0;
    m[0] = A + dpar; m[1] = 0.0; m[2] = 0.0;
    m[3] = 0.0; m[4] = A + dpar; m[5] = 0.0;
    m[6] = 0.0; m[7] = 0.0; m[8] = A + dper;
    return;
  }
  double cF, beta, gxz, gzx, mzz;
  rbl_wall_coeffs<UNIT>(P, dz, zi, zj, q, r2, A, Bc, cF, beta, gxz, gzx, mzz, K);
  const double bx = beta * dx, by = beta * dy;
  m[0] = __builtin_fma(bx, dx, cF); m[1] = bx * dy; m[2] = dx * gxz;
  m[3] = m[1]; m[4] = __builtin_fma(by, dy, cF); m[5] = dy * gxz;
  m[6] = dx * gzx; m[7] = dy * gzx; m[8] = mzz;
}

// ---------------------------------------------------------------------------
// Reference-order block (bit-compatible with the oracle).  b: row-major 3x3,
// NOT yet scaled by nf.  lo/hi are the (i<=j) roles of c_rigid_obj.cpp:430-447.
// ---------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ void rbl_rpy_ref(double rx, double ry, double rz, bool self,
                                            double inv_a, double *o, unsigned &flags)
{
  const double four3 = 4.0 / 3.0;
  if (self) {
    o[0] = four3; o[1] = 0.0; o[2] = 0.0; o[3] = four3; o[4] = 0.0; o[5] = four3;
    return;
  }
  rx = rx * inv_a; ry = ry * inv_a; rz = rz * inv_a;
  const double r2 = rx * rx + ry * ry + rz * rz;
  const double r = __builtin_sqrt(r2);
  if (r < 1e-12) flags |= RBL_FLAG_OVERLAP;
  const double invr = 1.0 / r;
  const double invr2 = invr * invr;
  if (r >= 2.0) {
    const double c1 = 1.0 + 2.0 / (3.0 * r2);
    const double c2 = (1.0 - 2.0 * invr2) * invr2;
    o[0] = (c1 + c2 * rx * rx) * invr;
    o[1] = (c2 * rx * ry) * invr;
    o[2] = (c2 * rx * rz) * invr;
    o[3] = (c1 + c2 * ry * ry) * invr;
    o[4] = (c2 * ry * rz) * invr;
    o[5] = (c1 + c2 * rz * rz) * invr;
  } else {
    const double c1 = four3 * (1.0 - 0.28125 * r);
    const double c2 = four3 * 0.09375 * invr;
    o[0] = c1 + c2 * rx * rx;
    o[1] = c2 * rx * ry;
    o[2] = c2 * rx * rz;
    o[3] = c1 + c2 * ry * ry;
    o[4] = c2 * ry * rz;
    o[5] = c1 + c2 * rz * rz;
  }
}

__device__ __forceinline__ void rbl_wall_ref(double rx, double ry, double rz, double *M,
                                             bool self, double hj, unsigned &flags)
{
  if (hj < 0.0) { flags |= RBL_FLAG_BELOW_WALL; }
  if (self) {
    const double iz = 1.0 / hj;
    const double iz3 = iz * iz * iz;
    const double iz5 = iz3 * iz * iz;
    M[0] += -(9 * iz - 2 * iz3 + iz5) / 12.0;
    M[4] += -(9 * iz - 2 * iz3 + iz5) / 12.0;
    M[8] += -(9 * iz - 4 * iz3 + iz5) / 6.0;
    return;
  }
  const double hh = hj / rz;
  const double iR = 1.0 / __builtin_sqrt(rx * rx + ry * ry + rz * rz);
  const double ex = rx * iR, ey = ry * iR, ez = rz * iR;
  const double iR3 = iR * iR * iR;
  const double iR5 = iR3 * iR * iR;
  const double f1 = -(3 * (1 + 2 * hh * (1 - hh) * ez * ez) * iR +
                      2 * (1 - 3 * ez * ez) * iR3 - 2 * (1 - 5 * ez * ez) * iR5) / 3.0;
  const double f2 = -(3 * (1 - 6 * hh * (1 - hh) * ez * ez) * iR -
                      6 * (1 - 5 * ez * ez) * iR3 + 10 * (1 - 7 * ez * ez) * iR5) / 3.0;
  const double f3 = ez *
                    (3 * hh * (1 - 6 * (1 - hh) * ez * ez) * iR -
                     6 * (1 - 5 * ez * ez) * iR3 + 10 * (2 - 7 * ez * ez) * iR5) *
                    2.0 / 3.0;
  const double f4 = ez * (3 * hh * iR - 10 * iR5) * 2.0 / 3.0;
  const double f5 = -(3 * hh * hh * ez * ez * iR + 3 * ez * ez * iR3 +
                      (2 - 15 * ez * ez) * iR5) * 4.0 / 3.0;
  M[0] += f1 + f2 * ex * ex;
  M[1] += f2 * ex * ey;
  M[2] += f2 * ex * ez + f3 * ex;
  M[3] += f2 * ey * ex;
  M[4] += f1 + f2 * ey * ey;
  M[5] += f2 * ey * ez + f3 * ey;
  M[6] += f2 * ez * ex + f4 * ex;
  M[7] += f2 * ez * ey + f4 * ey;
  M[8] += f1 + f2 * ez * ez + f3 * ez + f4 * ez + f5;
}

// block for the ordered roles (lo <= hi):  r_lo - r_hi,  h = z_hi / a
__device__ __forceinline__ void rbl_block_ref(const RblParams &P, bool wall, double xl,
                                              double yl, double zl, double xh, double yh,
                                              double zh, bool self, double *b, unsigned &flags)
{
  const double rx = xl - xh, ry = yl - yh, rz = zl - zh;
  double s[6];
  rbl_rpy_ref(rx, ry, rz, self, P.inv_a, s, flags);
  b[0] = s[0]; b[1] = s[1]; b[2] = s[2];
  b[3] = s[1]; b[4] = s[3]; b[5] = s[4];
  b[6] = s[2]; b[7] = s[4]; b[8] = s[5];
  if (wall) rbl_wall_ref(rx / P.a, ry / P.a, (rz + 2 * zh) / P.a, b, self, zh / P.a, flags);
}
#pragma clang fp contract(fast)
