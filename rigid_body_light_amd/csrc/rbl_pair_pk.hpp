// rbl_pair_pk.hpp -- packed single-precision pair arithmetic for the RELAXED mobility product (see below).
#pragma once
#include <hip/hip_runtime.h>

// ---------------------------------------------------------------------------
// RELAXED-precision form of rbl_pair_sym for far tile pairs (no overlap possible, i != j): the same algebra in packed
// single precision -- the two rows a lane owns travel in the two halves of 64-bit registers (v_pk_fma_f32: two pairs per
// instruction), v_rsq_f32 needs no Newton step.  ~38 VALU instructions per unordered pair instead of ~75.
// NOT used by default: an inexact Krylov method tolerates a product error of (tolerance / current residual), so the
// library's GMRES may switch to it once its residual estimate is small (RBL_OPT_RELAXED_KRYLOV); a product through this form
// agrees with the fp64 one to ~1e-6 relative.  Coordinates: x, y and z RELATIVE to an origin near the rows (so that
// single precision resolves the short distances), two_z0 = 2 x that origin's height, for R_z = z_i + z_j.
// ---------------------------------------------------------------------------
typedef float rbl_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rbl_f2 rbl_fma2(rbl_f2 a, rbl_f2 b, rbl_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ rbl_f2 rbl_splat(float x) { return (rbl_f2){x, x}; }
__device__ __forceinline__ rbl_f2 rbl_rsq2(rbl_f2 x) { return (rbl_f2){__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }

// scalar coefficients of the pair block for the two rows of a lane (see rbl_wall_coeffs for the algebra):
//   WALL: M = cF I + beta dl dl^T + gxz dl z^T + gzx z dl^T + (mzz - cF) z z^T ;  free space: M = cF I + beta d d^T
struct RblPkCoef {
  rbl_f2 dx, dy, dz, cF, beta, gxz, gzx, mzz;
};

template <bool WALL>
__device__ __forceinline__ RblPkCoef rbl_pk_coef(rbl_f2 xi, rbl_f2 yi, rbl_f2 zi, float xj, float yj, float zj, float two_z0)
{
  RblPkCoef K;
  const rbl_f2 dx = xi - rbl_splat(xj), dy = yi - rbl_splat(yj), dz = zi - rbl_splat(zj);
  const rbl_f2 q = rbl_fma2(dy, dy, dx * dx);
  const rbl_f2 r2 = rbl_fma2(dz, dz, q);
  const rbl_f2 invr = rbl_rsq2(r2);
  const rbl_f2 invr2 = invr * invr;
  const rbl_f2 s3 = invr2 * invr;
  const rbl_f2 A = rbl_fma2(s3, rbl_splat(2.0f / 3.0f), invr);
  const rbl_f2 Bc = rbl_fma2(s3, rbl_splat(-2.0f), invr) * invr2;
  K.dx = dx; K.dy = dy; K.dz = dz;
  if (!WALL) {
    K.cF = A; K.beta = Bc; K.gxz = Bc; K.gzx = Bc; K.mzz = A;
    return K;
  }
  // wall part: rbl_wall_coeffs<UNIT = true> term by term
  const rbl_f2 Rz = (zi + rbl_splat(zj)) + rbl_splat(two_z0);
  const rbl_f2 R2 = rbl_fma2(Rz, Rz, q);
  const rbl_f2 w = rbl_rsq2(R2);
  const rbl_f2 u = w * w;
  const rbl_f2 v = rbl_fma2(-q, u, rbl_splat(1.0f));
  const rbl_f2 rr = r2 * u;
  const rbl_f2 t1 = rbl_fma2(u, rbl_fma2(v, rbl_splat(-10.0f / 3.0f), rbl_splat(2.0f / 3.0f)), rbl_fma2(v, rbl_splat(2.0f), rbl_splat(-2.0f / 3.0f)));
  const rbl_f2 b1 = rbl_fma2(u, t1, rbl_fma2(rbl_splat(0.5f), rr, rbl_splat(-1.5f)));
  const rbl_f2 t2 = rbl_fma2(u, rbl_fma2(v, rbl_splat(70.0f / 3.0f), rbl_splat(-10.0f / 3.0f)), rbl_fma2(v, rbl_splat(-10.0f), rbl_splat(2.0f)));
  const rbl_f2 sixgk = rbl_fma2(-rr, rbl_splat(1.5f), rbl_splat(1.5f));
  const rbl_f2 b2p = rbl_fma2(u, t2, sixgk);
  const rbl_f2 uu = u * u;
  const rbl_f2 T1p = rbl_fma2(uu, rbl_splat(-20.0f / 3.0f), b2p);
  const rbl_f2 w3 = w * u;
  const rbl_f2 Bw = Bc - w3;
  K.cF = rbl_fma2(w, b1, A);
  K.beta = rbl_fma2(b2p, w3, Bw);
  const rbl_f2 gm = dz * Bw, gdw = Rz * w3;
  K.gxz = rbl_fma2(-T1p, gdw, gm);
  K.gzx = rbl_fma2(T1p, gdw, gm);
  const rbl_f2 c = rbl_fma2(uu, rbl_fma2(v, rbl_splat(20.0f), rbl_splat(-8.0f / 3.0f)), -(v * rbl_fma2(rbl_splat(4.0f), u, b2p)));
  K.mzz = rbl_fma2(w, c, rbl_fma2(gm, dz, K.cF));
  return K;
}

// u += M F  (TRANSPOSE = false) or u += M^T F (true) for the two rows' blocks
template <bool WALL, bool TRANSPOSE>
__device__ __forceinline__ void rbl_pk_apply(const RblPkCoef &K, rbl_f2 Fx, rbl_f2 Fy, rbl_f2 Fz, rbl_f2 &ux, rbl_f2 &uy, rbl_f2 &uz)
{
  if (!WALL) {
    const rbl_f2 tB = K.beta * rbl_fma2(K.dz, Fz, rbl_fma2(K.dy, Fy, K.dx * Fx));
    ux = rbl_fma2(K.cF, Fx, rbl_fma2(tB, K.dx, ux));
    uy = rbl_fma2(K.cF, Fy, rbl_fma2(tB, K.dy, uy));
    uz = rbl_fma2(K.cF, Fz, rbl_fma2(tB, K.dz, uz));
    return;
  }
  const rbl_f2 g_lat = TRANSPOSE ? K.gzx : K.gxz, g_z = TRANSPOSE ? K.gxz : K.gzx;
  const rbl_f2 p = rbl_fma2(K.dy, Fy, K.dx * Fx);
  const rbl_f2 l = rbl_fma2(K.beta, p, g_lat * Fz);
  ux = rbl_fma2(K.cF, Fx, rbl_fma2(l, K.dx, ux));
  uy = rbl_fma2(K.cF, Fy, rbl_fma2(l, K.dy, uy));
  uz = rbl_fma2(K.mzz, Fz, rbl_fma2(g_z, p, uz));
}

template <bool WALL>
__device__ __forceinline__ void rbl_pair_sym_pk(rbl_f2 xi, rbl_f2 yi, rbl_f2 zi, rbl_f2 Fix, rbl_f2 Fiy, rbl_f2 Fiz, float xj,
                                                float yj, float zj, float Fjx, float Fjy, float Fjz, float two_z0, rbl_f2 &uix,
                                                rbl_f2 &uiy, rbl_f2 &uiz, rbl_f2 &ujx, rbl_f2 &ujy, rbl_f2 &ujz)
{
  const RblPkCoef K = rbl_pk_coef<WALL>(xi, yi, zi, xj, yj, zj, two_z0);
  rbl_pk_apply<WALL, false>(K, rbl_splat(Fjx), rbl_splat(Fjy), rbl_splat(Fjz), uix, uiy, uiz);   // U_i += M F_j
  rbl_pk_apply<WALL, true>(K, Fix, Fiy, Fiz, ujx, ujy, ujz);                                      // U_j += M^T F_i
}
