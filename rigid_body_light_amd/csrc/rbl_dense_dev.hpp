// rbl_dense_dev.hpp -- device-side pieces shared by the dense kernels (rbl_dense.hip) and the dataflow tile factorisation
// (rbl_tilechol.hip): the in-register 32 x 32 pivot-block factorisation, buffer-descriptor loads / stores and the body of the
// one-workgroup diagonal-block factorisation.  Everything here is internal linkage; include inside an anonymous namespace user.
#pragma once
#include "rbl_internal.hpp"

#include <type_traits>

namespace {

constexpr int IB = 32;    // inner step width

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- potf2: one wavefront factors the IB x IB diagonal block at (k,k) ------------------
// Everything in registers, no LDS on the dependency chain: lane r < 32 holds row r of the block
// (x[c] = A[r][c]), lane 32 + m holds the unknowns of L y = e_m (x = e_m), and BOTH half-waves run
// the same fully unrolled right-looking recurrence
//     x[c] *= 1/L_cc ;   x[c'] -= x[c] * L[c'][c]   (c' > c)
// with the uniform L[c'][c] fetched by v_readlane from lane c': for the low half this is the
// Cholesky update of row r, for the high half it is forward substitution -- so L_kk^-1 (needed by
// the MFMA triangular solves) comes out of the same instruction stream for free.  Pivots by
// v_rsq_f64 + Newton (no sqrt/div chain).  Leaves Linv (IB x IB, row-major [c][m]) in Y.
// (Round 2, measured and dropped: multipliers of columns c + 3.. through LDS -- one write, broadcast reads -- instead of
// two v_readlane each: 1.5x SLOWER, 50 x 486 batch 0.58 -> 0.90 ms and n = 24 300 50.7 -> 46 TFLOP/s; the LDS round
// trip lands on the pivot chain, the v_readlane pairs do not.)
// t = lane (0..63).  Returns true when a pivot was not positive.
__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_row(double v)      // DPP move of both halves of a double (controls within a 16-lane row)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ bool potf2_wave(double *__restrict__ A, long n, long k, int nb,
                                           double (*Y)[IB + 1], int t)
{
  const int i = t & (IB - 1);
  const bool hi = t >= IB;
  const bool live = !hi && i < nb;
  double x[IB];
#pragma unroll
  for (int c = 0; c < IB; ++c) {   // rows/columns beyond nb (ragged last block): identity
    double v = (c == i) ? 1.0 : 0.0;
    if (live && c < nb && c <= i) v = A[(size_t)(k + c) * n + (k + i)];
    x[c] = v;
  }
  bool bad = false;
#pragma unroll
  for (int c = 0; c < IB; ++c) {
    const double d = readlane_f64(x[c], c);              // pivot, uniform
    bad = bad || !(d > 0.0);
    x[c] *= rbl_rsqrt(d);                                // L[r][c]  /  y_c
#pragma unroll
    for (int cp = c + 1; cp < IB; ++cp) x[cp] = __builtin_fma(-x[c], readlane_f64(x[c], cp), x[cp]);
  }
  if (!hi) {
#pragma unroll
    for (int c = 0; c < IB; ++c)
      if (live && c < nb && c <= i) A[(size_t)(k + c) * n + (k + i)] = x[c];
  } else {
#pragma unroll
    for (int r = 0; r < IB; ++r) Y[r][i] = x[r];         // Linv[r][m = i]
  }
  return bad;
}

// ---- diagonal block of an outer panel: ONE workgroup factors the whole pw x pw block -----------
// The right-looking IB-steps (potf2 -> trsm -> rank-IB update) of the block run inside one 8-wave
// workgroup with __syncthreads() between phases instead of ~3 kernel launches per step: on the
// critical path of the factorisation a launch boundary costs tens of microseconds (queueing behind
// the big trailing update + write-back of dirtied lines), a workgroup barrier ~1 us.
// The block (<= 2 MB) stays in L2 of this CU's XCD; the phases are latency-bound (one CU), so
//  * every 32x32 tile update issues ALL its loads (operands and the C entries it will overwrite)
//    before the first MFMA: one memory round trip per tile (8 waves x 256 VGPRs make room for that);
//  * inner lookahead: the update of step s does the tiles of block column s+1 first; then wave 0
//    factors diagonal block s+1 (serial, ~20 us) WHILE waves 1..7 finish the rest of the update.
// trsm and update are MFMA products (transposed tiles, as in the other kernels).  Also writes every
// L_kk^-1 to LinvAll[step] for k_trsm_tall.

typedef unsigned int rbl_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  const rbl_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}
typedef unsigned int rbl_u4 __attribute__((ext_vector_type(4)));
typedef double rbl_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rbl_d2 buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)   // 16 B: two rows
{
  const rbl_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return (rbl_d2){__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z)};
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double d)
{
  const rbl_u2 v = {(unsigned)__double2loint(d), (unsigned)__double2hiint(d)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}

// PBW: waves of the workgroup (8 in k_potrf_block, 4 inside the tile kernel of rbl_tilechol.hip); Y: IB x (IB + 1) doubles of LDS
template <int PBW>
__device__ __forceinline__ void potrf_block_body(double *__restrict__ A, long ld, long k, int pw,
                                                 double *__restrict__ LinvAll, unsigned *err, double (*Y)[IB + 1])
{
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // provably wave-uniform: tile indices live in SGPRs
  const int l15 = lane & 15, l4 = lane >> 4;
  const long pend = k + pw;
  const int nsteps = (pw + IB - 1) / IB;

  auto factor_diag = [&](int s) {      // wave 0 only
    const long kk = k + (long)s * IB;
    const int nb = (int)((pend - kk < IB) ? (pend - kk) : IB);
    const bool bad = potf2_wave(A, ld, kk, nb, Y, lane);
    if (bad && lane == 0) atomicOr(err, (unsigned)RBL_FLAG_NOT_SPD);
    for (int e = lane; e < IB * IB; e += 64) {
      const int c = e & (IB - 1), r = e >> 5;
      LinvAll[(size_t)s * IB * IB + r * IB + c] = Y[r][c];
    }
  };

  // Addressing: the block (columns k..pend-1, < 513 columns, < 2^31 bytes) through ONE buffer descriptor;
  // an entry (col, row) = scalar byte offset (col - k - (lane>>4)) ld 8  +  per-lane byte offset
  // ((lane>>4) ld + row) 8, so a tile's 48 loads share 2-4 offset registers instead of 48 address pairs.
  const unsigned ldb = (unsigned)ld * 8u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      A + (size_t)k * (size_t)ld, (short)0, (int)(((size_t)(pw - 1) * (size_t)ld + (size_t)pend) * 8), 0x00020000);
  auto lane_off = [&](long row0) -> unsigned {     // clamped row (ragged last panel) + this lane's column shift
    const long r = row0 + l15;
    return (unsigned)l4 * ldb + 8u * (unsigned)(r < pend ? r : pend - 1);
  };
  auto col_off = [&](long col) -> unsigned { return (unsigned)(col - k) * ldb; };   // uniform

  // C[i0.., j0..] -= P[i0..] P[j0..]^T with P = columns kk..kk+31 (already solved), 32x32 tile
  auto update_tile = [&](long kk, long i0, long j0) {
    const unsigned oi[2] = {lane_off(i0), lane_off(i0 + 16)}, oj[2] = {lane_off(j0), lane_off(j0 + 16)};
    double av[IB / 4][2], bv[IB / 4][2], cv[2][2][4];
#pragma unroll
    for (int ks = 0; ks < IB / 4; ++ks) {
      const unsigned so = col_off(kk + 4 * ks);
      av[ks][0] = buf_ld(rs, oj[0], so); av[ks][1] = buf_ld(rs, oj[1], so);
      bv[ks][0] = buf_ld(rs, oi[0], so); bv[ks][1] = buf_ld(rs, oi[1], so);
    }
    const bool ragged = j0 + 32 > pend;                        // uniform: tile sticks out of the block (last panel)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long cb = j0 + 16 * tj + 4 * v;                  // uniform column of lane group 0
        if (!ragged) {
          const unsigned so = col_off(cb);
          cv[tj][0][v] = buf_ld(rs, oi[0], so); cv[tj][1][v] = buf_ld(rs, oi[1], so);
        } else {                                               // per-lane clamped column, no scalar part
          const long cl = (cb + l4 < pend) ? cb + l4 : pend - 1;
          const unsigned sh = col_off(cl) - (unsigned)l4 * ldb;
          cv[tj][0][v] = buf_ld(rs, oi[0] + sh, 0u); cv[tj][1][v] = buf_ld(rs, oi[1] + sh, 0u);
        }
      }
    double4_t acc[2][2];   // [tj][ti]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < IB / 4; ++ks) {
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][0], bv[ks][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][0], bv[ks][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][1], bv[ks][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][1], bv[ks][1], acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long cb = j0 + 16 * tj + 4 * v;
        const unsigned so = col_off(cb);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
          if (i0 + 16 * ti + l15 < pend && cb + l4 < pend) buf_st(rs, oi[ti], so, cv[tj][ti][v] - acc[tj][ti][v]);
      }
  };

  if (wave == 0) factor_diag(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const long kk = k + (long)s * IB;
    const int nb = (int)((pend - kk < IB) ? (pend - kk) : IB);
    const long rem0 = kk + nb;
    if (rem0 >= pend) break;            // block-uniform: nothing below / right of this step
    const int nrt = (int)((pend - rem0 + 31) / 32);   // 32-row tiles below the diagonal block
    // ---- trsm: X = A[rows, kk:kk+32] * Linv^T, one 32-row tile per wave-iteration ------------
    for (int g = wave; g < nrt; g += PBW) {
      const long i0 = rem0 + 32L * g;
      const unsigned oi[2] = {lane_off(i0), lane_off(i0 + 16)};
      double bv[IB / 4][2];
#pragma unroll
      for (int ks = 0; ks < IB / 4; ++ks) {
        const unsigned so = col_off(kk + 4 * ks);
        bv[ks][0] = buf_ld(rs, oi[0], so); bv[ks][1] = buf_ld(rs, oi[1], so);
      }
      double4_t acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < IB / 4; ++ks) {
        const int m = 4 * ks + l4;
        const double a0 = Y[l15][m], a1 = Y[16 + l15][m];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv[ks][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv[ks][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv[ks][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv[ks][1], acc[1][1], 0, 0, 0);
      }
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const unsigned so = col_off(kk + 16 * tc + 4 * v);
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
            if (i0 + 16 * ti + l15 < pend) buf_st(rs, oi[ti], so, acc[tc][ti][v]);
        }
    }
    __syncthreads();
    // ---- rank-IB update, part 1: block column s+1 (tiles (bi, 0)), all waves ---------------------
    for (int bi = wave; bi < nrt; bi += PBW) update_tile(kk, rem0 + 32L * bi, rem0);
    __syncthreads();
    // ---- part 2: wave 0 factors diagonal block s+1; waves 1.. update the remaining lower tiles ----
    if (wave == 0) {
      factor_diag(s + 1);
    } else {
      int idx = 0;
      for (int bi = 1; bi < nrt; ++bi)
        for (int bj = 1; bj <= bi; ++bj, ++idx)
          if (idx % (PBW - 1) == wave - 1) update_tile(kk, rem0 + 32L * bi, rem0 + 32L * bj);
    }
    __syncthreads();
  }
}



}  // namespace
