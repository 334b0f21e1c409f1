// rbl_tilechol.hip -- per-body Cholesky factors AND their explicit inverses for LARGE bodies (3 N_blb > 512: shell_N_642 / 2562)
// as ONE dataflow launch over 128 x 128 tiles (reference Block_diag_invM, c_rigid_obj.cpp:461-487: one dense mobility per body,
// inverted; here L L^T = M_b and X = L^-1 so that both the substitution and the explicit-inverse applications are served).
//
// Why not the panel kernels of rbl_dense.hip: at n ~ 2 000 the batched right-looking factorisation is 12 launches whose panel
// kernels (one workgroup per body for the diagonal block, a latency chain) are 42 % of the time, whose rank-512 updates re-read
// every body's panel from HBM (a body's 6 MB panel does not stay in an XCD's 4 MB L2 when 200 bodies interleave) and whose tails
// leave the chip part-empty twelve times -- 21-24 TFLOP/s.  Here a body's matrix is cut into NT = ceil(n / 128) tile rows and
//     CHOL(i, j), i >= j :  T = A_ij - sum_{k<j} L_ik L_jk^T ;  i == j: L_jj = chol(T) ;  i > j: L_ij = T L_jj^-T
//     INV (j, i), i >= j :  Y = L^-T (upper triangular, Y = X^T):  Y_jj = L_jj^-T ,  Y_ji = -(sum_{j<=k<i} Y_jk L_ik^T) L_ii^-T
// are TASKS drawn in dependency order from eight queues (one per XCD: body b lives on queue b % 8, so a body's tiles meet in
// ONE L2; a workgroup serves the queue of the XCD it runs on -- HW_REG_XCC_ID -- and helps the others when that one is dry).
// A tile is one long MFMA product (K = 128 j columns at once: the staging pipeline of k_syrk_mfma, C read and written once),
// then a 128-column triangular solve on the matrix cores against the 32 x 32 inverses the diagonal task leaves.  Tasks of stage s
// are CHOL column s (NT - s tiles) and INV column s (s + 1 tiles): NT + 1 tiles per body and stage, the same all the way down.
// Hand-offs between workgroups follow the guide's agent-scope recipe: plain stores, every wave's vmcnt(0), workgroup barrier,
// ONE release fence, a relaxed atomic add on the row's counter; consumers poll that counter relaxed and fence-acquire once.
// No task waits on a later task of its queue, queues are claimed in order, so a waiting workgroup only ever waits for RUNNING
// ones: no residency assumption, no deadlock; every wait is bounded (RBL_FLAG_INTERNAL and a clean exit when it runs out).
#include "rbl_dense_dev.hpp"

namespace {

constexpr int TC = 128;          // tile edge
constexpr int TKC = 16;          // K columns per pipeline stage
constexpr int TLDP = 144;        // LDS column stride (doubles): conflict-free fragment reads, as in k_syrk_mfma
constexpr int TQS = 32;          // unsigned words between two queue heads (one 128-byte line each)
constexpr int TCU = 2048;        // per-CU words (XCC 3 bits, SE / SH / CU 8 bits of HW_ID)
constexpr unsigned TGRP = 4;     // bodies of a queue that share an iteration's slots (see the queue order in k_tile_chol)
constexpr unsigned SPIN_LIMIT = 4u << 20;   // polls of ~0.5 us before a wait gives up (seconds: a hang must end by itself)

// Diagnostic build only (RBL_EXTRA_FLAGS=-DRBL_TILE_PROF, tools/tile_phase_profile.py): shader-clock stamps around the phases of a task,
// summed by thread 0 of every workgroup.  No stamp exists in the normal build.
#ifdef RBL_TILE_PROF
__device__ unsigned long long g_tile_prof[16];
#define TP(i) { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); atomicAdd(&g_tile_prof[i], now_ - tp_prev); tp_prev = now_; } }
// ... and the timeline of body 0: when (100 MHz wall clock) each of its tasks was claimed and published, [kind][row tile][column tile][claim | publish]
__device__ unsigned long long g_tile_line[2][32][32][2];
#define TLINE(w) { if (threadIdx.x == 0 && b == 0 && ti_ < 32 && tj_ < 32) g_tile_line[chol ? 0 : 1][ti_][tj_][w] = __builtin_amdgcn_s_memrealtime(); }
#else
#define TP(i)
#define TLINE(w)
#endif

struct TileChol {
  double *A; long n; long strideA;            // the matrices (column-major, ld = n), factored in place
  double *Y; long ldy; long strideY;          // Y = L^-T = X row-major (layout XU of the explicit inverses), NULL: factor only
  double *Linv; long strideL;                 // 32 x 32 inverses of the diagonal blocks, [body][step][1024]
  double *W;                                  // 128 x 128 inverses of the diagonal TILES, [body][NT][128 * 128] (column-major, zero-padded)
  unsigned *cntL, *cntY;                      // finished tiles per tile row of L / of Y, [body][NT]
  unsigned *heads;                            // eight queue heads, TQS words apart
  unsigned *abort_;                           // set when a wait ran out
  unsigned *err;
  int batch, NT;
  unsigned *cu_busy;                          // small batches with the inverse: per physical CU, chain tiles at work there (NULL: off)
};

__device__ __forceinline__ unsigned ld_relaxed(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every wave for itself: poll (relaxed, uniform address) until *p >= need, then ONE agent-scope acquire
__device__ __forceinline__ void wait_ge(unsigned *p, unsigned need, const TileChol &P)
{
  unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed(p));
  for (unsigned spins = 0; v < need; ++spins) {
    __builtin_amdgcn_s_sleep(8);
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_relaxed(p));
    if ((spins & 1023u) == 1023u && (spins >= SPIN_LIMIT || __builtin_amdgcn_readfirstlane((int)ld_relaxed(P.abort_)))) {
      if ((threadIdx.x & 63) == 0) {
        __hip_atomic_store(P.abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(P.err, (unsigned)RBL_FLAG_INTERNAL);
      }
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// acc[tj][ti] (the wave's 64 x 64 part of the 128 x 128 tile, transposed MFMA tiles as in k_syrk_mfma) =
//   sum over the K columns starting at Ip / Jp of  I[rowI + .., k] J[rowJ + .., k]
// Staging pipeline of k_syrk_mfma: 16-column slabs global -> registers -> LDS, two LDS buffers, two register sets, the loads of
// stage s + 3 issued in stage s.  K is a multiple of 128 (>= 128); only the first KvI / KvJ columns of the two panels exist (the
// buffer descriptors end there: loads beyond return 0 -- the ragged last tile column).  wait(c): called before the first load of
// the c-th 128-column chunk is issued.
// Which of the wave's 4 x 4 MFMA tiles a product needs (all wave-uniform): nti / ntj = 16-row / 16-column tiles that reach into
// the matrix (the ragged last tile row holds 6 of 128 rows at n = 1926: 1/8 of a tile's matrix-core work instead of all of it);
// tri: X = T W^T with W lower triangular -- the K slab of stage s only reaches output column tiles 4 wj + tj >= s;
// lowtri: a 64 x 64 part ON the diagonal of a diagonal tile -- only its lower triangle is ever read.
struct TileMask {
  int nti, ntj;
  bool tri, lowtri;
  __device__ bool full() const { return nti == 4 && ntj == 4 && !tri && !lowtri; }
};

// MASKED = false: every MFMA tile of an active wave (the hot form: mk only says whether the wave takes part at all)
template <bool MASKED, class WaitFn>
__device__ __forceinline__ void tile_gemm(const double *Ip, long ldI, long rowI, long nrI, const double *Jp, long ldJ, long rowJ, long nrJ,
                                          int K, int KvI, int KvJ, double (*sI)[TKC * TLDP], double (*sJ)[TKC * TLDP], double4_t (&acc)[4][4],
                                          const TileMask mk, WaitFn &&wait)
{
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int wi = wave & 1, wj = wave >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int lrow = (t & 63) * 2, lcg = (t >> 6) * 4;
  long gi = rowI + lrow; if (gi >= nrI) gi = nrI - 2;      // rows beyond the matrix feed accumulators that are never stored
  long gj = rowJ + lrow; if (gj >= nrJ) gj = nrJ - 2;
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Ip), (short)0, (int)((size_t)KvI * (size_t)ldI * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsJ = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Jp), (short)0, (int)((size_t)KvJ * (size_t)ldJ * 8), 0x00020000);
  const unsigned ldbI = (unsigned)ldI * 8u, ldbJ = (unsigned)ldJ * 8u;
  const unsigned vI = (unsigned)lcg * ldbI + 8u * (unsigned)gi, vJ = (unsigned)lcg * ldbJ + 8u * (unsigned)gj;
  rbl_d2 rI[2][4], rJ[2][4];
  const int nst = K / TKC;
  auto gload = [&](auto set, int stg) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rI[set()][q] = buf_ld2(rsI, vI, (unsigned)(stg * TKC + q) * ldbI);
      rJ[set()][q] = buf_ld2(rsJ, vJ, (unsigned)(stg * TKC + q) * ldbJ);
    }
  };
  auto lwrite = [&](auto set, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<rbl_d2 *>(&sI[buf][(lcg + q) * TLDP + lrow]) = rI[set()][q];
      *reinterpret_cast<rbl_d2 *>(&sJ[buf][(lcg + q) * TLDP + lrow]) = rJ[set()][q];
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  wait(0);
  gload(S0{}, 0);
  lwrite(S0{}, 0);
  gload(S1{}, 1);
  gload(S0{}, 2);
  __syncthreads();
  const bool active = mk.nti > 0 && mk.ntj > 0;
  auto compute = [&](int cur, int s_) {
    if (!active) return;
    const double *fi = &sI[cur][l4 * TLDP + wi * 64 + l15];
    const double *fj = &sJ[cur][l4 * TLDP + wj * 64 + l15];
    if (!MASKED) {
#pragma unroll
      for (int ks = 0; ks < TKC / 4; ++ks) {
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = fj[ks * 4 * TLDP + 16 * q]; bv[q] = fi[ks * 4 * TLDP + 16 * q]; }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
          for (int ti = 0; ti < 4; ++ti)
            acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tj], bv[ti], acc[tj][ti], 0, 0, 0);
      }
      return;
    }
    const int tj_lo = mk.tri ? (s_ - 4 * wj > 0 ? s_ - 4 * wj : 0) : 0;
    if (tj_lo >= mk.ntj) return;
#pragma unroll
    for (int ks = 0; ks < TKC / 4; ++ks) {
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { av[q] = fj[ks * 4 * TLDP + 16 * q]; bv[q] = fi[ks * 4 * TLDP + 16 * q]; }
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
        if (tj >= tj_lo && tj < mk.ntj) {
#pragma unroll
          for (int ti = 0; ti < 4; ++ti)
            if (ti < mk.nti && (!mk.lowtri || ti >= tj))
              acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tj], bv[ti], acc[tj][ti], 0, 0, 0);
        }
    }
  };
  auto stage = [&](int s_, auto nset) {
    if (s_ + 1 < nst) lwrite(nset, (s_ + 1) & 1);
    if (s_ + 3 < nst) {
      if (((s_ + 3) & 7) == 0) wait((s_ + 3) >> 3);       // first slab of the next 128-column chunk
      gload(nset, s_ + 3);
    }
    compute(s_ & 1, s_);
    __syncthreads();
  };
  for (int s2 = 0; s2 < nst; s2 += 2) {                    // nst = 8 (K / 128): even
    stage(s2, S1{});
    stage(s2 + 1, S0{});
  }
}

// X = T W^T IN PLACE in the accumulators (W = L_cc^-1 lower triangular, 128 x 128 column-major at Wt, zero-padded): on entry
// acc[tj][ti] holds T (element (col 16 tj + l4 + 4 v, row 16 ti + l15) of the wave's 64 x 64 part), on exit X.  The 16-column slabs
// of T go DOWN from 7 to 0: slab s is written to LDS by the two waves that hold those columns (wj = s >> 2) straight from
// their accumulator registers, whose tile is dead from then on (X's column tile t only collects slabs <= t), so that tile's
// registers restart at zero as X -- no second accumulator set, no trip of T through memory.  The W slabs ride the usual
// global -> registers -> LDS prefetch.  One barrier per slab; a wave multiplies slab s into its column tiles 4 wj + tj >= s only.
__device__ __forceinline__ void tile_solve(double4_t (&acc)[4][4], const double *Wt, double (*sT)[TKC * TLDP], double (*sW)[TKC * TLDP],
                                           bool active)
{
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int wi = wave & 1, wj = wave >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int lrow = (t & 63) * 2, lcg = (t >> 6) * 4;
  rbl_d2 rW[2][4];
  auto gload = [&](auto set, int slab) {
#pragma unroll
    for (int q = 0; q < 4; ++q) rW[set()][q] = *reinterpret_cast<const rbl_d2 *>(Wt + (size_t)(slab * TKC + lcg + q) * TC + lrow);
  };
  auto lwrite = [&](auto set, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<rbl_d2 *>(&sW[buf][(lcg + q) * TLDP + lrow]) = rW[set()][q];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  gload(S0{}, 7);
  gload(S1{}, 6);
#pragma unroll
  for (int p = 0; p < 8; ++p) {                            // slab s = 7 - p
    const int s_ = 7 - p, buf = p & 1, tjs = s_ & 3;
    if (p & 1) lwrite(S1{}, buf); else lwrite(S0{}, buf);
    if (p + 2 < 8) { if (p & 1) gload(S1{}, s_ - 2); else gload(S0{}, s_ - 2); }
    if (wj == (s_ >> 2)) {                                 // this wave holds T's columns 16 s .. 16 s + 15
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          sT[buf][(l4 + 4 * v) * TLDP + wi * 64 + 16 * ti + l15] = acc[tjs][ti][v];
          acc[tjs][ti][v] = 0.0;
        }
    }
    __syncthreads();
    if (active && 4 * wj + 3 >= s_) {
      const double *fi = &sT[buf][l4 * TLDP + wi * 64 + l15];
      const double *fj = &sW[buf][l4 * TLDP + wj * 64 + l15];
      const int tj_lo = s_ - 4 * wj;
#pragma unroll
      for (int ks = 0; ks < TKC / 4; ++ks) {
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = fj[ks * 4 * TLDP + 16 * q]; bv[q] = fi[ks * 4 * TLDP + 16 * q]; }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          if (tj >= tj_lo) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
              acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tj], bv[ti], acc[tj][ti], 0, 0, 0);
          }
      }
    }
  }
  __syncthreads();                                         // the LDS buffers are free again
}

__device__ __forceinline__ int tiles16(long first, long limit)       // 16-wide MFMA tiles of [first, first + 64) that begin below `limit`
{
  const long r = limit - first;
  return r <= 0 ? 0 : (r >= 64 ? 4 : (int)((r + 15) / 16));
}

// W = L11^-1 for the factored diagonal tile L11 = A[k0 .. k0 + pw) (pw <= 128) from its 32 x 32 blocks and the inverses Li of its
// diagonal blocks (both left by potrf_block_body): block column c of W belongs to wave c,
//     W_cc = Linv_cc ,   W_rc = -Linv_rr sum_{c <= k < r} L_rk W_kc   (r > c),
// the operand loads of a block row in flight together (<= 3 memory round trips), a chain of <= 9 products of
// 32 x 32 x 32 on the matrix cores whose results feed the next product straight from the accumulator registers (the f64 C/D
// layout is the B-operand layout).  W goes to Wt (128 x 128, column-major, ld 128), zero above the diagonal and beyond pw:
// the triangular solves of the column's other tiles are then ONE pipelined product  X = T W^T  (tile_gemm with K = 128)
// instead of a latency chain of four dependent block steps per tile (a third of all workgroup time before).
__device__ __forceinline__ void tile_winv(const double *A, long ld, long k0, int pw, const double *Li, double *Wt)
{
  const int t = threadIdx.x;
  const int c = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nbk = (pw + IB - 1) / IB;                      // valid 32-blocks of the tile
  const long last = k0 + pw - 1;
  auto cl = [last](long x) -> long { return x < last ? x : last; };
  double4_t W[4][2][2];                                    // [r - c][tm][tn]: block W_rc in the C/D layout
#pragma unroll
  for (int dr = 0; dr < 4; ++dr)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) W[dr][a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
  if (c < nbk) {
    // W_cc = Linv_cc: element (row 16 tm + l4 + 4 v, col 16 tn + l15)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int v = 0; v < 4; ++v) W[0][tm][tn][v] = Li[(size_t)c * IB * IB + (16 * tm + l4 + 4 * v) * IB + 16 * tn + l15];
#pragma unroll
    for (int dr = 1; dr < 4; ++dr) {
      const int r = c + dr;
      if (dr < 4 - c && r < nbk) {
        // the operands of this block row, all loads in flight together: L_rk (c <= k < r) and Linv_rr
        double lA[3][2][IB / 4], iA[2][IB / 4];
#pragma unroll
        for (int dk = 0; dk < 3; ++dk)
          if (dk < dr) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
              for (int ks = 0; ks < IB / 4; ++ks)
                lA[dk][hf][ks] = A[(size_t)cl(k0 + 32 * (c + dk) + 4 * ks + l4) * (size_t)ld + cl(k0 + 32 * r + 16 * hf + l15)];
          }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int ks = 0; ks < IB / 4; ++ks) iA[hf][ks] = Li[(size_t)r * IB * IB + (16 * hf + l15) * IB + 4 * ks + l4];
        double4_t S[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) S[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int dk = 0; dk < 3; ++dk)
          if (dk < dr) {
#pragma unroll
            for (int ks = 0; ks < IB / 4; ++ks) {
#pragma unroll
              for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                  S[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(lA[dk][tm][ks], W[dk][ks >> 2][tn][ks & 3], S[tm][tn], 0, 0, 0);
            }
          }
#pragma unroll
        for (int ks = 0; ks < IB / 4; ++ks) {
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              W[dr][tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(-iA[tm][ks], S[ks >> 2][tn][ks & 3], W[dr][tm][tn], 0, 0, 0);
        }
      }
    }
  }
  // the wave's block column of Wt, all four row blocks (zero above the diagonal and beyond the valid blocks)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          double w = 0.0;
#pragma unroll
          for (int dr = 0; dr < 4; ++dr)
            if (r == c + dr) w = W[dr][tm][tn][v];
          Wt[(size_t)(32 * c + 16 * tn + l15) * TC + 32 * r + 16 * tm + l4 + 4 * v] = w;
        }
  }
}

// the finished tile becomes visible to every other workgroup, then its row's counter goes up (guide: Guideline 16, counter form)
__device__ __forceinline__ void publish(unsigned *cnt)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (ROCm 7.2 may drop the fence's own wait: keep this one, in this order)
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// where in an iteration the chain tile (s + 1, s) sits: slot h + 1.  h = NT - 1 puts it LAST: the diagonal tile (s, s) it waits
// for was claimed a whole iteration's worth of tasks earlier (with 25 bodies a queue: 400 tasks, six rounds of the XCD's 64
// slots -- mid-iteration, 200 tasks, the tile sat idle for up to 0.4 ms behind the long products of late diagonal tiles: 6.7 % of
// all workgroup time), and the NEXT diagonal tile, one slot later, meets it only in its last K chunk.
#ifndef RBL_TILE_CRIT_SLOT
#define RBL_TILE_CRIT_SLOT(NT) ((NT) - 1)
#endif

// slot u of iteration s of the queue order described in k_tile_chol -> task; false: an empty slot
__host__ __device__ inline bool tile_task(int NT, int s, int u, bool &chol, int &ti, int &tj)
{
  const int h = RBL_TILE_CRIT_SLOT(NT);                  // the slot before the chain tile (s + 1, s)
  if (s == NT) { if (u > NT - 1) return false; chol = false; ti = u; tj = NT - 1; return true; }   // last iteration: INV(., NT - 1), NT tiles
  if (u == 0) { chol = true; ti = s; tj = s; return true; }
  if (u == h + 1) { if (s + 1 >= NT) return false; chol = true; ti = s + 1; tj = s; return true; }
  const int c = s - 1, v = u - 1 - (u > h + 1 ? 1 : 0);                       // v = 0 .. NT - 2: index into the bulk of column c
  if (c < 0) return false;
  const int noff = NT - c - 2;                                                // off-diagonal tiles of column c below (c + 1, c)
  if (v < noff) { chol = true; ti = c + 2 + v; tj = c; }
  else { chol = false; ti = v - noff; tj = c; }
  return true;
}

__global__ __launch_bounds__(256, 2) void k_tile_chol(const TileChol P)
{
  __shared__ __attribute__((aligned(16))) double sI[2][TKC * TLDP];
  __shared__ __attribute__((aligned(16))) double sJ[2][TKC * TLDP];
  __shared__ int s_task[2];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int wi = wave & 1, wj = wave >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int NT = P.NT;
  const long n = P.n;
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;      // HW_REG_XCC_ID
  const unsigned per = (unsigned)(NT + 1);                 // tasks per body and stage
  auto nbq = [&](int q) -> unsigned { return P.batch > q ? (unsigned)((P.batch - q + 7) / 8) : 0u; };
  const bool with_inv = P.Y != nullptr;
  // this workgroup's physical CU: HW_REG_HW_ID bits 8 .. 15 (CU, SH, SE) under the XCC
  unsigned *const cu_word = P.cu_busy ? P.cu_busy + (((xcc << 8) | (((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xffu)) & (TCU - 1)) : nullptr;

#ifdef RBL_TILE_PROF
  unsigned long long tp_prev = clock64();
#endif
  for (;;) {
    TP(9)                                                  // end-of-task barrier of the previous task
    if (t == 0) {
      int q = -1; unsigned idx = 0;
      if (cu_word)                                         // a chain tile is at work on this CU: leave the SIMDs to it
        for (int n_ = 0; n_ < 1024 && ld_relaxed(cu_word) != 0u; ++n_) __builtin_amdgcn_s_sleep(32);
      if (!ld_relaxed(P.abort_)) {
        for (int a = 0; a < 8 && q < 0; ++a) {
          const int qq = (int)((xcc + (unsigned)a) & 7u);
          const unsigned ntq = nbq(qq) * (unsigned)(NT + 1) * per;
          if (!ntq || ld_relaxed(P.heads + qq * TQS) >= ntq) continue;
          const unsigned got = __hip_atomic_fetch_add(P.heads + qq * TQS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (got < ntq) { q = qq; idx = got; }
        }
      }
      s_task[0] = q; s_task[1] = (int)idx;
    }
    __syncthreads();
    const int q = s_task[0];
    const unsigned idx = (unsigned)s_task[1];
    __syncthreads();
    if (q < 0) break;
    TP(0)                                                  // claim
    // Queue order (the same for every body of the queue, bodies interleaved): iteration s = 0 .. NT holds NT + 1 slots,
    //   slot 0       : the diagonal tile (s, s)                                  -- the critical chain ...
    //   slot NT      : its successor (s + 1, s)                                  -- ... diag(s) -> (s + 1, s) -> diag(s + 1)
    //   other slots  : the BULK of column s - 1, one iteration behind the chain: CHOL(i, s - 1) for i >= s + 1, then INV(., s - 1)
    // so that what a task waits for was claimed a whole iteration (or half of one) earlier and a slot is not held idle while a
    // diagonal tile is factored.  Every task still comes after everything it depends on (see the header): any order with that
    // property is deadlock-free.
    // ... and the bodies of a queue go through an iteration in GROUPS of TGRP: slot-major inside a group (slot 0 of its bodies, slot 1
    // of its bodies, ...), so that a body's chain tile, last slot, comes TGRP x NT tasks -- a full round of the XCD's 64 workgroup
    // slots -- after its diagonal tile instead of right behind it in the queue (body-major, it idled there for the whole product
    // and factorisation of the diagonal tile: 5 % of all workgroup time), while the tiles of a body's column still run side by
    // side and share that column's row panel in the L2.
    const unsigned nb_q = nbq(q);
    const unsigned s_u = idx / (nb_q * per), r_it = idx % (nb_q * per);
    const unsigned grp = r_it / (TGRP * per), r_g = r_it - grp * (TGRP * per);
    const unsigned gsz = (nb_q - grp * TGRP < TGRP) ? nb_q - grp * TGRP : TGRP;
    const unsigned u = r_g / gsz;
    const int b = q + 8 * (int)(grp * TGRP + r_g % gsz), s = (int)s_u;        // s = 0 .. NT
    bool chol; int ti_, tj_;                                                  // CHOL(ti_, tj_) or INV(row tile ti_ of Y, column tile tj_)
    if (!tile_task(NT, s, (int)u, chol, ti_, tj_)) continue;
    if (!chol && !with_inv) continue;
    TLINE(0)
    double *Ab = P.A + (size_t)b * (size_t)P.strideA;
    double *Lib = P.Linv + (size_t)b * (size_t)P.strideL;
    unsigned *cL = P.cntL + (size_t)b * NT, *cY = P.cntY + (size_t)b * NT;
    // ---- both kinds of task as ONE flow:  T = [C] - sum_k I_k J_k^T  into tile (rt, ct) of Om, then  X = T W_ct^T ----
    //   CHOL(i, j): Om = A, (rt, ct) = (i, j), I = L(i, .), J = L(j, .), K columns [0, 128 j), C = the tile itself
    //   INV(jj, i): Om = Y, (rt, ct) = (jj, i), I = Y(jj, .), J = L(i, .), K columns [128 jj, 128 i), C = 0
    const int rt = ti_, ct = tj_;
    double *Om = chol ? Ab : P.Y + (size_t)b * (size_t)P.strideY;
    const long ldo = chol ? n : P.ldy;
    const int kt0 = chol ? 0 : rt, nkt = chol ? ct : ct - rt;                 // first tile column of the K range, its length in tiles
    const long r0 = (long)rt * TC, c0 = (long)ct * TC;
    const long i0 = r0 + wi * 64, j0 = c0 + wj * 64;
    const bool diag = rt == ct;
    const bool on_chain = cu_word != nullptr && chol && (diag || rt == ct + 1);      // diag(s) -> (s + 1, s) -> diag(s + 1): what a small batch waits for
    if (on_chain && t == 0) __hip_atomic_fetch_add(cu_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned *cA = chol ? cL + rt : cY + rt, *cB = chol ? cL + ct : cA;       // counters the K chunks wait on (cA: chunk kc needs >= kc + 1)
    unsigned *cOut = cA;                                                      // ... and the one this tile adds to
    const int pw = (int)((n - c0 < TC) ? (n - c0) : TC);
    double *Wb = P.W + ((size_t)b * NT + ct) * (size_t)(TC * TC);
    const bool upper_idle = chol && diag && wi == 0 && wj == 1;               // the 64 x 64 part above the diagonal of a diagonal tile
    TileMask mk{upper_idle ? 0 : tiles16(i0, n), tiles16(j0, n), false, chol && diag && wi == wj};
    const bool active = mk.nti > 0 && mk.ntj > 0;
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // Everything this task will ever wait for, looked at ONCE, up front (per wave, wave-uniform): the K chunks' counters and the
    // diagonal tile of the column.  All there (the rule, thanks to the queue order): ONE agent-scope acquire for the whole task --
    // an acquire costs microseconds on a busy CU (the guide prices it at 1.7 us x 4 at four blocks a CU) and round 5's first
    // version paid two or three per task: 6.6 % of all workgroup time.  Otherwise the task polls as it goes and acquires after
    // every poll that had to wait.
    const unsigned needD = (chol && diag) ? 0u : (unsigned)ct + 1u;           // finished tiles of L's row ct this task needs at some point
    const unsigned vA0 = ld_relaxed(cA), vB0 = ld_relaxed(cB);               // (ONE read each: a second read may already see more than the other counter holds)
    unsigned haveK = (unsigned)__builtin_amdgcn_readfirstlane((int)(vA0 < vB0 ? vA0 : vB0));
    bool haveD = __builtin_amdgcn_readfirstlane((int)(ld_relaxed(cL + ct) >= needD)) != 0;
    TP(10)                                                 // the counters read
    if (!chol && !haveD) { wait_ge(cL + ct, needD, P); haveD = true; }        // INV reads row ct of L from its first chunk on
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    TP(11)                                                 // the task's acquire (or the wait of an inverse tile for its row of L)
    if (nkt > 0) {
      auto waitfn = [&](int kc) {
        if ((unsigned)kc < haveK) return;                                     // published before the last acquire of this wave
        wait_ge(cA, (unsigned)kc + 1u, P);
        if (cB != cA) wait_ge(cB, (unsigned)kc + 1u, P);
        const unsigned a_ = ld_relaxed(cA), b_ = ld_relaxed(cB);              // (what else has arrived meanwhile is only trusted after ...
        haveK = (unsigned)__builtin_amdgcn_readfirstlane((int)(a_ < b_ ? a_ : b_));
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                    //  ... an acquire that follows the read)
      };
      const double *Ip = Om + (size_t)(kt0 * TC) * (size_t)ldo, *Jp = Ab + (size_t)(kt0 * TC) * (size_t)n;
      if (mk.full() || (mk.nti == 4 && mk.ntj == 4 && !mk.lowtri) || !active)
        tile_gemm<false>(Ip, ldo, r0, n, Jp, n, c0, n, TC * nkt, TC * nkt, TC * nkt, sI, sJ, acc, mk, waitfn);
      else
        tile_gemm<true>(Ip, ldo, r0, n, Jp, n, c0, n, TC * nkt, TC * nkt, TC * nkt, sI, sJ, acc, mk, waitfn);
    }
    TP(1)                                                  // product (incl. its waits)
    // T = C - acc (CHOL) or -acc (INV), kept in the accumulator registers; the 32 loads of two 16-column strips of C are all in
    // flight before the first use (one memory round trip per strip pair, not per entry).  Only a diagonal tile of L goes back
    // to memory here (its factorisation works there); INV on the diagonal is a copy: Y_jj = W_j^T
    if (!chol && diag) {
      for (int e = t; e < TC * TC; e += 256) {
        const int a_ = e & (TC - 1), b_ = e >> 7;          // Y_jj[a_][b_] = W[b_][a_]
        const long row = r0 + a_, col = c0 + b_;
        if (row < n && col < n) Om[(size_t)col * (size_t)ldo + row] = Wb[(size_t)a_ * TC + b_];
      }
    } else if (active && !chol) {
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[tj][ti] = -acc[tj][ti];
    } else if (active) {
#pragma unroll
      for (int tp = 0; tp < 4; tp += 2) {
        double cv[2][4][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const long col = j0 + 16 * (tp + h) + l4 + 4 * v;
            const double *cp = Ab + (size_t)(col < n ? col : n - 1) * (size_t)n;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
              const long row = i0 + 16 * ti + l15;
              cv[h][ti][v] = cp[row < n ? row : n - 1];   // unconditional (clamped) loads: a guarded load is a branch and a round trip EACH
            }
          }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) acc[tp + h][ti][v] = cv[h][ti][v] - acc[tp + h][ti][v];
      }
      if (diag && nkt > 0) {                               // the diagonal tile goes back to memory: its factorisation works there
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const long col = j0 + 16 * tj + l4 + 4 * v;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
              const long row = i0 + 16 * ti + l15;
              if (row < n && col < n) Ab[(size_t)col * (size_t)n + row] = acc[tj][ti][v];
            }
          }
      }
    }
    TP(2)                                                  // T formed
    if (chol && diag) {
      __syncthreads();
      potrf_block_body<4>(Ab, n, c0, pw, Lib + (size_t)(4 * ct) * IB * IB, P.err, reinterpret_cast<double (*)[IB + 1]>(&sI[0][0]));
      TP(3)                                                // diagonal tile
      tile_winv(Ab, n, c0, pw, Lib + (size_t)(4 * ct) * IB * IB, Wb);
      TP(8)                                                // its 128 x 128 inverse
    } else if (!diag) {
      if (!haveD) wait_ge(cL + ct, needD, P);              // the diagonal tile of this column and its inverse
      TP(4)
      tile_solve(acc, Wb, sI, sJ, tiles16(i0, n) > 0 && tiles16(j0, n) > 0);      // X = T L_cc^-T = T W^T, in the registers
      if (tiles16(i0, n) > 0 && tiles16(j0, n) > 0) {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const long col = j0 + 16 * tj + l4 + 4 * v;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
              const long row = i0 + 16 * ti + l15;
              if (row < n && col < n) Om[(size_t)col * (size_t)ldo + row] = acc[tj][ti][v];
            }
          }
      }
      TP(5)                                                // triangular solve
    }
    if (on_chain && t == 0) __hip_atomic_fetch_add(cu_word, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    publish(cOut);
    TLINE(1)
    TP(6)                                                  // publish
    __syncthreads();                                       // LDS and s_task are free for the next task
  }
}

// X column-major (layout XL) from X row-major (XU = what k_tile_chol leaves), lower triangle, optional single-precision copies
template <typename TX>
__global__ __launch_bounds__(256) void k_xu_to_xl(const double *__restrict__ Xsrc, long n, long ldx, long strideX, TX *__restrict__ Xdst,
                                                  int write_xu)
{
  __shared__ double tile[32][33];
  const long rb = blockIdx.x, cb = blockIdx.y;
  if (cb > rb) return;
  const double *XUs = Xsrc + (size_t)blockIdx.z * (size_t)strideX + (size_t)(ldx * n);
  TX *XL = Xdst + (size_t)blockIdx.z * (size_t)strideX, *XU = XL + (size_t)(ldx * n);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const long r = rb * 32 + k, c = cb * 32 + tx;
    const double v = (r < n && c < n && c <= r) ? XUs[(size_t)r * (size_t)ldx + c] : 0.0;
    tile[k][tx] = v;
    if (write_xu && r < n && c < n && c <= r) XU[(size_t)r * (size_t)ldx + c] = (TX)v;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const long c = cb * 32 + k, r = rb * 32 + tx;
    if (r < n && c < n && c <= r) XL[(size_t)c * (size_t)ldx + r] = (TX)tile[tx][k];
  }
}

}  // namespace

#ifdef RBL_TILE_PROF
extern "C" __attribute__((visibility("default"))) int rbl_debug_tile_prof(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_prof), z, sizeof(z)); }
  return 0;
}
extern "C" __attribute__((visibility("default"))) int rbl_debug_tile_line(unsigned long long *out)
{
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_line), sizeof(unsigned long long) * 2 * 32 * 32 * 2) == hipSuccess ? 0 : -1;
}
#endif

// test hook (tests/test_host_logic.py): the queue order of one body with NT tile rows, (NT + 1)^2 slots of {type, i, j}: type 1 CHOL(i, j),
// 2 INV(row tile i, column tile j) of Y, 0 empty -- every task once and after everything it waits for
extern "C" __attribute__((visibility("default"))) int rbl_debug_tile_order(int NT, int *out)
{
  if (NT < 1 || !out) return RBL_ERR_ARG;
  for (int s = 0; s <= NT; ++s)
    for (int u = 0; u <= NT; ++u) {
      bool chol = false; int ti = 0, tj = 0;
      int *o = out + 3 * ((size_t)s * (NT + 1) + u);
      if (!tile_task(NT, s, u, chol, ti, tj)) { o[0] = o[1] = o[2] = 0; continue; }
      o[0] = chol ? 1 : 2; o[1] = ti; o[2] = tj;
    }
  return RBL_OK;
}

// bodies of more than 512 unknowns whose K panels fit one buffer descriptor (2 GB): everything the library calls "large bodies"
bool rbl_tile_cholesky_fits(int64_t n) { return n > 512 && (size_t)n * (size_t)(n + 32) * 8 < ((size_t)1 << 31); }

static size_t tile_counter_bytes(int64_t n, int batch)
{
  const int64_t NT = (n + TC - 1) / TC;
  size_t words = 8 * TQS + 32 + 2 * (size_t)batch * (size_t)NT + TCU;      // queue heads, abort word (own line), counters, the per-CU words
  words = (words + 63) / 64 * 64;                                    // (a multiple of 256 bytes: the W tiles behind it stay aligned)
  return words * sizeof(unsigned);
}

// counters and queue heads (zeroed by every launch) + the 128 x 128 inverses of the diagonal tiles (2 MB a body at n = 1926)
size_t rbl_tile_cholesky_work_bytes(int64_t n, int batch)
{
  const int64_t NT = (n + TC - 1) / TC;
  return tile_counter_bytes(n, batch) + sizeof(double) * (size_t)batch * (size_t)NT * TC * TC;
}

// d_M: `batch` SPD matrices (ld = n, strideA apart), factored in place (lower triangles; the 32 x 32 inverses of the diagonal
// blocks go to d_Linv as rbl_launch_cholesky_batched leaves them).  d_X / d_Xf (either may be NULL): explicit inverses
// [XL | XU] per body with ld = rbl_block_inverse_ld(n), fp64 / fp32; d_Xf alone needs d_X as scratch, so d_X must be given with it.
int rbl_launch_tile_cholesky(hipStream_t st, double *d_M, int64_t n, int batch, int64_t strideA, unsigned *d_err, double *d_Linv,
                             double *d_X, float *d_Xf, void *d_work, int n_cu)
{
  if (!rbl_tile_cholesky_fits(n) || batch <= 0 || !d_work || (d_Xf && !d_X)) return RBL_ERR_ARG;
  const int64_t NT = (n + TC - 1) / TC, nsteps = (n + IB - 1) / IB, ldx = rbl_block_inverse_ld(n);
  unsigned *w = (unsigned *)d_work;
  if (hipMemsetAsync(d_work, 0, tile_counter_bytes(n, batch), st) != hipSuccess) return RBL_ERR_HIP;
  TileChol P;
  P.A = d_M; P.n = (long)n; P.strideA = (long)strideA;
  P.Y = d_X ? d_X + (size_t)(ldx * n) : nullptr; P.ldy = (long)ldx; P.strideY = 2 * (long)(ldx * n);
  P.Linv = d_Linv; P.strideL = (long)(nsteps * IB * IB);
  P.heads = w; P.abort_ = w + 8 * TQS; P.cntL = w + 8 * TQS + 32; P.cntY = P.cntL + (size_t)batch * (size_t)NT;
  P.W = (double *)((char *)d_work + tile_counter_bytes(n, batch));
  P.err = d_err; P.batch = batch; P.NT = (int)NT;
  P.cu_busy = nullptr;
  const size_t tasks = (size_t)batch * (size_t)(NT + 1) * (size_t)(NT + 1);
  size_t grid = 2 * (size_t)(n_cu > 0 ? n_cu : 256);      // two workgroups per CU are resident; later ones find the queues dry
  // A small batch, factor only (a stage's tasks fit one workgroup a CU): ONE workgroup a CU.  A diagonal tile's factorisation is a
  // chain of short fp64 operations; a co-resident workgroup's MFMA stream on the same SIMDs (one pipe for both) doubles its time --
  // body 0's timeline in tools/tile_phase_profile.py: 196 us from the chain tile to the diagonal tile alone, 390 us beside inverse
  // tiles -- and the chain of diagonal tiles is all a small batch waits for (25 x 1926: 3.47 -> 2.96 ms).  With the inverse the
  // launch is work-bound at either grid (4.9 ms).
  if (!d_X && (size_t)batch * (size_t)NT <= grid) grid /= 2;
  // ... with the inverse tiles in the launch (work for every slot: 256 slots cannot hold a stage's 800 tasks) the chain keeps its
  // CUs another way: while a tile of the chain diag(s) -> (s + 1, s) -> diag(s + 1) is at work on a CU, the CU's other workgroup
  // claims nothing (bounded: a millisecond at most).  Factor + inverse, on and off alternating on one box: 25 x 1926 4.94-5.02 ->
  // 4.51-4.60 ms, 50 x 1926 7.23 -> 7.07; from 100 bodies on nothing or a loss (there the chain is not what the launch waits for)
  else if (d_X && (size_t)batch * (size_t)NT <= 2 * grid) P.cu_busy = P.cntY + (size_t)batch * (size_t)NT;
  if (grid > tasks) grid = tasks;
  hipLaunchKernelGGL(k_tile_chol, dim3((unsigned)grid), dim3(256), 0, st, P);
  if (d_X) {
    const dim3 eg((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), (unsigned)batch);
    const long strideX = 2 * (long)(ldx * n);
    for (int b0 = 0; b0 < batch; b0 += 65535) {
      const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
      const dim3 g(eg.x, eg.y, (unsigned)nb);
      if (d_Xf) hipLaunchKernelGGL(k_xu_to_xl<float>, g, dim3(256), 0, st, (const double *)d_X + (size_t)b0 * (size_t)strideX, (long)n, (long)ldx, strideX,
                                   d_Xf + (size_t)b0 * (size_t)strideX, 1);
      hipLaunchKernelGGL(k_xu_to_xl<double>, g, dim3(256), 0, st, (const double *)d_X + (size_t)b0 * (size_t)strideX, (long)n, (long)ldx, strideX,
                         d_X + (size_t)b0 * (size_t)strideX, 0);
    }
  }
  return RBL_OK;
}
