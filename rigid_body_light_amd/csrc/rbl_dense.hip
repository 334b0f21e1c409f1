// rbl_dense.hip -- dense fp64 kernels behind M_half_W (reference c_rigid_obj.cpp:661-675):
// in-place blocked lower Cholesky (what Eigen::LLT computes, :670-671) whose
// trailing updates run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), and the
// triangular product L W (:672).
//
// Storage: column-major n x n, ld = n, 64-bit indexing (cfg 5 has 2.36e10 entries).
//
// Blocking: outer panels of NB columns; inside a panel, IB-wide steps of
//   potf2 (ONE wavefront, rows in registers, v_readlane broadcasts; also emits L_kk^-1)
//   -> trsm as an MFMA product with L_kk^-1 -> rank-IB MFMA update of the rest of the
//   panel; then ONE rank-NB MFMA update of the trailing matrix (LDS-tiled, double-buffered).
// Only the lower triangle is referenced/updated (tiles strictly above the diagonal
// are skipped; diagonal tiles are updated whole).
#include "rbl_dense_dev.hpp"

namespace {

constexpr int NB = 512;   // outer panel width (K of the trailing MFMA update)
constexpr int NBB_INV = 512;   // panel width of the block inversion of large bodies (rbl_launch_block_inverse_large)
constexpr int PBW = 8;   // waves of k_potrf_block

__global__ __launch_bounds__(64 * PBW) void k_potrf_block(double *__restrict__ A, long ld, long k, int pw,
                                                          double *__restrict__ LinvAll, unsigned *err,
                                                          long strideA, long strideL)
{
  A += (size_t)blockIdx.y * (size_t)strideA;        // batched: one matrix per blockIdx.y
  LinvAll += (size_t)blockIdx.y * (size_t)strideL;
  __shared__ double Y[IB][IB + 1];   // L_kk^-1 of the current step, row-major [c][m]
  potrf_block_body<PBW>(A, ld, k, pw, LinvAll, err, Y);
}


// ---- tall panel solve: rows below an outer panel, ALL its IB-blocks in one launch -----------
//   X = A21 L11^{-T}  for rows [r_begin, n) and the panel columns [k0, k0 + nblk*IB)
// Left-looking per 32-column block j:  X_j = (A_j - sum_{m<j} X_m L_jm^T) L_jj^{-T}.
// A wave owns 32 rows and walks the column blocks; everything is MFMA, transposed tiles
// (D'[c][i]) as in the solve of k_potrf_block.  The fp64 C/D layout (row = (lane>>4) + 4v, col = lane&15) IS the
// B-operand layout (k = lane>>4 (+4 ks), n = lane&15), so T = A_j - S goes from the accumulator
// registers straight into the product with Linv_jj -- no LDS, no shuffles.  X_m written by this
// workgroup is re-read (by other lanes) after a workgroup barrier (same CU -> same L1).
// This replaces nblk x (triangular solve + rank-32 update) launches over the tall rows of the
// right-looking panel: the latency-bound chain on the critical path only sees the NB x NB block.
__global__ __launch_bounds__(256) void k_trsm_tall(double *__restrict__ A, long ld, long n, long k0,
                                                   int nblk, long r_begin,
                                                   const double *__restrict__ LinvAll, long strideA, long strideL)
{
  A += (size_t)blockIdx.z * (size_t)strideA;        // batched: one matrix per blockIdx.z
  LinvAll += (size_t)blockIdx.z * (size_t)strideL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long i0 = r_begin + ((long)blockIdx.x * 4 + wave) * 32;
  long irow[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) { long ir = i0 + 16 * q + l15; irow[q] = ir < n ? ir : n - 1; }
  for (int j = 0; j < nblk; ++j) {
    double4_t acc[2][2];  // [tc][ti]  S tile, transposed: rows = panel column c, cols = matrix row i
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const long cj = k0 + (long)j * IB;  // first column (and first L11 row) of block j
    for (int m = 0; m < j; ++m) {
      const double *colp = A + (size_t)(k0 + (long)m * IB + l4) * (size_t)ld;
#pragma unroll
      for (int ks = 0; ks < IB / 4; ++ks) {
        const double *cp = colp + (size_t)(4 * ks) * (size_t)ld;
        const double a0 = cp[cj + l15], a1 = cp[cj + 16 + l15];   // L_jm[c][kk]
        const double b0 = cp[irow[0]], b1 = cp[irow[1]];          // X_m[i][kk]
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    // T = A_j - S in the accumulator layout: element (c = 16 tc + l4 + 4 v, i = 16 ti + l15)
    double4_t T[2][2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          T[tc][ti][v] = A[(size_t)(cj + 16 * tc + l4 + 4 * v) * (size_t)ld + irow[ti]] - acc[tc][ti][v];
    // X_j^T = Linv_jj T : k-step ks uses accumulator register v = ks & 3 of tile tc = ks >> 2 as B operand
    const double *Li = LinvAll + (size_t)j * IB * IB;
    double4_t X[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) X[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < IB / 4; ++ks) {
      const double a0 = Li[l15 * IB + 4 * ks + l4], a1 = Li[(16 + l15) * IB + 4 * ks + l4];
      const double b0 = T[ks >> 2][0][ks & 3], b1 = T[ks >> 2][1][ks & 3];
      X[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, X[0][0], 0, 0, 0);
      X[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, X[0][1], 0, 0, 0);
      X[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, X[1][0], 0, 0, 0);
      X[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, X[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        const long row = i0 + 16 * ti + l15;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (row < n) A[(size_t)(cj + 16 * tc + l4 + 4 * v) * (size_t)ld + row] = X[tc][ti][v];
      }
    __syncthreads();  // X_j visible to every lane of the workgroup before block j+1 re-reads it
  }
}

// ---- rank-K update on the matrix cores ---------------------------------------------------
//   C[i][j] -= sum_{k in [k0,k0+K)} A[i][k] A[j][k]     i in [r0,n), j in [r0,c1), i-tile >= j-tile
// Workgroup = 4 waves (2x2) on a 128x128 tile; wave = 64x64 = 4x4 MFMA tiles of 16x16x4.
// The two 128 x KC panel slabs (i rows, j rows) are staged global -> registers -> LDS, double
// buffered in LDS and in registers (prefetch distance two stages, see the loader); one
// barrier per stage.  LDS column stride 144 doubles (= 32 banks mod 64) keeps the 16-lane x
// 4-column fragment reads conflict-free.  The MFMA computes the TRANSPOSED tile (A-operand
// from the j rows, B-operand from the i rows) so a lane group stores 16 consecutive ROWS of one
// column of C: 128-B runs in column-major C.  K must be a multiple of 2 KC and >= 4 KC (syrk_K_ok): the
// launchers only ever pass whole panels (256 or 512).  A second, generic stage loop next to the peeled one made
// the register allocator spill the prefetch sets (73 VGPRs, each reload behind its own vmcnt(0)).
constexpr int KC = 16;
constexpr int LDP = 144;

// Tile <- workgroup mapping (SyrkGrid): the 1-D grid walks "super-blocks" of 64 tiles (SBI x SBJ), row direction
// fastest, instead of whole tile columns: the ~512 workgroups in flight then touch 64 + 8 panel slabs (36 MB,
// resident in the 256 MB Infinity Cache) instead of 512 + 1.  Measured +2 % at n = 57 780; pinning a super-block
// to one XCD (ids with equal id % 8) was 5-15 % SLOWER than letting its 64 workgroups spread over all XCDs.
struct SyrkGrid {
  int TI, TJ;        // tiles (128) of the updated region: rows, columns (same origin r0)
  int SBI, SBJ;      // super-block shape, SBI * SBJ = 64
  int NSI, NSJ;      // super-blocks per direction
  int tri;           // square update with square super-blocks: only the NSI (NSI + 1) / 2 super-blocks on or below
                     // the diagonal are enumerated (a workgroup of an empty tile still waits for a half-CU slot)
  unsigned nwg;      // = (tri ? NSI (NSI + 1) / 2 : NSI * NSJ) * 64
  long rshift;       // rows of the updated region start at r0 + rshift (0: the symmetric update of the factorisation) ...
  int full;          // ... 1: C[i][j] -= sum_k A[i][k] A[j][k] for EVERY (i, j) tile of the region, no lower-triangle test:
                     //     the row panel and the column panel are different rows of the same columns (block inversion)
};

static bool syrk_K_ok(int64_t K) { return K >= 4 * KC && K % (2 * KC) == 0; }

static SyrkGrid syrk_grid(int64_t rows, int64_t cols)
{
  SyrkGrid g;
  g.TI = (int)((rows + 127) / 128);
  g.TJ = (int)((cols + 127) / 128);
  g.SBJ = 8;
  while (g.SBJ > 1 && g.SBJ / 2 >= g.TJ) g.SBJ /= 2;      // narrow updates (the lookahead columns): 1, 2, 4
  g.SBI = 64 / g.SBJ;
  g.NSI = (g.TI + g.SBI - 1) / g.SBI;
  g.NSJ = (g.TJ + g.SBJ - 1) / g.SBJ;
  g.tri = (rows == cols && g.SBI == g.SBJ) ? 1 : 0;
  g.nwg = (g.tri ? (unsigned)g.NSI * (unsigned)(g.NSI + 1) / 2u : (unsigned)g.NSI * (unsigned)g.NSJ) * 64u;
  g.rshift = 0; g.full = 0;
  return g;
}

// rows [r0 + rshift, ...) x columns [r0, ...): a general C -= A_I A_J^T over a rectangle of tiles
static SyrkGrid syrk_grid_rect(int64_t rows, int64_t cols, int64_t rshift)
{
  SyrkGrid g = syrk_grid(rows, cols);
  if (g.tri) { g.tri = 0; g.nwg = (unsigned)g.NSI * (unsigned)g.NSJ * 64u; }
  g.rshift = (long)rshift; g.full = 1;
  return g;
}

// Diagnostic build only (RBL_EXTRA_FLAGS=-DRBL_SYRK_PROF, tools/syrk_phase_profile.py): s_memtime stamps around the
// phases of a stage, summed over the interior tiles by wave 0 of each workgroup.  No stamp exists in the normal build.
#ifdef RBL_SYRK_PROF
__device__ unsigned long long g_syrk_prof[16];
#define PT(i) { const unsigned long long now_ = clock64(); prof[i] += now_ - tprev; tprev = now_; }
#else
#define PT(i)
#endif
__global__ __launch_bounds__(256, 2) void k_syrk_mfma(double *__restrict__ A, long ld, long r0,
                                                      long c1, long k0, int K, long strideA,
                                                      long n /* rows < n are updated */, SyrkGrid G)
{
  A += (size_t)blockIdx.z * (size_t)strideA;
  __shared__ __attribute__((aligned(16))) double sI[2][KC * LDP];
  __shared__ __attribute__((aligned(16))) double sJ[2][KC * LDP];
  const unsigned w = blockIdx.x;
  const unsigned sb = w >> 6, in = w & 63u;
  int SJ, SI;
  if (G.tri) {   // sb = off(SJ) + SI - SJ, off(c) = c (2 NS - c + 1) / 2: super-column SJ holds the NS - SJ blocks SI >= SJ
    const int NS = G.NSI;
    auto off = [NS](int c) { return (unsigned)(c * (2 * NS - c + 1)) >> 1; };
    const float bq = 2.f * (float)NS + 1.f;
    int c = (int)((bq - __builtin_sqrtf(bq * bq - 8.f * (float)sb)) * 0.5f);
    c = c < 0 ? 0 : (c > NS - 1 ? NS - 1 : c);
    while (c + 1 < NS && off(c + 1) <= sb) ++c;
    while (c > 0 && off(c) > sb) --c;
    SJ = c; SI = c + (int)(sb - off(c));
  } else {
    SJ = (int)(sb / (unsigned)G.NSI); SI = (int)(sb % (unsigned)G.NSI);
  }
  const int bi = SI * G.SBI + (int)(in / (unsigned)G.SBJ), bj = SJ * G.SBJ + (int)(in % (unsigned)G.SBJ);
  if (bi >= G.TI || bj >= G.TJ || (!G.full && bi < bj)) return;  // outside / strictly-upper block tile (block-uniform)
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;   // provably uniform: scalar offsets below
#ifdef RBL_SYRK_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
  const unsigned long long tstart = tprev, rstart = __builtin_amdgcn_s_memrealtime();   // shader cycles, 100 MHz ticks
#endif
  const int wi = wave & 1, wj = wave >> 1;
  const long bi0 = r0 + G.rshift + (long)bi * 128, bj0 = r0 + (long)bj * 128;
  if (bj0 >= c1) return;  // block-uniform
  const long i0 = bi0 + wi * 64, j0 = bj0 + wj * 64;
  const bool active = (i0 < n) && (j0 < c1) && (G.full || !(i0 + 63 < j0));  // wave tile holds lower-triangle entries
  const int l15 = lane & 15, l4 = lane >> 4;

  // loader: thread -> row PAIR 2 (t & 63), column group (t >> 6) * 4 .. +3 of the KC-wide slab: 16-byte loads
  // (one wave instruction = 128 consecutive rows of a column) and ds_write_b128 -- half the memory and LDS
  // instructions of 8-byte pieces; a wave spent 19 % of its time issuing those.
  // Prefetch distance TWO stages through two register sets: the loads of stage s+3 are issued in stage s and
  // land in LDS in stage s+2 -- under load the memory latency exceeds one stage (64 MFMAs = 1.7 us), and a
  // one-stage prefetch left every wave stalled ~40 % of the time in front of its ds_writes.
  const int lrow = (t & 63) * 2, lcg = (t >> 6) * 4;
  long gi = bi0 + lrow; if (gi >= n) gi = n - 2;      // rows >= n feed accumulators that are never stored; a pair
  long gj = bj0 + lrow; if (gj >= n) gj = n - 2;      // (n-1, n) reads one element of the next column: in range
  // the K panel columns through one buffer descriptor: scalar offset = column, one per-lane offset per slab
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      A + (size_t)k0 * (size_t)ld, (short)0, (int)((size_t)K * (size_t)ld * 8), 0x00020000);
  const unsigned ldb = (unsigned)ld * 8u;
  const unsigned vI = (unsigned)lcg * ldb + 8u * (unsigned)gi, vJ = (unsigned)lcg * ldb + 8u * (unsigned)gj;
  rbl_d2 rI[2][4], rJ[2][4];
  const int nst = K / KC;
  auto gload = [&](auto set, int stg) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned so = (unsigned)(stg * KC + q) * ldb;
      rI[set()][q] = buf_ld2(rs, vI, so); rJ[set()][q] = buf_ld2(rs, vJ, so);
    }
  };
  auto lwrite = [&](auto set, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<rbl_d2 *>(&sI[buf][(lcg + q) * LDP + lrow]) = rI[set()][q];
      *reinterpret_cast<rbl_d2 *>(&sJ[buf][(lcg + q) * LDP + lrow]) = rJ[set()][q];
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  gload(S0{}, 0);
  lwrite(S0{}, 0);
  if (nst > 1) gload(S1{}, 1);
  if (nst > 2) gload(S0{}, 2);
  __syncthreads();

  double4_t acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  auto compute = [&](int cur) {
    if (!active) return;
    const double *fi = &sI[cur][l4 * LDP + wi * 64 + l15];
    const double *fj = &sJ[cur][l4 * LDP + wj * 64 + l15];
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { av[q] = fj[ks * 4 * LDP + 16 * q]; bv[q] = fi[ks * 4 * LDP + 16 * q]; }
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tj], bv[ti], acc[tj][ti], 0, 0, 0);
    }
  };
  // epilogue pieces, C -= acc: per pair of 16-column strips, ALL 32 loads of a lane are issued before the first use
  // (a guarded read-modify-write per element costs one memory round trip EACH and was 2/3 of a tile's time).
  // Interior tiles (block-uniform test) skip the guards.
  const bool interior = (i0 + 64 <= n) && (j0 + 64 <= c1);
  // interior tiles address C through a buffer descriptor over the 64 columns of this wave: scalar offset =
  // column, four per-lane offsets (one per 16-row group) instead of 32 address pairs
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
      A + (size_t)j0 * (size_t)ld, (short)0, (int)((size_t)64 * (size_t)ld * 8), 0x00020000);
  unsigned vC[4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) vC[ti] = (unsigned)l4 * ldb + 8u * (unsigned)(i0 + 16 * ti + l15);
  auto load_C_fast = [&](int tp, double (&cv)[2][4][4]) {      // interior tiles only
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const unsigned so = (unsigned)(16 * (tp + h) + 4 * v) * ldb;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) cv[h][ti][v] = buf_ld(rc, vC[ti], so);
      }
  };
  auto load_C = [&](int tp, double (&cv)[2][4][4]) {
    if (interior) { load_C_fast(tp, cv); return; }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        long row = i0 + 16 * ti + l15;
        if (row >= n) row = n - 1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          long col = j0 + 16 * (tp + h) + l4 + 4 * v;
          if (col >= c1) col = c1 - 1;
          cv[h][ti][v] = A[(size_t)col * (size_t)ld + row];
        }
      }
  };
  auto store_C = [&](int tp, const double (&cv)[2][4][4]) {
    if (interior) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const unsigned so = (unsigned)(16 * (tp + h) + 4 * v) * ldb;
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) buf_st(rc, vC[ti], so, cv[h][ti][v] - acc[tp + h][ti][v]);
        }
      return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        const long row = i0 + 16 * ti + l15;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const long col = j0 + 16 * (tp + h) + l4 + 4 * v;
          if (row < n && col < c1) A[(size_t)col * (size_t)ld + row] = cv[h][ti][v] - acc[tp + h][ti][v];
        }
      }
  };
  // stage s (data in LDS buffer s & 1): stage s+1 sits in register set (s+1) & 1 -> LDS buffer (s+1) & 1 (free
  // since the barrier that ended stage s-1), then that set is refilled with stage s+3.  No panel loads are
  // issued in the last three stages, so the first half of the C tile is requested two stages before the
  // end (its registers are the then idle prefetch sets) and has landed when the epilogue starts.
  auto stage = [&](int s_, auto nset) {
    if (s_ + 1 < nst) lwrite(nset, (s_ + 1) & 1);
    PT(0)
    if (s_ + 3 < nst) gload(nset, s_ + 3);
    PT(1)
    compute(s_ & 1);
    PT(2)
    __syncthreads();
    PT(3)
  };
  double cv0[2][4][4];
  PT(4)
  {                                        // nst is even and >= 4 (K = 256, 512; checked by the launchers): last two stages peeled
    for (int s2 = 0; s2 < nst - 2; s2 += 2) {
      stage(s2, S1{});
      stage(s2 + 1, S0{});
    }
    lwrite(S1{}, (nst - 1) & 1);          // stage nst-2: the last slab goes to LDS, both register sets are idle now
    if (active && interior) load_C_fast(0, cv0);   // (edge tiles load late: their clamped addresses need the registers)
    compute((nst - 2) & 1);
    __syncthreads();
    compute((nst - 1) & 1);               // stage nst-1
    PT(5)
    if (!active) return;
    if (!interior) load_C(0, cv0);
  }
  store_C(0, cv0);
  double cv1[2][4][4];
  load_C(2, cv1);
  store_C(2, cv1);
#ifdef RBL_SYRK_PROF
  __builtin_amdgcn_s_waitcnt(0);
  PT(6)
  if (lane == 0 && wave == 0 && interior) {
    for (int q = 0; q < 7; ++q) atomicAdd(&g_syrk_prof[q], prof[q]);
    atomicAdd(&g_syrk_prof[7], tprev - tstart);
    atomicAdd(&g_syrk_prof[8], 1ull);
    atomicAdd(&g_syrk_prof[9], __builtin_amdgcn_s_memrealtime() - rstart);
  }
#endif
}

__global__ void k_zero_upper(double *__restrict__ A, long n)
{
  const long i = (long)blockIdx.y * blockDim.x + threadIdx.x;  // row
  const long j = blockIdx.x;                                   // column (x: up to 2^31-1)
  if (i < n && i < j) A[(size_t)j * n + i] = 0.0;
}

// ---- out = L W (lower-triangular matvec), HBM-read bound -------------------------
constexpr int TR_ROWS = 256;
constexpr int TR_COLS = 1024;

__global__ __launch_bounds__(TR_ROWS) void k_trmv_partial(const double *__restrict__ L, long n,
                                                          const double *__restrict__ W,
                                                          double *__restrict__ part)
{
  const long row = (long)blockIdx.x * TR_ROWS + threadIdx.x;
  const long cb = (long)blockIdx.y * TR_COLS;
  const long row_hi = (long)blockIdx.x * TR_ROWS + TR_ROWS - 1;
  double acc = 0.0;
  if (cb <= row_hi && row < n) {
    long ce = cb + TR_COLS; if (ce > n) ce = n;
    if (ce > row + 1) ce = row + 1;
    const double *p = L + (size_t)cb * (size_t)n + row;
    long c = cb;
    for (; c + 4 <= ce; c += 4) {
      const double l0 = p[0], l1 = p[(size_t)n], l2 = p[2 * (size_t)n], l3 = p[3 * (size_t)n];
      acc = __builtin_fma(l0, W[c], acc);
      acc = __builtin_fma(l1, W[c + 1], acc);
      acc = __builtin_fma(l2, W[c + 2], acc);
      acc = __builtin_fma(l3, W[c + 3], acc);
      p += 4 * (size_t)n;
    }
    for (; c < ce; ++c) { acc = __builtin_fma(p[0], W[c], acc); p += (size_t)n; }
  }
  if (row < n) part[(size_t)blockIdx.y * (size_t)n + row] = acc;
}

__global__ void k_trmv_reduce(const double *__restrict__ part, long n, int nchunks,
                              double *__restrict__ out)
{
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double s = 0.0;
  const int used = (int)(row / TR_COLS) + 1;  // chunks beyond the diagonal hold zeros
  for (int c = 0; c < used && c < nchunks; ++c) s += part[(size_t)c * (size_t)n + row];
  out[row] = s;
}

}  // namespace

#ifdef RBL_SYRK_PROF
extern "C" __attribute__((visibility("default"))) int rbl_debug_syrk_prof(unsigned long long *out, int reset)
{
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_syrk_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_syrk_prof), z, sizeof(z)); }
  return 0;
}
#endif

size_t rbl_cholesky_work_bytes(int64_t) { return sizeof(double) * (NB / IB) * IB * IB; }   // one L_kk^-1 per IB-step of a panel

// Panel p is factored on the high-priority auxiliary stream while the big trailing update R_{p-1}
// of the previous panel still runs on the caller's stream (one-panel lookahead):
//   aux   : [wait L_{p-1}] factor panel p, record P_p
//   caller: wait P_p ; L_p = update of the NEXT panel's columns, record L_p ; R_p = the rest
// R_{p-1} touches only columns >= k_p + NB, the panel only its own NB columns, so the two
// never write the same entries; same-column updates stay ordered on the caller's stream.
int rbl_launch_cholesky(hipStream_t st, double *d_M, int64_t n, bool zero_upper, unsigned *d_err,
                        double *d_work, size_t work_bytes, const RblCholAux *aux)
{
  if (!d_work || work_bytes < sizeof(double) * (NB / IB) * IB * IB) return RBL_ERR_ARG;
  double *Linv = d_work;
  const bool look = aux && aux->stream && n > 4 * NB;
  // sp: panel stream (the HIGH-priority auxiliary stream, so the small latency-bound panel kernels
  // are dispatched ahead of the remaining workgroups of the big update);  su: update stream (caller's)
  hipStream_t sp = look ? aux->stream : st, su = st;
  if (look) {
    if (hipEventRecord(aux->ev[2], st) != hipSuccess) return RBL_ERR_HIP;
    if (hipStreamWaitEvent(sp, aux->ev[2], 0) != hipSuccess) return RBL_ERR_HIP;
  }
  bool pending_L = false;
  // panel width: NB while the trailing matrix is large (its rank-NB update then hides the panel
  // chain), NB/2 near the end where the panel chain itself is the critical path
  auto width_at = [n](int64_t k0) -> int64_t {
    const int64_t w = (n - k0 > 6144) ? NB : NB / 2;          // measured: 6144 best at n = 24 300 and 57 780
    return (n - k0 < w) ? (n - k0) : w;
  };
  for (int64_t k = 0; k < n;) {
    const int64_t pw = width_at(k);  // panel width
    const int64_t pend = k + pw;
    if (look && pending_L) {
      if (hipStreamWaitEvent(sp, aux->ev[1], 0) != hipSuccess) return RBL_ERR_HIP;
      pending_L = false;
    }
    // the NB x NB diagonal block: one 16-wave workgroup, barriers instead of launches
    hipLaunchKernelGGL(k_potrf_block, dim3(1), dim3(64 * PBW), 0, sp, d_M, (long)n, (long)k, (int)pw, Linv, d_err,
                       0L, 0L);
    if (pend < n)            // rows below the block: X = A21 L11^-T, all NB/IB column blocks in one launch
      hipLaunchKernelGGL(k_trsm_tall, dim3((unsigned)((n - pend + 127) / 128)), dim3(256), 0, sp, d_M, (long)n,
                         (long)n, (long)k, (int)(pw / IB), (long)pend, Linv, 0L, 0L);
    if (pend < n) {  // trailing update with the whole panel, K = pw = NB (a short panel is the last one)
      if (!syrk_K_ok(pw)) return RBL_ERR_ARG;
      if (look) {
        if (hipEventRecord(aux->ev[0], sp) != hipSuccess) return RBL_ERR_HIP;
        if (hipStreamWaitEvent(su, aux->ev[0], 0) != hipSuccess) return RBL_ERR_HIP;
      }
      const int64_t lend = pend + width_at(pend);               // L_p: the next panel's columns
      {
        const SyrkGrid G = syrk_grid(n - pend, lend - pend);
        hipLaunchKernelGGL(k_syrk_mfma, dim3(G.nwg), dim3(256), 0, su, d_M, (long)n, (long)pend, (long)lend,
                           (long)k, (int)pw, 0L, (long)n, G);
      }
      if (look) {
        if (hipEventRecord(aux->ev[1], su) != hipSuccess) return RBL_ERR_HIP;
        pending_L = true;
      }
      if (lend < n) {                                          // R_p: everything right of it
        const SyrkGrid G = syrk_grid(n - lend, n - lend);
        hipLaunchKernelGGL(k_syrk_mfma, dim3(G.nwg), dim3(256), 0, su, d_M, (long)n, (long)lend, (long)n,
                           (long)k, (int)pw, 0L, (long)n, G);
      }
    }
    k = pend;
  }
  if (look) {   // the last panel ran on sp
    if (hipEventRecord(aux->ev[2], sp) != hipSuccess) return RBL_ERR_HIP;
    if (hipStreamWaitEvent(st, aux->ev[2], 0) != hipSuccess) return RBL_ERR_HIP;
  }
  if (zero_upper)
    hipLaunchKernelGGL(k_zero_upper, dim3((unsigned)n, (unsigned)((n + 255) / 256)), dim3(256), 0,
                       st, d_M, (long)n);
  return RBL_OK;
}


// ---------------------------------------------------------------------------------------------
// Batched variant for the block-diagonal preconditioner (reference Block_diag_invM, :461-487):
// `batch` SPD matrices of order n (one per rigid body, n = 3 N_blb), stride strideA doubles.
// Same kernels with one matrix per blockIdx.y/z, outer panels of 256 columns, no lookahead.  All diagonal-block
// inverses are KEPT: Linv[b][step][IB*IB] feed the substitution kernel below.
// ---------------------------------------------------------------------------------------------
size_t rbl_cholesky_batched_work_bytes(int64_t n, int batch)
{
  const int64_t nsteps = (n + IB - 1) / IB;
  return sizeof(double) * (size_t)batch * (size_t)nsteps * IB * IB;
}

int rbl_launch_cholesky_batched(hipStream_t st, double *d_M, int64_t n, int batch, int64_t strideA,
                                unsigned *d_err, double *d_Linv)
{
  constexpr int GRID_YZ_MAX = 65535;                 // bodies ride in gridDim.y / .z: more than that go in several rounds
  if (batch > GRID_YZ_MAX) {
    const size_t lstride = rbl_cholesky_batched_work_bytes(n, 1) / sizeof(double);
    for (int b0 = 0; b0 < batch; b0 += GRID_YZ_MAX) {
      const int nb = batch - b0 < GRID_YZ_MAX ? batch - b0 : GRID_YZ_MAX;
      const int rc = rbl_launch_cholesky_batched(st, d_M + (size_t)b0 * (size_t)strideA, n, nb, strideA, d_err,
                                                 d_Linv + (size_t)b0 * lstride);
      if (rc) return rc;
    }
    return RBL_OK;
  }
  constexpr int NBB = 512;   // outer panel of the batched factorisation: rank-512 trailing updates (n = 486: 256 same, 128 / 64 slower)
  const int64_t nsteps = (n + IB - 1) / IB;
  const long strideL = (long)(nsteps * IB * IB);
  for (int64_t k = 0; k < n; k += NBB) {
    const int64_t pw = (n - k < NBB) ? (n - k) : NBB;
    const int64_t pend = k + pw;
    double *Lk = d_Linv + (size_t)(k / IB) * IB * IB;   // this panel's L_kk^-1 blocks (all are kept)
    // diagonal block of every matrix: one workgroup each; then the rows below it, all column steps fused
    hipLaunchKernelGGL(k_potrf_block, dim3(1, batch), dim3(64 * PBW), 0, st, d_M, (long)n, (long)k, (int)pw, Lk, d_err,
                       (long)strideA, strideL);
    if (pend < n)
      hipLaunchKernelGGL(k_trsm_tall, dim3((unsigned)((n - pend + 127) / 128), 1, batch), dim3(256), 0, st, d_M, (long)n,
                         (long)n, (long)k, (int)(pw / IB), (long)pend, (const double *)Lk, (long)strideA, strideL);
    if (pend < n) {        // trailing matrix, K = NBB (a short panel is the last one)
      if (!syrk_K_ok(pw)) return RBL_ERR_ARG;
      const SyrkGrid G = syrk_grid(n - pend, n - pend);
      hipLaunchKernelGGL(k_syrk_mfma, dim3(G.nwg, 1, batch), dim3(256), 0, st, d_M, (long)n, (long)pend, (long)n, (long)k,
                         (int)pw, (long)strideA, (long)n, G);
    }
  }
  return RBL_OK;
}

namespace {
// x = (L L^T)^-1 v for every matrix of the batch: one workgroup per matrix, the vector lives in
// LDS, L is streamed once per sweep.  Diagonal IB-blocks are applied through their stored
// inverses (no sequential 32-step chains).
//   forward :  t = Linv_kk y_k ; y_k = t ; y[rest] -= L[rest, k-block] t
//   backward:  x_k = Linv_kk^T y_k ;        y[before] -= L[k-block, before]^T x_k
constexpr int SOLVE_MAXN = 8192;   // 64 KB of LDS

constexpr int BS_T = 1024;   // threads per matrix: the sweeps are latency-bound, more rows in flight per step

// NV right-hand sides per matrix (vector v of body b at in + v * rhs_pitch + b * vec_stride): L is streamed ONCE for
// all of them -- the sweeps are latency chains, so NV vectors cost about what one does.
// GLOBAL (bodies whose vector does not fit the 64 KB of LDS a workgroup gets: more than 2 730 blobs): the working vector is
// the OUTPUT vector itself in HBM / L2 -- one workgroup owns it, its waves share the CU's vector L1, and __syncthreads() orders
// their global accesses -- only the IB-entry scratch stays in LDS.  Same operation order per row, no size limit.
template <int NV, bool GLOBAL = false>
__global__ __launch_bounds__(BS_T) void k_block_solve(const double *__restrict__ L, long n, long strideA,
                                                     const double *__restrict__ Linv, long strideL,
                                                     const double *in, double *out, long vec_stride, long rhs_pitch,
                                                     int mode /* 0: L L^T, 1: L only, 2: L^T only */)
{
  extern __shared__ double y_lds[];                  // NV x n doubles + NV x IB scratch (GLOBAL: the scratch only)
  const int b = blockIdx.x, t = threadIdx.x;
  double *const y = GLOBAL ? out + (size_t)b * (size_t)vec_stride : y_lds;     // vector v at y + v * ypitch
  const size_t ypitch = GLOBAL ? (size_t)rhs_pitch : (size_t)n;
  double *tbuf = GLOBAL ? y_lds : y_lds + (size_t)NV * n;                      // tbuf[v * IB + m]
  const double *Lb = L + (size_t)b * (size_t)strideA;
  const double *Lib = Linv + (size_t)b * (size_t)strideL;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double *vin = in + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
    if (!GLOBAL || vin != y + (size_t)v * ypitch)
      for (long e = t; e < n; e += BS_T) y[(size_t)v * ypitch + e] = vin[e];
  }
  __syncthreads();
  const int nsteps = (int)((n + IB - 1) / IB);
  const int tv = t / IB, tt = t % IB;                // threads 0 .. NV*IB-1: (vector, row of the diagonal block)
  for (int s = 0; s < nsteps && mode != 2; ++s) {    // ---- forward: L y' = v
    const long k = (long)s * IB;
    const int nb = (int)((n - k < IB) ? n - k : IB);
    const double *Li = Lib + (size_t)s * IB * IB;
    if (t < IB * NV) {
      // all 32 entries of the row of Linv in flight together (a loop with a data-dependent trip count waits for every load in turn:
      // that chain, not the bytes, was most of a step -- the backward sweep's strided form took 17 us a step, 1.04 ms a sweep)
      double li[IB];
#pragma unroll
      for (int m = 0; m < IB; ++m) li[m] = Li[tt * IB + m];
      double acc = 0.0;
      const double *yv = y + (size_t)tv * ypitch + k;
#pragma unroll
      for (int m = 0; m < IB; ++m)
        if (m <= tt && tt < nb) acc = __builtin_fma(li[m], yv[m], acc);
      tbuf[t] = acc;
    }
    __syncthreads();
    if (t < IB * NV && tt < nb) y[(size_t)tv * ypitch + k + tt] = tbuf[t];
    if (nb == IB) {   // full block: 32 independent strided loads per row are issued back to back
      for (long r = k + IB + t; r < n; r += BS_T) {
        const double *col = Lb + (size_t)k * (size_t)n + r;
        double lv[IB];
#pragma unroll
        for (int m = 0; m < IB; ++m) lv[m] = col[(size_t)m * n];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double a0 = y[(size_t)v * ypitch + r], a1 = 0.0;
#pragma unroll
          for (int m = 0; m < IB; m += 2) {
            a0 = __builtin_fma(-lv[m], tbuf[v * IB + m], a0);
            a1 = __builtin_fma(-lv[m + 1], tbuf[v * IB + m + 1], a1);
          }
          y[(size_t)v * ypitch + r] = a0 + a1;
        }
      }
    } else {
      for (long r = k + nb + t; r < n; r += BS_T) {
        const double *col = Lb + (size_t)k * (size_t)n + r;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double acc = y[(size_t)v * ypitch + r];
          for (int m = 0; m < nb; ++m) acc = __builtin_fma(-col[(size_t)m * n], tbuf[v * IB + m], acc);
          y[(size_t)v * ypitch + r] = acc;
        }
      }
    }
    __syncthreads();
  }
  for (int s = nsteps - 1; s >= 0 && mode != 1; --s) {   // ---- backward: L^T x = y'
    const long k = (long)s * IB;
    const int nb = (int)((n - k < IB) ? n - k : IB);
    const double *Li = Lib + (size_t)s * IB * IB;
    if (t < IB * NV) {
      double li[IB];
#pragma unroll
      for (int m = 0; m < IB; ++m) li[m] = Li[m * IB + tt];                                // column tt of Linv, all loads in flight
      double acc = 0.0;
      const double *yv = y + (size_t)tv * ypitch + k;
#pragma unroll
      for (int m = 0; m < IB; ++m)
        if (m >= tt && m < nb && tt < nb) acc = __builtin_fma(li[m], yv[m], acc);           // Linv^T
      tbuf[t] = acc;
    }
    __syncthreads();
    if (t < IB * NV && tt < nb) y[(size_t)tv * ypitch + k + tt] = tbuf[t];
    if (nb == IB) {   // columns before the block: y[c] -= sum_r L[k+r][c] x_r
      // The 32 entries of a column are one 256-byte run.  Sixteen lanes share a column (16 bytes = two rows each), so a
      // wave's load instruction covers four whole runs -- eight cache lines for 1 KB, as coalesced as the forward sweep --
      // and the 32-term sum closes with four DPP steps inside the 16-lane row.  (One column per lane, the round-1 form,
      // touched 64 lines per instruction: the backward sweep took twice the forward one, 1.23 vs 0.61 ms at cfg 3.)
      const int lane = t & 63, wv = t >> 6, sub = lane >> 4, h = lane & 15;
      double x0[NV], x1[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) { x0[v] = tbuf[v * IB + 2 * h]; x1[v] = tbuf[v * IB + 2 * h + 1]; }
      constexpr int UB = 8, CS = 4 * (BS_T / 64);                  // UB loads in flight per lane (16 spill: 128 VGPRs at 1024 threads)
      for (long cb = 4 * wv; cb < k; cb += (long)UB * CS) {
        double l0[UB], l1[UB];
#pragma unroll
        for (int q = 0; q < UB; ++q) {
          const long c = cb + (long)q * CS + sub;                  // k is a multiple of 32: c0 + 3 < k whenever c0 < k
          if (cb + (long)q * CS < k) {
            const double *p = Lb + (size_t)c * (size_t)n + k + 2 * h;
            l0[q] = p[0]; l1[q] = p[1];
          }
        }
#pragma unroll
        for (int q = 0; q < UB; ++q) {
          const long c = cb + (long)q * CS + sub;
          if (cb + (long)q * CS < k) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              double a = __builtin_fma(l0[q], x0[v], l1[q] * x1[v]);
              a += dpp_row<0xB1>(a);                               // quad_perm [1,0,3,2]
              a += dpp_row<0x4E>(a);                               // quad_perm [2,3,0,1]
              a += dpp_row<0x141>(a);                              // row_half_mirror
              a += dpp_row<0x140>(a);                              // row_mirror: every lane of the row holds the sum
              if (h == 0) y[(size_t)v * ypitch + c] -= a;
            }
          }
        }
      }
    } else {
      for (long c = t; c < k; c += BS_T) {
        const double *row = Lb + (size_t)c * (size_t)n + k;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double acc = y[(size_t)v * ypitch + c];
          for (int m = 0; m < nb; ++m) acc = __builtin_fma(-row[m], tbuf[v * IB + m], acc);
          y[(size_t)v * ypitch + c] = acc;
        }
      }
    }
    __syncthreads();
  }
  if (GLOBAL) return;                                // the working vector WAS the output
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double *o = out + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
    for (long e = t; e < n; e += BS_T) o[e] = y[(size_t)v * ypitch + e];
  }
}

// ---- the same substitution as ONE software pipeline (round 5; bodies of more than 170 blobs whose vectors fit LDS) ----------
// k_block_solve spends a step as: inverse of the diagonal block from HBM (a round trip), barrier, every thread's loads of the
// step issued together, a wait for the slowest, the sums, barrier -- 11 us a step at n = 1926 where the step's bytes take 5-8.
// Here the dependent chain lives in ONE wave and the other fifteen only stream:
//   * the diag wave owns, in step s, the 32 x 32 block BELOW (forward) / LEFT OF (backward) the diagonal block -- the only
//     entries of the step whose results the NEXT step's diagonal solve needs -- and then solves that next diagonal block
//     with its stored inverse: x_{s+1} is in LDS when the step's barrier falls, so a step is ONE barrier and no wave waits
//     for a diagonal solve.  What it reads arrives by LDS-direct loads (buffer_load ... lds: no registers) in a ring of three
//     slots, asked for TWO steps ahead and retired by a counted s_waitcnt vmcnt(16): the chain never sees a memory round trip
//     (in-kernel stamps, tools/pipe_step_profile.py: 0.06 us waiting, 0.6 us of sums, 0.2 us issuing a step).
//   * the streaming waves hold two register sets and refill a set as soon as it has been summed -- with data of the NEXT
//     step when the step is over: addresses never depend on the solution -- so every load is asked for two units ahead and
//     the memory pipe is never drained, not even across the barrier.  Forward: a unit = 1920 rows x 8 columns, two rows
//     (16 bytes) a lane, 1 KB runs an instruction; backward: 480 columns x 32 rows, a wave on 32 consecutive columns, sixteen
//     lanes a column with two rows each (four 256-byte runs an instruction), the 32-term sums closed inside the 16-lane row
//     by DPP; its last steps (one unit each) alternate the two sets step by step, i.e. ask two STEPS ahead.  A wave whose rows
//     / columns lie beyond the step's sits the step out: a CU's address path takes 16 clocks an instruction whatever it
//     fetches, and a step's worth of instructions that fetch nothing used to cost what a full step costs (4.7 us).
// Measured at cfg 3 (200 x 1926, tools/bench_block_pipe.py): L^-1 v 0.61 -> 0.51 ms = 5.9 TB/s, L^-T v 0.75 -> 0.60 ms (its
// 256-byte runs start anywhere in a 128-byte line: the L2 fetches 1.33 x the forward sweep's bytes, profiles/r05_block_pipe*),
// (L L^T)^-1 v 1.33 -> 1.09 ms; a rank's 25 bodies 1.03 -> 0.68 ms.
// Per row the terms are added in a fixed order (columns ascending forward, blocks descending backward): results are bitwise
// reproducible run to run; they differ from k_block_solve's in the last bits (other association).
// Needs n >= 3 IB; Linv blocks of a ragged last step are padded with the identity (potf2_wave), y with zeros.
constexpr int BP_SW = BS_T / 64 - 1;   // streaming waves
constexpr int BP_ST = 64 * BP_SW;      // streaming threads (a forward unit: two rows each)
constexpr int BP_BQ = 8;               // 16-byte loads per lane of a backward unit
constexpr int BP_BC = 4 * BP_SW * BP_BQ;   // columns of a backward unit
constexpr unsigned BP_OOB = 0xF0000000u;   // an offset beyond any factor's descriptor (8190^2 x 8 B = 0.5 GB): such a load returns 0
constexpr int BP_SLOT = 2 * IB * IB;       // a ring slot: the block beside a diagonal block | the inverse of the next diagonal block
constexpr int BP_RING = 3;                 // slots: the step at work, one landed or landing, one just asked for
constexpr int BP_K = 16;                   // LDS-direct loads a step: 2 x 8 KB at 16 bytes a lane.  Two steps' worth stay below the 63 a wave
                                           // can have in flight (with 4-byte loads, 40 a step, the ISSUE of a step's loads waited for the last one's)
static_assert(BP_K == 16, "the counted waits in k_block_solve_pipe say vmcnt(16)");
typedef __attribute__((address_space(3))) void rbl_lds_void;
constexpr size_t BP_LDS_MAX = 160 * 1024;  // a workgroup may hold the CU's whole LDS (one workgroup of 1024 threads a CU anyway)

#if defined(RBL_PIPE_PROF)   // diagnostic build only (tools/pipe_step_profile.py): shader-clock stamp of streaming thread 0 after every barrier
__device__ unsigned long long g_pipe_prof[256 * 512];
#define PIPE_STAMP() do { if (t == 0 && b < 128 && pstamp < 512) g_pipe_prof[b * 512 + pstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#define PIPE_DSTAMP() do { if (lane == 0 && b < 128 && pstamp < 512) g_pipe_prof[(128 + b) * 512 + pstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PIPE_STAMP() do { } while (0)
#define PIPE_DSTAMP() do { } while (0)
#endif

template <int NV>
__global__ __launch_bounds__(BS_T) void k_block_solve_pipe(const double *__restrict__ L, long n, long strideA,
                                                          const double *__restrict__ Linv, long strideL, const double *in,
                                                          double *out, long vec_stride, long rhs_pitch,
                                                          int mode /* 0: L L^T, 1: L only, 2: L^T only */)
{
  extern __shared__ double y_lds[];                  // NV x npad doubles (the vectors, zero padded) + 2 x NV x IB (x of two steps)
  const int b = blockIdx.x, t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int nsteps = (int)((n + IB - 1) / IB);
  const long npad = (long)nsteps * IB;
#if defined(RBL_PIPE_PROF)
  int pstamp = 0;
#endif
  double *const y = y_lds;                           // y[v * npad + e]
  double *const xb = y_lds + (size_t)NV * npad;      // xb[((s & 1) * NV + v) * IB + m]
  double *const ring = xb + 2 * NV * IB;             // the diag wave's prefetch ring: BP_RING slots of BP_SLOT doubles, filled by LDS-direct loads
  const double *Lb = L + (size_t)b * (size_t)strideA;
  const double *Lib = Linv + (size_t)b * (size_t)strideL;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Lb), (short)0, (int)((size_t)n * (size_t)n * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsLi =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Lib), (short)0, (int)((size_t)nsteps * IB * IB * 8), 0x00020000);
  const unsigned ldb = (unsigned)n * 8u;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double *vin = in + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
    for (long e = t; e < npad; e += BS_T) y[(size_t)v * npad + e] = e < n ? vin[e] : 0.0;
  }
  __syncthreads();
  // The two roles run their own loops (a wave-uniform branch; every wave meets the same number of barriers): each loop's
  // registers are its own -- one loop with a branch per step made the compiler carry both roles' register sets through every merge
  const bool diag = wave == BP_SW;
  const int dr = lane >> 1, dh = lane & 1;           // diag wave: row (forward) / column (backward) of the block, and which 16 of its 32 terms
  const int sub = lane >> 4, h = lane & 15;          // streaming lane of the backward sweep: column of its wave's four, rows 2 h, 2 h + 1

  // ---- forward: L y' = v ------------------------------------------------------------------------------------------------------
  if (mode != 2 && diag) {
    __builtin_amdgcn_s_setprio(3);                   // the chain every other wave waits for
    // D(s) = the 32 x 32 block below diagonal block s + the inverse of diagonal block s + 1, into ring slot s mod 3
    auto issue = [&](int s_) {
      const int s = __builtin_amdgcn_readfirstlane(s_);   // (wave-uniform by construction; said so)
      double *slot = ring + (size_t)(s % BP_RING) * BP_SLOT;
      // four columns of the block per load, as they lie: sixteen lanes a column, two rows (16 bytes) a lane.  Rows beyond the matrix
      // (the ragged last block) read whatever follows -- the next column's top, or zeros past the descriptor's end -- and are never used
      const unsigned vo = (unsigned)(lane >> 4) * ldb + (unsigned)(32 * s + 32 + 2 * (lane & 15)) * 8u;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rbl_lds_void *)(slot + 128 * q), 16, (int)vo, (int)((unsigned)(32 * s + 4 * q) * ldb), 0, 0);
      // the inverse, 16 bytes a lane, pair-transposed: LDS chunk c = 64 q + lane <- entries 2 m2, 2 m2 + 1 of row c mod 32, m2 = c / 32
      // (row dr's sixteen pairs are then 512 bytes apart and the 32 rows of a pair consecutive: conflict-free 16-byte reads)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = 64 * q + lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsLi, (rbl_lds_void *)(slot + IB * IB + 128 * q), 16,
                                                 (int)(((unsigned)(c & 31) * IB + 2u * (unsigned)(c >> 5)) * 8u), (int)((unsigned)(s + 1) * IB * IB * 8u), 0, 0);
      }
    };
    auto solve_next = [&](int sn, const double *lis) {   // x_sn = Linv_sn y_sn (y_sn final in LDS; lis: Linv_sn's pair-transposed image)
      double *xn = xb + (size_t)((sn & 1) * NV) * IB;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const double *yv = y + (size_t)v * npad + 32L * sn + 16 * dh;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const rbl_d2 w2 = *reinterpret_cast<const rbl_d2 *>(lis + ((size_t)(8 * dh + jj) * IB + dr) * 2);
          a[(2 * jj) & 3] = __builtin_fma(16 * dh + 2 * jj <= dr ? w2.x : 0.0, yv[2 * jj], a[(2 * jj) & 3]);
          a[(2 * jj + 1) & 3] = __builtin_fma(16 * dh + 2 * jj + 1 <= dr ? w2.y : 0.0, yv[2 * jj + 1], a[(2 * jj + 1) & 3]);
        }
        double r = (a[0] + a[1]) + (a[2] + a[3]);
        r += dpp_row<0xB1>(r);                       // the other half of the row's sum sits in the neighbouring lane
        if (dh == 0) { xn[v * IB + dr] = r; y[(size_t)v * npad + 32L * sn + dr] = r; }
      }
    };
    {                                                // x_0 = Linv_0 y_0: the inverse of block 0 through the last slot's image
      double *slot = ring + (size_t)(BP_RING - 1) * BP_SLOT;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = 64 * q + lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsLi, (rbl_lds_void *)(slot + IB * IB + 128 * q), 16,
                                                 (int)(((unsigned)(c & 31) * IB + 2u * (unsigned)(c >> 5)) * 8u), 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      solve_next(0, slot + IB * IB);
    }
    issue(0);
    issue(nsteps > 2 ? 1 : 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (a raw barrier: __syncthreads() would drain the ring's loads)
    for (int s = 0; s + 1 < nsteps; ++s) {
      const double *x = xb + (size_t)((s & 1) * NV) * IB;
      const double *slot = ring + (size_t)(s % BP_RING) * BP_SLOT;
      const long r0 = 32L * (s + 1);
      PIPE_DSTAMP();
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // D(s) has landed; D(s + 1) may still be on its way (BP_K loads a step)
      PIPE_DSTAMP();
#pragma unroll
      for (int v = 0; v < NV; ++v) {                 // y_head -= L_head x_s
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j & 3] = __builtin_fma(slot[(16 * dh + j) * IB + dr], x[v * IB + 16 * dh + j], a[j & 3]);
        double r = (a[0] + a[1]) + (a[2] + a[3]);
        r += dpp_row<0xB1>(r);
        if (dh == 0 && r0 + dr < n) y[(size_t)v * npad + r0 + dr] -= r;      // (the padding of a ragged last block stays zero)
      }
      __builtin_amdgcn_wave_barrier();               // one wave: its LDS accesses execute in program order
      solve_next(s + 1, slot + IB * IB);
      PIPE_DSTAMP();
      issue(s + 2 < nsteps - 1 ? s + 2 : nsteps - 2);   // two steps ahead, into the slot step s - 1 used (past the end: the last block again, never read)
      PIPE_DSTAMP();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
  } else if (mode != 2) {
    // A unit = 1920 rows (two a lane: 16-byte loads, 1 KB an instruction -- with 8-byte loads the CU's address path, 16 clocks an
    // instruction whatever it fetches, capped a body at 50 GB/s) x 8 columns.  A step is four units a row group: sets A, B, A, B.
    // A wave whose 128 rows lie beyond the matrix sits the step out (it has nothing to ask for in any later step either).
    constexpr int FR = 2 * BP_ST;
    const long wrow = 64 + 128L * wave;              // the wave's first row of a row group, counted from the step's diagonal block
    auto niw = [&](int s) { const long R = n - 32L * s - wrow; return R > 0 ? (int)((R + FR - 1) / FR) : 0; };
    double A[16], B[16];
    // every load of an active wave is issued by every lane: a lane without rows (or a fetch past the sweep's end) points beyond the
    // descriptor's range and gets zeros without a memory access -- no branch around the loads, a register set is simply overwritten
    auto fload = [&](double (&buf)[16], int s_, int i_, int k, bool any_) {
      const int s = __builtin_amdgcn_readfirstlane(s_), i = __builtin_amdgcn_readfirstlane(i_);   // (wave-uniform by construction; said so)
      const bool any = __builtin_amdgcn_readfirstlane((int)any_) != 0;
      const long r = 32L * s + 64 + 2L * t + (long)FR * i;
      const unsigned vo = any && r < n ? (unsigned)r * 8u : BP_OOB;
      const unsigned so = any ? (unsigned)(32 * s + 8 * k) * ldb : 0u;           // (a column that exists even when the fetch is a dummy)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const rbl_d2 w2 = buf_ld2(rs, vo, so + (unsigned)j * ldb);
        buf[2 * j] = w2.x; buf[2 * j + 1] = w2.y;
      }
    };
    fload(A, 0, 0, 0, niw(0) > 0);
    fload(B, 0, 0, 1, niw(0) > 0);
    __syncthreads();
    PIPE_STAMP();
    for (int s = 0; s + 1 < nsteps; ++s) {
      const double *x = xb + (size_t)((s & 1) * NV) * IB;
      const int ni = niw(s), nin = niw(s + 1);
      for (int i = 0; i < ni; ++i) {
        const long r = 32L * s + 64 + 2L * t + (long)FR * i;
        const long rl = r + 1 < npad ? r : npad - 2;  // (a lane without rows sums zeros into a copy of the last rows and stores nothing)
        const bool more = i + 1 < ni;                // the row group after this one: the step's next 1920 rows, or the next step's first
        const int sn = more ? s : s + 1, in = more ? i + 1 : 0;
        const bool any = more || nin > 0;
        double a0[NV], a1[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const rbl_d2 y2 = *reinterpret_cast<const rbl_d2 *>(y + (size_t)v * npad + rl);
          a0[v] = y2.x; a1[v] = y2.y;
        }
        // A set is refilled as soon as it has been summed, with the unit after next: every load is asked for two units ahead.
        // (the scheduler, left alone, hoists a refill above the sums that still read the set -- into a third set, spilled: the
        // fences keep "sum a set, refill it" as written; the empty asm pins the sums where they are written)
        auto quarter = [&](const double (&buf)[16], int k) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            double p0 = a0[v], p1 = a1[v];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const double xj = x[v * IB + 8 * k + j];
              p0 = __builtin_fma(-buf[2 * j], xj, p0);
              p1 = __builtin_fma(-buf[2 * j + 1], xj, p1);
            }
            asm volatile("" : "+v"(p0), "+v"(p1));
            a0[v] = p0; a1[v] = p1;
          }
        };
        quarter(A, 0);
        __builtin_amdgcn_sched_barrier(0);
        fload(A, s, i, 2, true);
        __builtin_amdgcn_sched_barrier(0);
        quarter(B, 1);
        __builtin_amdgcn_sched_barrier(0);
        fload(B, s, i, 3, true);
        __builtin_amdgcn_sched_barrier(0);
        quarter(A, 2);
        __builtin_amdgcn_sched_barrier(0);
        fload(A, sn, in, 0, any);
        __builtin_amdgcn_sched_barrier(0);
        quarter(B, 3);
        __builtin_amdgcn_sched_barrier(0);
        fload(B, sn, in, 1, any);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (r < n) y[(size_t)v * npad + r] = a0[v];
          if (r + 1 < n) y[(size_t)v * npad + r + 1] = a1[v];
        }
      }
      __syncthreads();
      PIPE_STAMP();
    }
  }

  // ---- backward: L^T x = y' ----------------------------------------------------------------------------------------------------
  int s0 = nsteps - 1;
  if (mode != 1) {                                   // the top block, and -- when it is a ragged one -- its few rows by the plain form
    auto top = [&](int sn) {                         // x_sn = Linv_sn^T y_sn by the diag wave, with loads of its own
      if (diag) {
        double *xn = xb + (size_t)((sn & 1) * NV) * IB;
        const double *p = Lib + (size_t)sn * IB * IB + (size_t)(16 * dh) * IB + dr;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const double *yv = y + (size_t)v * npad + 32L * sn + 16 * dh;
          double a = 0.0;
#pragma unroll
          for (int j = 0; j < 16; ++j) a = __builtin_fma(16 * dh + j >= dr ? p[j * IB] : 0.0, yv[j], a);
          a += dpp_row<0xB1>(a);
          if (dh == 0) { xn[v * IB + dr] = a; y[(size_t)v * npad + 32L * sn + dr] = a; }
        }
      }
      __syncthreads();
    };
    top(s0);
    const int nb = (int)(n - 32L * s0);
    if (nb < IB) {
      const double *x = xb + (size_t)((s0 & 1) * NV) * IB;
      const long k = 32L * s0;
      for (long c = t; c < k; c += BS_T) {
        const double *row = Lb + (size_t)c * (size_t)n + k;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double acc = y[(size_t)v * npad + c];
          for (int m = 0; m < nb; ++m) acc = __builtin_fma(-row[m], x[v * IB + m], acc);
          y[(size_t)v * npad + c] = acc;
        }
      }
      __syncthreads();
      top(--s0);
    }
  }
  if (mode != 1 && diag) {
    __builtin_amdgcn_s_setprio(3);
    // D(s) = the 32 x 32 block left of diagonal block s, pair-transposed (LDS chunk c = 64 q + lane <- rows 2 rp, 2 rp + 1 of column
    // c mod 32, rp = c / 32: lane dr then reads column dr in 16-byte pieces 512 bytes apart, the 32 columns of a piece consecutive)
    // + the inverse of diagonal block s - 1 as it lies, into ring slot s mod 3.  Blocks s0 .. 1 are full ones
    auto issue = [&](int s_) {
      const int s = __builtin_amdgcn_readfirstlane(s_);   // (wave-uniform by construction; said so)
      double *slot = ring + (size_t)(s % BP_RING) * BP_SLOT;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = 64 * q + lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rbl_lds_void *)(slot + 128 * q), 16, (int)((unsigned)(c & 31) * ldb + (unsigned)(c >> 5) * 16u),
                                                 (int)((unsigned)(32 * (s - 1)) * ldb + (unsigned)(32 * s) * 8u), 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsLi, (rbl_lds_void *)(slot + IB * IB + 128 * q), 16, lane * 16,
                                                 (int)((unsigned)(s - 1) * IB * IB * 8u + 1024u * q), 0, 0);
    };
    issue(s0);
    issue(s0 > 1 ? s0 - 1 : 1);
    for (int s = s0; s > 0; --s) {
      const double *x = xb + (size_t)((s & 1) * NV) * IB;
      double *xn = xb + (size_t)(((s - 1) & 1) * NV) * IB;
      const double *slot = ring + (size_t)(s % BP_RING) * BP_SLOT;
      const double *lis = slot + IB * IB;
      const long r0 = 32L * (s - 1);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
      for (int v = 0; v < NV; ++v) {                 // y_head -= L_head^T x_s
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const rbl_d2 w2 = *reinterpret_cast<const rbl_d2 *>(slot + ((size_t)(8 * dh + jj) * IB + dr) * 2);
          a[(2 * jj) & 3] = __builtin_fma(w2.x, x[v * IB + 16 * dh + 2 * jj], a[(2 * jj) & 3]);
          a[(2 * jj + 1) & 3] = __builtin_fma(w2.y, x[v * IB + 16 * dh + 2 * jj + 1], a[(2 * jj + 1) & 3]);
        }
        double r = (a[0] + a[1]) + (a[2] + a[3]);
        r += dpp_row<0xB1>(r);
        if (dh == 0) y[(size_t)v * npad + r0 + dr] -= r;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int v = 0; v < NV; ++v) {                 // x_{s-1} = Linv_{s-1}^T y_head
        const double *yv = y + (size_t)v * npad + r0 + 16 * dh;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j & 3] = __builtin_fma(16 * dh + j >= dr ? lis[(16 * dh + j) * IB + dr] : 0.0, yv[j], a[j & 3]);
        double r = (a[0] + a[1]) + (a[2] + a[3]);
        r += dpp_row<0xB1>(r);
        if (dh == 0) { xn[v * IB + dr] = r; y[(size_t)v * npad + r0 + dr] = r; }
      }
      issue(s - 2 > 0 ? s - 2 : 1);                  // (past the end: block 1 again, never read)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
  } else if (mode != 1) {
    // A unit = 480 columns x the step's 32 rows: a wave takes 32 consecutive columns of it, sixteen lanes a column with two rows
    // (16 bytes) each, four columns (four 256-byte runs) an instruction.  A wave whose columns lie beyond the step's sits it out.
    const long wcol = 32L * wave;
    auto nuw = [&](int s) { const long K = 32L * (s - 1) - wcol; return K > 0 ? (int)((K + BP_BC - 1) / BP_BC) : 0; };
    double A[2 * BP_BQ], B[2 * BP_BQ];
    auto bload = [&](double (&buf)[2 * BP_BQ], int s_, int u_, bool any_) {
      // (wave-uniform by construction; said so, or one instantiation addresses every load through a loop over "distinct" offsets)
      const int s = __builtin_amdgcn_readfirstlane(s_), u = __builtin_amdgcn_readfirstlane(u_);
      const bool any = __builtin_amdgcn_readfirstlane((int)any_) != 0;
      // ("no such unit": the vector offset carries the out-of-range marker and the scalar one is zero -- whichever of the two the
      // descriptor's range check looks at, it refuses the access)
      const unsigned vo = any ? (unsigned)(wcol + sub) * ldb + (unsigned)(2 * h) * 8u : BP_OOB;
      const unsigned so = any ? (unsigned)(BP_BC * u) * ldb + (unsigned)(32 * s) * 8u : 0u;
#pragma unroll
      for (int q = 0; q < BP_BQ; ++q) {
        const rbl_d2 w2 = buf_ld2(rs, vo, so + (unsigned)(4 * q) * ldb);
        buf[2 * q] = w2.x; buf[2 * q + 1] = w2.y;
      }
    };
    auto bsum = [&](const double (&buf)[2 * BP_BQ], int s_, int u_) {
      const int s = __builtin_amdgcn_readfirstlane(s_), u = __builtin_amdgcn_readfirstlane(u_);
      const double *x = xb + (size_t)((s & 1) * NV) * IB;
      double x0[NV], x1[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) { x0[v] = x[v * IB + 2 * h]; x1[v] = x[v * IB + 2 * h + 1]; }
      const long c0 = (long)BP_BC * u + wcol + sub;
#pragma unroll
      for (int q = 0; q < BP_BQ; ++q) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double a = __builtin_fma(buf[2 * q], x0[v], buf[2 * q + 1] * x1[v]);
          a = rbl_row_sum16(a);
          if (h == 0) y[(size_t)v * npad + c0 + 4 * q] -= a;
        }
      }
    };
    // Set A takes a step's even units, set B the odd ones; a set is refilled as soon as it has been summed -- with the step's unit
    // after next, or the next step's first (A) / second (B) -- so every load is asked for a whole pair of units ahead.
    bload(A, s0, 0, nuw(s0) > 0);
    bload(B, s0, 1, nuw(s0) > 1);
    PIPE_STAMP();
    int s = s0;
    for (; s > 0 && nuw(s) > 1; --s) {
      const int nu = nuw(s), nun = nuw(s - 1);
      for (int u = 0; u < nu; u += 2) {
        bsum(A, s, u);
        __builtin_amdgcn_sched_barrier(0);
        { const bool more = u + 2 < nu; bload(A, more ? s : s - 1, more ? u + 2 : 0, more || nun > 0); }
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < nu) {
          bsum(B, s, u + 1);
          __builtin_amdgcn_sched_barrier(0);
          { const bool more = u + 3 < nu; bload(B, more ? s : s - 1, more ? u + 3 : 1, more || nun > 1); }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      PIPE_STAMP();
    }
    // The wave's last steps have one unit each and little to stream: what a step costs is the round trip of its loads.  The two
    // sets take turns, step by step, so each step's columns are asked for TWO steps ahead
    if (s > 0) {
      bload(B, s - 1, 0, nuw(s - 1) > 0);
      while (s > 0) {
        if (nuw(s) > 0) bsum(A, s, 0);
        __builtin_amdgcn_sched_barrier(0);
        bload(A, s - 2, 0, nuw(s - 2) > 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        PIPE_STAMP();
        if (--s == 0) break;
        if (nuw(s) > 0) bsum(B, s, 0);
        __builtin_amdgcn_sched_barrier(0);
        bload(B, s - 2, 0, nuw(s - 2) > 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        PIPE_STAMP();
        --s;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double *o = out + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
    for (long e = t; e < n; e += BS_T) o[e] = y[(size_t)v * npad + e];
  }
}

// The same sweeps for bodies of order n <= BSS_T (shell_N_12 / 42 / 162: n = 36 / 126 / 486), where a sweep is a chain of
// n / IB dependent steps and nothing else: thread t owns row t (forward) / column t (backward) for the whole solve and
// the IB factor entries it needs for step s + 1 are already on their way to registers while step s runs (double
// buffered), as is the next diagonal inverse -- so a step no longer waits for a round trip to L2 / HBM.  Measured at
// n = 486: 147 -> 123 us for both sweeps only (the chain of 2 x 16 steps with two barriers each remains), which is why
// bodies of 65..170 blobs use explicit inverses instead (below) and this kernel serves the short chains (n <= 192) and
// the shared body-frame factor of free-space systems (rotations fused: Q, rot).  Same operation order per row as k_block_solve.
constexpr int BSS_T = 512;

__device__ __forceinline__ void quat_rot_d(const double *q, double *R)      // R(Q), scalar-first unit quaternion
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


template <int NV>
__global__ __launch_bounds__(BSS_T) void k_block_solve_small(const double *__restrict__ L, long n, long strideA,
                                                             const double *__restrict__ Linv, long strideL,
                                                             const double *in, double *out, long vec_stride, long rhs_pitch,
                                                             int mode /* 0: L L^T, 1: L only, 2: L^T only */,
                                                             const double *__restrict__ Q, int rot)
{
  // Q, rot: body-frame factor shared by all bodies (strideA = strideL = 0): rot & 1 rotates the input into the body frame
  // (v_k <- R_b^T v_k per blob), rot & 2 the result back (x_k <- R_b x_k) -- see bf_build in rbl_bodies.hip
  extern __shared__ double y[];                      // NV x n doubles + NV x IB scratch
  double *tbuf = y + (size_t)NV * n;                 // tbuf[v * IB + m]
  const int b = blockIdx.x, t = threadIdx.x;
  const double *Lb = L + (size_t)b * (size_t)strideA;
  const double *Lib = Linv + (size_t)b * (size_t)strideL;
  double R[9];
  if (rot) quat_rot_d(Q + 4 * (size_t)b, R);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double *vin = in + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
    if (3 * t < n) {                                 // one blob (three entries) per thread
      const double a0 = vin[3 * t], a1 = vin[3 * t + 1], a2 = vin[3 * t + 2];
      double *d = y + (size_t)v * n + 3 * t;
      if (rot & 1) {
        d[0] = R[0] * a0 + R[3] * a1 + R[6] * a2;
        d[1] = R[1] * a0 + R[4] * a1 + R[7] * a2;
        d[2] = R[2] * a0 + R[5] * a1 + R[8] * a2;
      } else { d[0] = a0; d[1] = a1; d[2] = a2; }
    }
  }
  const int nsteps = (int)((n + IB - 1) / IB);
  const bool diag = t < IB * NV;                     // threads 0 .. NV*IB-1: (vector, row of the diagonal block)
  const int tv = diag ? t / IB : 0, tt = t % IB;
  double li[IB], la[IB], lb[IB];

  if (mode != 2) {                                   // ---- forward: L y' = v
    auto load_panel = [&](int s, double (&dst)[IB]) {            // L[t][k + m], rows below block s (a full block, or none)
      const long k = (long)s * IB;
      if (t >= k + IB && t < n) {
        const double *col = Lb + (size_t)k * (size_t)n + t;
#pragma unroll
        for (int m = 0; m < IB; ++m) dst[m] = col[(size_t)m * n];
      }
    };
    auto load_diag = [&](int s) {                                // row tt of Linv_kk
      if (diag) {
        const double *Li = Lib + (size_t)s * IB * IB + tt * IB;
#pragma unroll
        for (int m = 0; m < IB; ++m) li[m] = Li[m];
      }
    };
    auto step = [&](int s, double (&cur)[IB], double (&nxt)[IB]) {
      const long k = (long)s * IB;
      const int nb = (int)((n - k < IB) ? n - k : IB);
      if (s + 1 < nsteps) load_panel(s + 1, nxt);
      if (diag) {
        const double *yv = y + (size_t)tv * n + k;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < IB; ++m)
          if (m <= tt && tt < nb) acc = __builtin_fma(li[m], yv[m], acc);
        tbuf[t] = acc;
      }
      if (s + 1 < nsteps) load_diag(s + 1);
      __syncthreads();
      if (diag && tt < nb) y[(size_t)tv * n + k + tt] = tbuf[t];
      if (t >= k + IB && t < n) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double a0 = y[(size_t)v * n + t], a1 = 0.0;
#pragma unroll
          for (int m = 0; m < IB; m += 2) {
            a0 = __builtin_fma(-cur[m], tbuf[v * IB + m], a0);
            a1 = __builtin_fma(-cur[m + 1], tbuf[v * IB + m + 1], a1);
          }
          y[(size_t)v * n + t] = a0 + a1;
        }
      }
      __syncthreads();
    };
    load_diag(0);
    load_panel(0, la);
    __syncthreads();
    for (int s = 0; s < nsteps; s += 2) {
      step(s, la, lb);
      if (s + 1 < nsteps) step(s + 1, lb, la);
    }
  } else {
    __syncthreads();
  }

  if (mode != 1) {                                   // ---- backward: L^T x = y'
    auto load_panel = [&](int s, double (&dst)[IB]) {            // L[k + m][t], columns before block s
      const long k = (long)s * IB;
      const int nb = (int)((n - k < IB) ? n - k : IB);
      if (t < k) {
        const double *row = Lb + (size_t)t * (size_t)n + k;
#pragma unroll
        for (int m = 0; m < IB; ++m) dst[m] = m < nb ? row[m] : 0.0;
      }
    };
    auto load_diag = [&](int s) {                                // column tt of Linv_kk
      if (diag) {
        const double *Li = Lib + (size_t)s * IB * IB + tt;
#pragma unroll
        for (int m = 0; m < IB; ++m) li[m] = Li[m * IB];
      }
    };
    auto step = [&](int s, double (&cur)[IB], double (&nxt)[IB]) {
      const long k = (long)s * IB;
      const int nb = (int)((n - k < IB) ? n - k : IB);
      if (s > 0) load_panel(s - 1, nxt);
      if (diag) {
        const double *yv = y + (size_t)tv * n + k;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < IB; ++m)
          if (m >= tt && m < nb) acc = __builtin_fma(li[m], yv[m], acc);
        tbuf[t] = acc;                                           // 0 for rows beyond a ragged last block
      }
      if (s > 0) load_diag(s - 1);
      __syncthreads();
      if (diag && tt < nb) y[(size_t)tv * n + k + tt] = tbuf[t];
      if (t < k) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          double a0 = y[(size_t)v * n + t], a1 = 0.0;
#pragma unroll
          for (int m = 0; m < IB; m += 2) {
            a0 = __builtin_fma(-cur[m], tbuf[v * IB + m], a0);
            a1 = __builtin_fma(-cur[m + 1], tbuf[v * IB + m + 1], a1);
          }
          y[(size_t)v * n + t] = a0 + a1;
        }
      }
      __syncthreads();
    };
    load_diag(nsteps - 1);
    load_panel(nsteps - 1, la);
    for (int s = nsteps - 1; s >= 0; s -= 2) {
      step(s, la, lb);
      if (s - 1 >= 0) step(s - 1, lb, la);
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double *o = out + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
    if (3 * t < n) {
      const double *x = y + (size_t)v * n + 3 * t;
      if (rot & 2) {
        o[3 * t] = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
        o[3 * t + 1] = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
        o[3 * t + 2] = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
      } else { o[3 * t] = x[0]; o[3 * t + 1] = x[1]; o[3 * t + 2] = x[2]; }
    }
  }
}

// ---- explicit per-body inverses X = L^-1 for small bodies (n <= BSS_T) ---------------------------------------------
// A substitution sweep is a chain of n / IB dependent steps whatever is prefetched (n = 486: ~60 us a sweep, 50 bodies or
// 5000); with X at hand a sweep is one triangular matrix-vector product, every output independent.  X costs n^3 / 3
// flops per body and factorisation, the same as the factorisation itself.
//   k_trtri_small : workgroup (block column j, body): forward substitution of the IB unit vectors of block j through L,
//                   a wave keeps 64 rows of X[:, block j] in MFMA accumulators (rank-IB updates on the matrix cores; a
//                   VALU form with the T operand broadcast from LDS was LDS-bandwidth-bound, 709 us at 50 x 486), the
//                   IB x IB diagonal products go through LDS; stores X twice: XL column-major (X[r][c] at c n + r) and XU row-major (r n + c), so that
//                   X v and X^T v both read consecutive addresses across a wavefront.
//   k_block_inv_apply : workgroup = 63 outputs (21 blobs) x 4 waves, each wave a quarter of the sum (interleaved), fixed-order LDS
//                   reduction; out = X v (upper = 0) or X^T v (upper = 1).
__global__ __launch_bounds__(BSS_T) void k_trtri_small(const double *__restrict__ L, long n, long strideA,
                                                       const double *__restrict__ Linv, long strideL, double *__restrict__ X)
{
  __shared__ double sY[IB][IB + 1], sLi[IB][IB + 1], sT[IB][IB + 1];
  const int j = blockIdx.x, b = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
  const int wave = t >> 6, lane = t & 63, l15 = lane & 15, l4 = lane >> 4;
  const long k0 = (long)j * IB;
  const int nbj = (int)((n - k0 < IB) ? n - k0 : IB);
  const int nsteps = (int)((n + IB - 1) / IB);
  const double *Lb = L + (size_t)b * (size_t)strideA;
  const double *Lib = Linv + (size_t)b * (size_t)strideL;
  double *XL = X + (size_t)b * 2 * (size_t)(n * n), *XU = XL + (size_t)(n * n);
  // a wave owns 64 rows of X[:, block j] as 4 x 2 transposed MFMA tiles: acc[ti][tc][v] = Y[i][c],
  // c = 16 tc + l4 + 4 v, i = rbase + 16 ti + l15  (the C/D layout of v_mfma_f64_16x16x4_f64, as in k_trsm_tall)
  const long rbase = 64L * wave;
  double4_t acc[4][2];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) acc[ti][tc] = (double4_t){0.0, 0.0, 0.0, 0.0};
  constexpr int LP = IB * IB / 128;                  // Linv entries per thread at the smallest workgroup (128 threads)
  double bq[4][IB / 4], lip[LP];

  auto load_panel = [&](int s) {                     // B operands: L[i][k + 4 ks + l4] for the tiles below block s
    const long k = (long)s * IB;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const long i = rbase + 16 * ti + l15;
      const bool on = rbase + 16 * ti >= k + IB && i < n;
      const double *p = Lb + (size_t)(k + l4) * (size_t)n + (on ? i : 0);
#pragma unroll
      for (int ks = 0; ks < IB / 4; ++ks) bq[ti][ks] = on ? p[(size_t)(4 * ks) * (size_t)n] : 0.0;
    }
  };
  auto load_diag = [&](int s) {
    const double *Li = Lib + (size_t)s * IB * IB;
#pragma unroll
    for (int i = 0; i < LP; ++i) { const int e = t + i * nt; if (e < IB * IB) lip[i] = Li[e]; }
  };
  auto stage_rows = [&](double4_t (&y0)[2], double4_t (&y1)[2]) {       // the two tiles of block s -> sY[m][c]
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        sY[l15][16 * tc + l4 + 4 * v] = y0[tc][v];
        sY[16 + l15][16 * tc + l4 + 4 * v] = y1[tc][v];
      }
  };
  auto take_rows = [&](double4_t (&y0)[2], double4_t (&y1)[2]) {        // rows of block s are final: X = T
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        y0[tc][v] = sT[l15][16 * tc + l4 + 4 * v];
        y1[tc][v] = sT[16 + l15][16 * tc + l4 + 4 * v];
      }
  };
  load_diag(j);
  load_panel(j);
  for (int s = j; s < nsteps; ++s) {
    const long k = (long)s * IB;
    const int wo = (int)(k >> 6), hi = (int)((k >> 5) & 1);              // owner wave of block s, which pair of its tiles
    if (s == j) {                                                        // Y = E_j: the block is the identity
      for (int e = t; e < IB * IB; e += nt) sY[e / IB][e % IB] = (e / IB == e % IB) ? 1.0 : 0.0;
    } else if (wave == wo) {
      if (hi) stage_rows(acc[2], acc[3]); else stage_rows(acc[0], acc[1]);
    }
#pragma unroll
    for (int i = 0; i < LP; ++i) { const int e = t + i * nt; if (e < IB * IB) sLi[e / IB][e % IB] = lip[i]; }
    __syncthreads();
    if (s + 1 < nsteps) load_diag(s + 1);
    for (int e = t; e < IB * IB; e += nt) {                              // T = Linv_ss Y_s
      const int a = e / IB, c = e % IB;
      double sum = 0.0;
#pragma unroll
      for (int m = 0; m < IB; ++m)
        if (m <= a) sum = __builtin_fma(sLi[a][m], sY[m][c], sum);
      sT[a][c] = sum;
    }
    __syncthreads();
    if (wave == wo) {
      if (hi) take_rows(acc[2], acc[3]); else take_rows(acc[0], acc[1]);
    }
    if (rbase + 63 >= k + IB && s + 1 < nsteps) {                         // Y[i][:] -= L[i][k-block] T on the matrix cores
      double ta[2][IB / 4];
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int ks = 0; ks < IB / 4; ++ks) ta[tc][ks] = -sT[4 * ks + l4][16 * tc + l15];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
        if (rbase + 16 * ti >= k + IB) {
#pragma unroll
          for (int ks = 0; ks < IB / 4; ++ks) {
            acc[ti][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[0][ks], bq[ti][ks], acc[ti][0], 0, 0, 0);
            acc[ti][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[1][ks], bq[ti][ks], acc[ti][1], 0, 0, 0);
          }
        }
    }
    if (s + 1 < nsteps) load_panel(s + 1);                               // in flight during the next diagonal product
  }
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    const long i = rbase + 16 * ti + l15;
    if (i >= k0 && i < n) {
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = 16 * tc + l4 + 4 * v;
          if (c < nbj) {
            XL[(size_t)(k0 + c) * (size_t)n + i] = acc[ti][tc][v];
            XU[(size_t)i * (size_t)n + k0 + c] = acc[ti][tc][v];
          }
        }
    }
  }
}

constexpr int BIA_W = 4;      // waves per workgroup of k_block_inv_apply: BIA_R outputs, each wave a quarter of every sum (measured at 50 x 486: 2 waves 27 us, 4 waves 15.8, 8 waves 18.2)
constexpr int BIA_RQ = 63;    // outputs per workgroup of the rotated forms: 21 whole blobs (the output rotation needs whole blobs)

// mstride: doubles between the matrices of consecutive bodies (2 n^2; 0 = ONE body-frame matrix shared by all bodies, see
// the body-frame factors in rbl_bodies.hip).  rot & 1: the input is rotated into the body frame first (v_k <- R_b^T v_k per
// blob), rot & 2: the output is rotated back (x_k <- R_b x_k); Q: quaternions of the bodies (4 per body, relative to b = 0).
// TM: storage type of the matrix (double; float = the single-precision copy of the explicit inverses of large bodies:
// half the bytes, converted on load, sums in fp64).  The row blocks with the longest sums are dealt first (revx).
// ROWS: outputs per workgroup -- 63 (whole blobs: the rotated forms) or 64 (with ldx a multiple of 16 every wave load is then
// four whole 128-byte lines; 63-row segments of an unpadded matrix straddle five: 4.5 instead of 5.3 TB/s at 200 x 1926).
// ldx: doubles between consecutive columns of a layout (n for the small-body inverses, padded for the large ones).
template <int NV, typename TM, int ROWS>
__global__ __launch_bounds__(64 * BIA_W) void k_block_inv_apply(const TM *__restrict__ X, long n, long ldx, long mstride,
                                                                const double *in, double *out, long vec_stride, long rhs_pitch,
                                                                int upper, const double *__restrict__ Q, int rot)
{
  constexpr int BIA_R = ROWS;
  extern __shared__ double v[];                      // NV x n vector, then BIA_W x 64 x NV partial sums
  double *red = v + (size_t)NV * n;
  const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int bx = upper ? (int)blockIdx.x : (int)(gridDim.x - 1 - blockIdx.x);      // X v: last rows have the longest sums; X^T v: the first
  const TM *A = X + (size_t)b * (size_t)mstride + (upper ? (size_t)(ldx * n) : 0);
  double R[9];
  if (rot) quat_rot_d(Q + 4 * (size_t)b, R);
#pragma unroll
  for (int vv = 0; vv < NV; ++vv) {
    const double *vin = in + (size_t)vv * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
    for (long k = t; 3 * k < n; k += 64 * BIA_W) {   // one blob (three entries) per thread
      const double a0 = vin[3 * k], a1 = vin[3 * k + 1], a2 = vin[3 * k + 2];
      double *d = v + (size_t)vv * n + 3 * k;
      if (rot & 1) {                                 // R^T a
        d[0] = R[0] * a0 + R[3] * a1 + R[6] * a2;
        d[1] = R[1] * a0 + R[4] * a1 + R[7] * a2;
        d[2] = R[2] * a0 + R[5] * a1 + R[8] * a2;
      } else { d[0] = a0; d[1] = a1; d[2] = a2; }
    }
  }
  __syncthreads();
  const long e0 = (long)bx * BIA_R, e = e0 + lane, ec = e < n ? e : n - 1;
  const long eend = (e0 + BIA_R < n) ? e0 + BIA_R : n;
  // X v: columns q <= e (the wave's range ends with its last row);  X^T v: rows q >= e
  const long qlo = upper ? e0 : 0, qhi = upper ? n : eend;
  double acc[NV];
#pragma unroll
  for (int vv = 0; vv < NV; ++vv) acc[vv] = 0.0;
  constexpr int U = 32;                              // loads in flight per lane: the whole sum is latency, not bandwidth
  for (long q0 = qlo + w; q0 < qhi; q0 += BIA_W * U) {
    TM a[U];                                         // unconditional loads (clamped column): a guarded load is a branch and a
#pragma unroll                                       // memory round trip EACH -- the float instantiation ran 7x slower that way
    for (int u = 0; u < U; ++u) {
      const long q = q0 + BIA_W * u;
      if constexpr (std::is_same<TM, double>::value) a[u] = q < qhi ? A[(size_t)q * (size_t)ldx + ec] : 0.0;   // (compiles to 32 loads back to back)
      else a[u] = A[(size_t)(q < qhi ? q : qhi - 1) * (size_t)ldx + ec];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long q = q0 + BIA_W * u;
      const bool ok = q < qhi && (upper ? q >= e : q <= e);
      const double av = ok ? (double)a[u] : 0.0;
      const long qc = q < qhi ? q : qhi - 1;
#pragma unroll
      for (int vv = 0; vv < NV; ++vv) acc[vv] = __builtin_fma(av, v[(size_t)vv * n + qc], acc[vv]);
    }
  }
#pragma unroll
  for (int vv = 0; vv < NV; ++vv) red[(w * 64 + lane) * NV + vv] = acc[vv];
  __syncthreads();
  double r[NV];
  if (w == 0) {
#pragma unroll
    for (int vv = 0; vv < NV; ++vv) {
      double sum = red[lane * NV + vv];
#pragma unroll
      for (int ww = 1; ww < BIA_W; ++ww) sum += red[(ww * 64 + lane) * NV + vv];        // fixed order
      r[vv] = sum;
    }
  }
  if (rot & 2) {                                     // whole blobs of this workgroup: x_k <- R x_k
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int vv = 0; vv < NV; ++vv) red[vv * 64 + lane] = r[vv];
    }
    __syncthreads();
    if (w == 0) {
      const int k3 = 3 * (lane / 3), d = lane % 3;
#pragma unroll
      for (int vv = 0; vv < NV; ++vv) {
        const double *x = red + vv * 64 + k3;
        r[vv] = R[3 * d] * x[0] + R[3 * d + 1] * x[1] + R[3 * d + 2] * x[2];
      }
    }
  }
  if (w == 0 && lane < BIA_R && e < n) {
#pragma unroll
    for (int vv = 0; vv < NV; ++vv) out[(size_t)vv * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride + e] = r[vv];
  }
}

// v_k <- R_b v_k (transpose = 0) or R_b^T v_k (1) for every blob k of bodies 0 .. (pointers relative to the first body)
__global__ void k_rotate_bodies(const double *__restrict__ Q, const double *in, double *out, int N_blb, long nblobs,
                                long rhs_pitch, int transpose)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nblobs) return;
  double R[9];
  quat_rot_d(Q + 4 * (k / N_blb), R);
  const size_t o = (size_t)blockIdx.y * (size_t)rhs_pitch + 3 * (size_t)k;
  const double a0 = in[o], a1 = in[o + 1], a2 = in[o + 2];
  if (transpose) {
    out[o] = R[0] * a0 + R[3] * a1 + R[6] * a2; out[o + 1] = R[1] * a0 + R[4] * a1 + R[7] * a2;
    out[o + 2] = R[2] * a0 + R[5] * a1 + R[8] * a2;
  } else {
    out[o] = R[0] * a0 + R[1] * a1 + R[2] * a2; out[o + 1] = R[3] * a0 + R[4] * a1 + R[5] * a2;
    out[o + 2] = R[6] * a0 + R[7] * a1 + R[8] * a2;
  }
}
}  // namespace

// measured (tools/bench_block_solve.py, both sweeps): n = 486 x 50 bodies 124 -> 43 us; n = 126 x 400 and n = 36 x 2000 are
// chains of 4 and 2 steps only, where the substitution kernel (23 us) beats two matrix-vector launches
bool rbl_block_inverse_fits(int64_t n) { return n > 192 && n <= BSS_T; }
// explicit inverses of LARGE bodies are applied with one vector of the body in LDS: up to 2 666 blobs (beyond: substitution, whose
// working vector lives in HBM then)
bool rbl_block_inverse_large_fits(int64_t n) { return n > BSS_T && sizeof(double) * ((size_t)n + 64 * BIA_W) <= 65536; }
// leading dimension of the two layouts of an explicit inverse: n for small bodies (k_trtri_small), a multiple of 32 entries
// (128 bytes of fp32, 256 of fp64) for large ones so that 64-row segments of a column are whole cache lines
int64_t rbl_block_inverse_ld(int64_t n) { return n <= BSS_T ? n : ((n + 31) / 32) * 32; }
size_t rbl_block_inverse_bytes(int64_t n, int batch) { return sizeof(double) * 2 * (size_t)(rbl_block_inverse_ld(n) * n) * (size_t)batch; }

// X_b = L_b^-1 for `batch` factored bodies (d_L, d_Linv as left by rbl_launch_cholesky_batched); d_X: 2 n^2 doubles a body
int rbl_launch_block_inverse(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_Linv,
                             double *d_X)
{
  if (n > BSS_T) return RBL_ERR_SIZE;
  const int64_t nsteps = (n + IB - 1) / IB;
  const int th = (int)(n <= 128 ? 128 : ((n + 63) / 64) * 64);
  for (int b0 = 0; b0 < batch; b0 += 65535) {        // bodies ride in gridDim.y
    const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
    hipLaunchKernelGGL(k_trtri_small, dim3((unsigned)nsteps, nb), dim3(th), 0, st, d_L + (size_t)b0 * (size_t)strideA, (long)n,
                       (long)strideA, d_Linv + (size_t)b0 * (size_t)(nsteps * IB * IB), (long)(nsteps * IB * IB),
                       d_X + (size_t)b0 * 2 * (size_t)(n * n));
  }
  return RBL_OK;
}

// ---- explicit inverses of LARGE bodies (n > 512: shell_N_642 / 2562, n = 1926 / 7686) -- the reference's own form of the
// block preconditioner (Block_diag_invM, c_rigid_obj.cpp:461-487: Mob.inverse()) ----------------------------------------
// Y = L^-T solves Y L^T = I: the SAME recurrence the factorisation runs on the rows below a panel (X = A21 L11^-T).  So the
// factorisation's own kernels do the inversion on an augmented matrix [L ; I] (2 np x np, np = n rounded up to 32, L padded
// by the identity): per 512-column panel k_trsm_tall solves the rows [0, pend) of the lower half against the panel's diagonal
// block, then k_syrk_mfma (rectangular form: SyrkGrid::full) takes that panel out of the columns to its right,
// Y[0:pend, pend:] -= Y[0:pend, panel] L[pend:, panel]^T -- rows above the panel's last column only (Y is upper triangular:
// a third of the flops of a dense solve, n^3 / 3).  The lower half then IS X^T column-major = X row-major (layout XU of the
// small-body inverses); one tiled transpose writes XL.
template <typename TX>
__global__ __launch_bounds__(256) void k_aug_fill(const double *__restrict__ L, long n, long strideL, double *__restrict__ Aug, long np)
{
  const long col = blockIdx.x;                       // 0 .. np - 1
  const double *Lb = L + (size_t)blockIdx.y * (size_t)strideL;
  double *Ab = Aug + (size_t)blockIdx.y * (size_t)(2 * np * np) + (size_t)col * (size_t)(2 * np);
  for (long i = threadIdx.x; i < 2 * np; i += 256) {
    double v;
    if (i < np) v = (i < n && col < n) ? (i >= col ? Lb[(size_t)col * (size_t)n + i] : 0.0) : (i == col ? 1.0 : 0.0);
    else v = (i - np == col) ? 1.0 : 0.0;
    Ab[i] = v;
  }
}

// lower triangle of X (32 x 32 tiles rb >= cb) from the augmented buffer: XU[r n + c] = XL[c n + r] = X[r][c] = Aug[r][np + c]
template <typename TX>
__global__ __launch_bounds__(256) void k_aug_extract(const double *__restrict__ Aug, long n, long np, long ldx, TX *__restrict__ X)
{
  __shared__ double tile[32][33];
  const long rb = blockIdx.x, cb = blockIdx.y;
  if (cb > rb) return;
  const double *Ab = Aug + (size_t)blockIdx.z * (size_t)(2 * np * np);
  TX *XL = X + (size_t)blockIdx.z * 2 * (size_t)(ldx * n), *XU = XL + (size_t)(ldx * n);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const long r = rb * 32 + k, c = cb * 32 + tx;
    const double v = (r < n && c < n) ? Ab[(size_t)r * (size_t)(2 * np) + np + c] : 0.0;
    tile[k][tx] = v;
    if (r < n && c < n && c <= r) XU[(size_t)r * (size_t)ldx + c] = (TX)v;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const long c = cb * 32 + k, r = rb * 32 + tx;
    if (r < n && c < n && c <= r) XL[(size_t)c * (size_t)ldx + r] = (TX)tile[tx][k];
  }
}

size_t rbl_block_inverse_large_aug_bytes(int64_t n, int batch, int *chunk_out)
{
  const int64_t np = ((n + IB - 1) / IB) * IB;
  const size_t per = sizeof(double) * 2 * (size_t)(np * np);
  int chunk = (int)(((size_t)4 << 30) / per);        // ~4 GB of scratch at a time
  if (chunk < 1) chunk = 1;
  if (chunk > batch) chunk = batch;
  if (chunk > 65535) chunk = 65535;
  if (chunk_out) *chunk_out = chunk;
  return per * (size_t)chunk;
}

// d_X: [XL | XU] per body, 2 n^2 entries of fp64 (d_X) and / or fp32 (d_Xf), either may be NULL; d_aug: scratch of
// rbl_block_inverse_large_aug_bytes
int rbl_launch_block_inverse_large(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_Linv,
                                   double *d_X, float *d_Xf, double *d_aug)
{
  const int64_t np = ((n + IB - 1) / IB) * IB, nsteps = np / IB, ld = 2 * np;
  const long strideL = (long)(nsteps * IB * IB), strideAug = (long)(ld * np);
  if ((size_t)NBB_INV * (size_t)ld * 8 >= ((size_t)1 << 31)) return RBL_ERR_SIZE;   // one panel through one buffer descriptor
  int chunk = 1;
  (void)rbl_block_inverse_large_aug_bytes(n, batch, &chunk);
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int nb = batch - b0 < chunk ? batch - b0 : chunk;
    const double *Lb = d_L + (size_t)b0 * (size_t)strideA;
    const double *Li = d_Linv + (size_t)b0 * (size_t)strideL;
    hipLaunchKernelGGL(k_aug_fill<double>, dim3((unsigned)np, nb), dim3(256), 0, st, Lb, (long)n, (long)strideA, d_aug, (long)np);
    for (int64_t k = 0; k < np; k += NBB_INV) {
      const int64_t pw = (np - k < NBB_INV) ? (np - k) : NBB_INV, pend = k + pw;
      const double *Lk = Li + (size_t)(k / IB) * IB * IB;
      hipLaunchKernelGGL(k_trsm_tall, dim3((unsigned)((pend + 127) / 128), 1, nb), dim3(256), 0, st, d_aug, (long)ld, (long)(np + pend),
                         (long)k, (int)(pw / IB), (long)np, Lk, strideAug, strideL);
      if (pend < np) {
        if (!syrk_K_ok(pw)) return RBL_ERR_ARG;
        const SyrkGrid G = syrk_grid_rect(pend, np - pend, np - pend);
        hipLaunchKernelGGL(k_syrk_mfma, dim3(G.nwg, 1, nb), dim3(256), 0, st, d_aug, (long)ld, (long)pend, (long)np, (long)k, (int)pw,
                           strideAug, (long)(np + pend), G);
      }
    }
    const dim3 eg((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), nb);
    const int64_t ldx = rbl_block_inverse_ld(n);
    if (d_X) hipLaunchKernelGGL(k_aug_extract<double>, eg, dim3(256), 0, st, (const double *)d_aug, (long)n, (long)np, (long)ldx,
                                d_X + (size_t)b0 * 2 * (size_t)(ldx * n));
    if (d_Xf) hipLaunchKernelGGL(k_aug_extract<float>, eg, dim3(256), 0, st, (const double *)d_aug, (long)n, (long)np, (long)ldx,
                                 d_Xf + (size_t)b0 * 2 * (size_t)(ldx * n));
  }
  return RBL_OK;
}

// mode 0: x = X^T X v ;  1: x = X v ;  2: x = X^T v  (= (L L^T)^-1 v, L^-1 v, L^-T v).  d_tmp: nv vectors of
// rhs_pitch doubles (mode 0 only; laid out like d_out).  In place is fine for mode 0, not for 1 / 2.
// d_Q != NULL: d_X is ONE body-frame inverse shared by all bodies and the factor meant is G_b = R_b L (G G^T = M_b):
// G^-1 v = X R^T v, G^-T v = R X^T v, (G G^T)^-1 v = R X^T X R^T v  (d_Q: the bodies' quaternions, first body first).
// f32: d_X points to the single-precision copy ([XL | XU] of floats).
template <typename TM>
static int block_inv_apply_t(hipStream_t st, const TM *d_X, int64_t n, int batch, const double *d_in, double *d_out,
                             int64_t vec_stride, int nv, int64_t rhs_pitch, int mode, double *d_tmp, const double *d_Q)
{
  const int64_t ldx = rbl_block_inverse_ld(n);
  if (batch > 65535) {                               // bodies ride in gridDim.y: more than that go in several rounds
    for (int b0 = 0; b0 < batch; b0 += 65535) {
      const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
      const size_t vo = (size_t)b0 * (size_t)vec_stride;
      const int rc = block_inv_apply_t<TM>(st, d_Q ? d_X : d_X + (size_t)b0 * 2 * (size_t)(ldx * n), n, nb, d_in + vo, d_out + vo,
                                           vec_stride, nv, rhs_pitch, mode, d_tmp ? d_tmp + vo : nullptr,
                                           d_Q ? d_Q + 4 * (size_t)b0 : nullptr);
      if (rc) return rc;
    }
    return RBL_OK;
  }
  const bool r64 = !d_Q && ldx % 16 == 0 && n > BSS_T;          // aligned 64-row segments (large per-configuration inverses)
  const int R = r64 ? 64 : BIA_RQ;
  const dim3 grid((unsigned)((n + R - 1) / R), batch);
  const long mstride = d_Q ? 0 : 2 * (long)(ldx * n);
  int gmax = 3;                                      // vectors sharing one pass over X: what 64 KB of LDS hold
  while (gmax > 1 && sizeof(double) * ((size_t)gmax * (size_t)n + 64 * BIA_W * (size_t)gmax) > 65536) --gmax;
  auto pass = [&](const double *in, double *out, int upper) {
    const int rot = d_Q ? (upper ? 2 : 1) : 0;
    for (int v0 = 0; v0 < nv;) {
      const int g = nv - v0 >= gmax ? gmax : nv - v0;
      const size_t lds = sizeof(double) * ((size_t)g * (size_t)n + 64 * BIA_W * (size_t)g);
      const double *pi = in + (size_t)v0 * (size_t)rhs_pitch;
      double *po = out + (size_t)v0 * (size_t)rhs_pitch;
#define RBL_BIA_LAUNCH(NVV, RR) hipLaunchKernelGGL((k_block_inv_apply<NVV, TM, RR>), grid, dim3(64 * BIA_W), lds, st, d_X, (long)n, (long)ldx, \
                                                   mstride, pi, po, (long)vec_stride, (long)rhs_pitch, upper, d_Q, rot)
      if (r64) { if (g == 3) RBL_BIA_LAUNCH(3, 64); else if (g == 2) RBL_BIA_LAUNCH(2, 64); else RBL_BIA_LAUNCH(1, 64); }
      else { if (g == 3) RBL_BIA_LAUNCH(3, BIA_RQ); else if (g == 2) RBL_BIA_LAUNCH(2, BIA_RQ); else RBL_BIA_LAUNCH(1, BIA_RQ); }
#undef RBL_BIA_LAUNCH
      v0 += g;
    }
  };
  if (mode == 0) { pass(d_in, d_tmp, 0); pass(d_tmp, d_out, 1); }
  else pass(d_in, d_out, mode == 1 ? 0 : 1);
  return RBL_OK;
}

// ---------------------------------------------------------------------------
// ONE matrix, many vectors (round 4): in free space every body's factor / inverse is the SAME body-frame matrix seen through
// the body's rotation, so applying it to the N_bod x nv vectors of a sweep is a matrix-MATRIX product  Y = Op [x_1 .. x_C]  with
// C = N_bod nv columns -- 2 n^2 C flops, the matrix read ONCE.  The batched matrix-vector kernel above re-read the 1.9 MB table
// for every body (98 MB through L2 per application at 50 x 486: 12.8 us at 7.6 TB/s); this kernel is the same operation on the
// fp64 matrix cores: workgroup = 48 rows (16 whole blobs: the output rotation needs whole blobs) x 16 columns, the rotated
// input panel staged in LDS once, K split over eight waves (v_mfma_f64_16x16x4, three row tiles per wave, every operand load
// in flight before the first wait), partial tiles
// added in wave order, rotated back per blob and stored.  Op = A restricted to k <= m (TRI = 1: X = L^-1, column-major),
// k >= m (TRI = 2: X^T, read from the row-major copy) or full (TRI = 0: the symmetric M_body^-1 table).
// rot & 1: x <- R_b^T x per blob before, rot & 2: y <- R_b y after.  Column c = b nv + v reads in + v rhs_pitch + b vec_stride.
// ---------------------------------------------------------------------------
constexpr int SG_M = 48, SG_N = 16, SG_W = 8, SG_KW = 16;     // 8 waves per workgroup, <= 16 k steps (64 rows of the input) per wave

template <int TRI>
__global__ __launch_bounds__(64 * SG_W) void k_shared_gemm(const double *__restrict__ A, long n, long lda, const double *in, double *out,
                                                          long vec_stride, long rhs_pitch, int ncol, int nv,
                                                          const double *__restrict__ Q, int rot)
{
  extern __shared__ double sg[];                     // rotated inputs sB[k][16], k < kpad; afterwards red[SG_W][48][16]
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const long m0 = (long)blockIdx.x * SG_M;
  const int c0 = (int)blockIdx.y * SG_N;
  // rows of the input this row tile multiplies: [klo, khi), widened to whole blobs and to multiples of 4
  long klo = (TRI == 2) ? m0 : 0, khi = (TRI == 1) ? ((m0 + SG_M < n) ? m0 + SG_M : n) : n;
  klo = (klo / 12) * 12;
  const long kpad = ((khi - klo + 3) / 4) * 4;
  const long nks = kpad / 4;                         // k steps of the tile, dealt to the waves round-robin: wave w takes w, w + 8, ...
  // A operands first: EVERY load of this wave is in flight before anything waits (the whole kernel is two memory round trips:
  // these and the input panel below; a loop of load -> MFMA batches was one round trip per batch and no faster than the
  // matrix-vector form it replaces)
  double a[SG_KW][3];
#pragma unroll
  for (int u = 0; u < SG_KW; ++u) {
    const long ks = w + (long)SG_W * u;
    const long k = klo + 4 * ks + l4;                // this lane's row of the input = column of Op
    const bool kin = ks < nks && k < n;
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
      const long m = m0 + 16 * ti + l15;
      const bool ok = kin && m < n && (TRI == 0 || (TRI == 1 ? k <= m : k >= m));
      const double x = A[(size_t)(kin ? k : 0) * (size_t)lda + (m < n ? m : 0)];     // unconditional (clamped) load, masked after
      a[u][ti] = ok ? x : 0.0;
    }
  }
  const int col = t & 15;                            // this thread's column while staging and while storing
  const int c = c0 + col;
  const bool live = c < ncol;
  const int b = live ? c / nv : 0, v = live ? c - b * nv : 0;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (rot && live) quat_rot_d(Q + 4 * (size_t)b, R);
  const double *vin = in + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride;
  for (long q = klo / 3 + (t >> 4); 3 * q < klo + kpad; q += 4 * SG_W) {   // blob q of column `col`
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (live && 3 * q + 2 < n) { a0 = vin[3 * q]; a1 = vin[3 * q + 1]; a2 = vin[3 * q + 2]; }
    double d0 = a0, d1 = a1, d2 = a2;
    if (rot & 1) {                                   // R^T a
      d0 = R[0] * a0 + R[3] * a1 + R[6] * a2;
      d1 = R[1] * a0 + R[4] * a1 + R[7] * a2;
      d2 = R[2] * a0 + R[5] * a1 + R[8] * a2;
    }
    const long r0 = 3 * q - klo;
    if (r0 < kpad) sg[r0 * SG_N + col] = d0;
    if (r0 + 1 < kpad) sg[(r0 + 1) * SG_N + col] = d1;
    if (r0 + 2 < kpad) sg[(r0 + 2) * SG_N + col] = d2;
  }
  __syncthreads();
  double4_t acc[3];
#pragma unroll
  for (int ti = 0; ti < 3; ++ti) acc[ti] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < SG_KW; ++u) {
    const long ks = w + (long)SG_W * u;
    const double bb = ks < nks ? sg[(4 * ks + l4) * SG_N + l15] : 0.0;
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti], bb, acc[ti], 0, 0, 0);
  }
  __syncthreads();                                   // the input panel is dead: its LDS holds the partial tiles now
  double *red = sg;                                  // red[w][row][col]
#pragma unroll
  for (int ti = 0; ti < 3; ++ti)
#pragma unroll
    for (int vv = 0; vv < 4; ++vv) red[((size_t)w * SG_M + 16 * ti + l4 + 4 * vv) * SG_N + l15] = acc[ti][vv];   // D: row = l4 + 4 v, col = l15
  __syncthreads();
  const int q = t >> 4;                              // blob q of the tile (rows 3 q .. 3 q + 2), column `col`
  if (q < SG_M / 3 && live && m0 + 3 * q + 2 < n) {
    double y[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      double sum = red[(size_t)(3 * q + d) * SG_N + col];
#pragma unroll
      for (int ww = 1; ww < SG_W; ++ww) sum += red[((size_t)ww * SG_M + 3 * q + d) * SG_N + col];   // waves in fixed order
      y[d] = sum;
    }
    double *o = out + (size_t)v * (size_t)rhs_pitch + (size_t)b * (size_t)vec_stride + m0 + 3 * q;
    if (rot & 2) {                                   // R y
      o[0] = R[0] * y[0] + R[1] * y[1] + R[2] * y[2];
      o[1] = R[3] * y[0] + R[4] * y[1] + R[5] * y[2];
      o[2] = R[6] * y[0] + R[7] * y[1] + R[8] * y[2];
    } else { o[0] = y[0]; o[1] = y[1]; o[2] = y[2]; }
  }
}

// tri: 0 full, 1 lower (k <= m), 2 upper (k >= m).  n a multiple of 3, n <= 512 (the input panel of a row tile in 64 KB of LDS)
bool rbl_shared_gemm_fits(int64_t n) { return n % 3 == 0 && n >= 48 && (n + 3) / 4 <= (int64_t)SG_W * SG_KW && (size_t)(((n + 3) / 4) * 4) * SG_N * sizeof(double) <= 65536; }

int rbl_launch_shared_gemm(hipStream_t st, const double *d_A, int64_t n, int64_t lda, int tri, const double *d_in, double *d_out,
                           int64_t vec_stride, int64_t rhs_pitch, int nbod, int nv, const double *d_Q, int rot)
{
  if (!rbl_shared_gemm_fits(n) || d_in == d_out) return RBL_ERR_ARG;
  const int ncol = nbod * nv;
  if (ncol <= 0) return RBL_OK;
  const dim3 grid((unsigned)((n + SG_M - 1) / SG_M), (unsigned)((ncol + SG_N - 1) / SG_N));
  size_t lds = (size_t)(((n + 3) / 4) * 4 + 12) * SG_N * sizeof(double);
  const size_t red = (size_t)SG_W * SG_M * SG_N * sizeof(double);
  if (lds < red) lds = red;
  if (lds > 65536) lds = 65536;
  if (tri == 1) hipLaunchKernelGGL(k_shared_gemm<1>, grid, dim3(64 * SG_W), lds, st, d_A, (long)n, (long)lda, d_in, d_out, (long)vec_stride, (long)rhs_pitch, ncol, nv, d_Q, rot);
  else if (tri == 2) hipLaunchKernelGGL(k_shared_gemm<2>, grid, dim3(64 * SG_W), lds, st, d_A, (long)n, (long)lda, d_in, d_out, (long)vec_stride, (long)rhs_pitch, ncol, nv, d_Q, rot);
  else hipLaunchKernelGGL(k_shared_gemm<0>, grid, dim3(64 * SG_W), lds, st, d_A, (long)n, (long)lda, d_in, d_out, (long)vec_stride, (long)rhs_pitch, ncol, nv, d_Q, rot);
  return RBL_OK;
}

int rbl_launch_block_inv_apply(hipStream_t st, const double *d_X, int64_t n, int batch, const double *d_in, double *d_out,
                               int64_t vec_stride, int nv, int64_t rhs_pitch, int mode, double *d_tmp, const double *d_Q, int f32)
{
  if (sizeof(double) * ((size_t)n + 64 * BIA_W) > 65536) return RBL_ERR_SIZE;      // one vector + partial sums in LDS
  if (mode == 0 && !d_tmp) return RBL_ERR_ARG;
  if (mode != 0 && d_in == d_out) return RBL_ERR_ARG;
  const bool no_gemm = (f32 & 2) != 0;               // RBL_OPT_SHARED_GEMM = 0: the batched matrix-vector form
  f32 &= 1;
  if (d_Q && !f32 && !no_gemm && rbl_shared_gemm_fits(n) && vec_stride == n) {
    // ONE body-frame matrix for every body (free space): the sweep is a matrix-matrix product on the fp64 matrix cores
    const int64_t ldx = rbl_block_inverse_ld(n);
    const double *XL = d_X, *XU = d_X + (size_t)(ldx * n);
    if (mode == 0) {
      int rc = rbl_launch_shared_gemm(st, XL, n, ldx, 1, d_in, d_tmp, vec_stride, rhs_pitch, batch, nv, d_Q, 1);
      return rc ? rc : rbl_launch_shared_gemm(st, XU, n, ldx, 2, d_tmp, d_out, vec_stride, rhs_pitch, batch, nv, d_Q, 2);
    }
    return mode == 1 ? rbl_launch_shared_gemm(st, XL, n, ldx, 1, d_in, d_out, vec_stride, rhs_pitch, batch, nv, d_Q, 1)
                     : rbl_launch_shared_gemm(st, XU, n, ldx, 2, d_in, d_out, vec_stride, rhs_pitch, batch, nv, d_Q, 2);
  }
  if (f32) return block_inv_apply_t<float>(st, (const float *)d_X, n, batch, d_in, d_out, vec_stride, nv, rhs_pitch, mode, d_tmp, d_Q);
  return block_inv_apply_t<double>(st, d_X, n, batch, d_in, d_out, vec_stride, nv, rhs_pitch, mode, d_tmp, d_Q);
}

// y_b = L_b x_b as a triangular matrix-vector product with every row independent (the kernel of the inverse applications
// on the factor itself; k_block_trmv walks the columns in one workgroup per body: 50 us at n = 486, and one CU per body
// whatever the size).  strideA = 0: one shared body-frame factor, then d_Q rotates the result (y_b = R_b L x_b).  Not in place.
int rbl_launch_block_trmv_small(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_in,
                                double *d_out, int64_t vec_stride, const double *d_Q)
{
  if (sizeof(double) * ((size_t)n + 64 * BIA_W) > 65536 || d_in == d_out) return RBL_ERR_ARG;
  for (int b0 = 0; b0 < batch; b0 += 65535) {
    const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
    const size_t vo = (size_t)b0 * (size_t)vec_stride;
    hipLaunchKernelGGL((k_block_inv_apply<1, double, BIA_RQ>), dim3((unsigned)((n + BIA_RQ - 1) / BIA_RQ), nb), dim3(64 * BIA_W),
                       sizeof(double) * ((size_t)n + 64 * BIA_W), st, d_L + (size_t)b0 * (size_t)strideA, (long)n, (long)n, (long)strideA,
                       d_in + vo, d_out + vo, (long)vec_stride, 0L, 0, d_Q ? d_Q + 4 * (size_t)b0 : nullptr, d_Q ? 2 : 0);
  }
  return RBL_OK;
}

// nv vectors (rhs_pitch apart) of `batch` bodies with N_blb blobs each: every blob's three entries rotated by its body's
// R (transpose = 0) or R^T (1).  In place is fine.
void rbl_launch_rotate_bodies(hipStream_t st, const double *d_Q, const double *d_in, double *d_out, int N_blb, int batch, int nv,
                              int64_t rhs_pitch, int transpose)
{
  const long nblobs = (long)N_blb * batch;
  if (nblobs <= 0 || nv <= 0) return;
  hipLaunchKernelGGL(k_rotate_bodies, dim3((unsigned)((nblobs + 255) / 256), nv), dim3(256), 0, st, d_Q, d_in, d_out, N_blb,
                     nblobs, (long)rhs_pitch, transpose);
}

// y_b = L_b x_b for every matrix of the batch (lower-triangular product): one workgroup per matrix, x in LDS,
// column sweep with rows spread over the threads.
// (GLOBAL: bodies of more than 2 730 blobs -- x is read where it lies, through the CU's caches)
template <bool GLOBAL>
__global__ __launch_bounds__(BS_T) void k_block_trmv(const double *__restrict__ L, long n, long strideA,
                                                     const double *__restrict__ in, double *__restrict__ out,
                                                     long vec_stride)
{
  extern __shared__ double x_lds[];
  const int b = blockIdx.x, t = threadIdx.x;
  const double *Lb = L + (size_t)b * (size_t)strideA;
  const double *v = in + (size_t)b * (size_t)vec_stride;
  const double *x = GLOBAL ? v : x_lds;
  if (!GLOBAL) {
    for (long e = t; e < n; e += BS_T) x_lds[e] = v[e];
    __syncthreads();
  }
  double *o = out + (size_t)b * (size_t)vec_stride;
  for (long r = t; r < n; r += BS_T) {
    double a0 = 0.0, a1 = 0.0;
    const double *row = Lb + r;
    long c = 0;
    for (; c + 1 <= r; c += 2) {
      a0 = __builtin_fma(row[(size_t)c * n], x[c], a0);
      a1 = __builtin_fma(row[(size_t)(c + 1) * n], x[c + 1], a1);
    }
    if (c <= r) a0 = __builtin_fma(row[(size_t)c * n], x[c], a0);
    o[r] = a0 + a1;
  }
}

int rbl_launch_block_trmv(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA, const double *d_in,
                          double *d_out, int64_t vec_stride)
{
  if (n > SOLVE_MAXN)
    hipLaunchKernelGGL(k_block_trmv<true>, dim3(batch), dim3(BS_T), 0, st, d_L, (long)n, (long)strideA, d_in, d_out, (long)vec_stride);
  else
    hipLaunchKernelGGL(k_block_trmv<false>, dim3(batch), dim3(BS_T), sizeof(double) * (size_t)n, st, d_L, (long)n, (long)strideA,
                       d_in, d_out, (long)vec_stride);
  return RBL_OK;
}

// mode 0: x = (L L^T)^-1 v ;  1: x = L^-1 v ;  2: x = L^-T v  (| 0x100: ONE factor shared by the whole batch).  nv vectors per matrix, rhs_pitch doubles apart: up to three
// share one pass over L (LDS: nv (n + 32) doubles <= 64 KB), more are done in groups.  In place (d_out == d_in) is fine.
int rbl_launch_block_solve_multi(hipStream_t st, const double *d_L, int64_t n, int batch, int64_t strideA,
                                 const double *d_Linv, const double *d_in, double *d_out, int64_t vec_stride, int nv,
                                 int64_t rhs_pitch, int mode, const double *d_Q)
{
  // d_Q (with | 0x100 and n <= 512 only): the shared factor is a body-frame one, G_b = R_b L -- rotations fused into the sweep
  const int64_t nsteps = (n + IB - 1) / IB;
  const bool shared = (mode & 0x100) != 0;           // one factor for every body of the batch (strideA = 0 by the caller)
  const bool classic = (mode & 0x200) != 0;          // RBL_OPT_BLOCK_SOLVE_PIPE = 0: the two-barrier kernel (k_block_solve) for A/B runs
  mode &= 0xff;
  const long strideL = shared ? 0 : (long)(nsteps * IB * IB);
  for (int v0 = 0; v0 < nv;) {
    int g = nv - v0 >= 3 ? 3 : nv - v0;
    while (g > 1 && (size_t)g * (size_t)(n + IB) * sizeof(double) > 65536) --g;
    const size_t lds = sizeof(double) * (size_t)g * (size_t)(n + IB);
    const double *in = d_in + (size_t)v0 * (size_t)rhs_pitch;
    double *out = d_out + (size_t)v0 * (size_t)rhs_pitch;
    if (lds > 65536) {         // one vector of a body does not fit LDS (> 2 730 blobs): the output vector in HBM is the working vector
      if (d_Q) return RBL_ERR_ARG;
      hipLaunchKernelGGL((k_block_solve<1, true>), dim3(batch), dim3(BS_T), sizeof(double) * IB, st, d_L, (long)n, (long)strideA, d_Linv,
                         strideL, in, out, (long)vec_stride, (long)rhs_pitch, mode);
      v0 += 1;
      continue;
    }
    if (d_Q && !(shared && n <= BSS_T)) return RBL_ERR_ARG;
    if (n <= BSS_T) {          // small bodies: one row per thread, factor entries prefetched a step ahead
      const int th = (int)(n <= 128 ? 128 : ((n + 63) / 64) * 64);
      const int rot = d_Q ? ((mode != 2 ? 1 : 0) | (mode != 1 ? 2 : 0)) : 0;
      if (g == 3)
        hipLaunchKernelGGL(k_block_solve_small<3>, dim3(batch), dim3(th), lds, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in,
                           out, (long)vec_stride, (long)rhs_pitch, mode, d_Q, rot);
      else if (g == 2)
        hipLaunchKernelGGL(k_block_solve_small<2>, dim3(batch), dim3(th), lds, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in,
                           out, (long)vec_stride, (long)rhs_pitch, mode, d_Q, rot);
      else
        hipLaunchKernelGGL(k_block_solve_small<1>, dim3(batch), dim3(th), lds, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in,
                           out, (long)vec_stride, (long)rhs_pitch, mode, d_Q, rot);
      v0 += g;
      continue;
    }
    if (!classic && n >= 3 * IB) {   // the one-barrier pipeline (k_block_solve_pipe); its LDS: zero-padded vectors + x of two steps + the diag wave's ring
      auto pipe_lds = [&](int nvp) { return sizeof(double) * ((size_t)nvp * (size_t)(nsteps * IB + 2 * IB) + (size_t)BP_RING * BP_SLOT); };
      int gp = g;
      while (gp > 1 && pipe_lds(gp) > BP_LDS_MAX) --gp;
      const size_t ldp = pipe_lds(gp);
      if (ldp <= BP_LDS_MAX) {
        static const bool lds_raised = [] {          // more than the default 64 KB of dynamic LDS: asked for once per kernel
          bool ok = hipFuncSetAttribute((const void *)k_block_solve_pipe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BP_LDS_MAX) == hipSuccess;
          ok = hipFuncSetAttribute((const void *)k_block_solve_pipe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BP_LDS_MAX) == hipSuccess && ok;
          return hipFuncSetAttribute((const void *)k_block_solve_pipe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BP_LDS_MAX) == hipSuccess && ok;
        }();
        if (!lds_raised) return RBL_ERR_HIP;
        if (gp == 3)
          hipLaunchKernelGGL(k_block_solve_pipe<3>, dim3(batch), dim3(BS_T), ldp, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in,
                             out, (long)vec_stride, (long)rhs_pitch, mode);
        else if (gp == 2)
          hipLaunchKernelGGL(k_block_solve_pipe<2>, dim3(batch), dim3(BS_T), ldp, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in,
                             out, (long)vec_stride, (long)rhs_pitch, mode);
        else
          hipLaunchKernelGGL(k_block_solve_pipe<1>, dim3(batch), dim3(BS_T), ldp, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in,
                             out, (long)vec_stride, (long)rhs_pitch, mode);
        v0 += gp;
        continue;
      }
    }
    if (g == 3)
      hipLaunchKernelGGL(k_block_solve<3>, dim3(batch), dim3(BS_T), lds, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in, out,
                         (long)vec_stride, (long)rhs_pitch, mode);
    else if (g == 2)
      hipLaunchKernelGGL(k_block_solve<2>, dim3(batch), dim3(BS_T), lds, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in, out,
                         (long)vec_stride, (long)rhs_pitch, mode);
    else
      hipLaunchKernelGGL(k_block_solve<1>, dim3(batch), dim3(BS_T), lds, st, d_L, (long)n, (long)strideA, d_Linv, strideL, in, out,
                         (long)vec_stride, (long)rhs_pitch, mode);
    v0 += g;
  }
  return RBL_OK;
}

#if defined(RBL_PIPE_PROF)
extern "C" __attribute__((visibility("default"))) int rbl_debug_pipe_prof(unsigned long long *out, int clear)
{
  if (clear) {
    static unsigned long long z[256 * 512];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_prof), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pipe_prof), sizeof(unsigned long long) * 256 * 512);
}
#endif

size_t rbl_trmv_part_bytes(int64_t n)
{
  const int64_t nch = (n + TR_COLS - 1) / TR_COLS;
  return (size_t)nch * (size_t)n * sizeof(double);
}

void rbl_launch_trmv_lower(hipStream_t st, const double *d_L, int64_t n, const double *d_W,
                           double *d_out, double *d_part)
{
  if (n <= 0) return;
  const int nch = (int)((n + TR_COLS - 1) / TR_COLS);
  dim3 grid((unsigned)((n + TR_ROWS - 1) / TR_ROWS), (unsigned)nch);
  hipLaunchKernelGGL(k_trmv_partial, grid, dim3(TR_ROWS), 0, st, d_L, (long)n, d_W, d_part);
  hipLaunchKernelGGL(k_trmv_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_part,
                     (long)n, nch, d_out);
}
