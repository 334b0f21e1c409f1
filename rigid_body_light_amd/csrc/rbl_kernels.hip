// rbl_kernels.hip -- hand-written gfx950 kernels of the blob-mobility hot path.
//
//   k_apply_M_sym<WALL,NI>  matrix-free U = [B] M [B] F, every UNORDERED pair once (default):
//                           wave-private LDS tile, systolic lane<->column pairing, ds_add_f64
//                           column sums, slabs reduced by k_reduce_sym   (reference :641-659,:413-459)
//   k_apply_M<WALL>         the same product by ordered pairs for a ROW RANGE (row sharding,
//                           very large N): j tiles staged in LDS, lane = row, LDS broadcast;
//                           j-split partial slabs reduced by k_reduce_parts
//   k_apply_M_mrhs<WALL>    16 right-hand sides: pair blocks on the VALU, nine v_mfma_f64_16x16x4
//                           per step apply them (k_pack_rhs / k_unpack_rhs transpose the vectors)
//   k_build_M<WALL>         dense column-major assembly (reference :413-459), HBM-write bound,
//                           reference-order arithmetic -> bit-identical entries; batched over bodies
//   k_blob_positions        r_k = R(Q_b) c_k + X_b                       (reference :257-293)
//   k_pair_blocks           test hook: independent 3x3 blocks
//   k_normal                counter-based N(0,1) generator (replaces clock-seeded :730-741)
//   BLAS-1 helpers for Lanczos (deterministic two-stage reductions)
// All fp64-VALU-bound kernels share the pair arithmetic of rbl_pair.hpp.  No float atomics on
// global memory anywhere: every sum has a fixed order, results are bitwise reproducible.
#include "rbl_internal.hpp"
#include "rbl_pair_pk.hpp"

#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <cmath>

#ifdef RBL_WAVE_TRACE
extern __device__ unsigned long long *g_wave_trace;   // tools/wave_trace.hip
#endif

namespace {

constexpr int TB = 256;  // threads per block = j-tile length

struct __attribute__((aligned(16))) JBlob {
  double x, y, z, fx, fy, fz;  // 48 B: three 16-B LDS broadcasts per j
};

// parameter set of the radius-scaled coordinates (x/a): the mobility entries depend on r/a only, so the
// matvec kernels scale positions once when they load them and run the pair arithmetic with a = 1
__device__ __forceinline__ RblParams unit_params(const RblParams &P)
{
  RblParams Pu = P;
  Pu.a = 1.0; Pu.inv_a = 1.0; Pu.four_a2 = 4.0; Pu.tiny2 = 1e-24; Pu.c_near_A = -0.375; Pu.c_near_B = 0.125;
  return Pu;
}

__device__ __forceinline__ double damp_of(const RblParams &P, double z)
{
  if (P.no_damp) return 1.0;
  return (z >= P.a) ? 1.0 : z / P.a;  // c_rigid_obj.cpp:629-633
}

// ---------------------------------------------------------------------------
// Matrix-free matvec.  grid = (i-tiles, jsplit).  Block (bx,by) owns rows
// row_begin + bx*TB .. and columns [by*jchunk, (by+1)*jchunk).
// jsplit == 1 : writes final U (scaled, damped) to out[3*(i-row_begin)]
// jsplit  > 1 : writes raw partial sums to part[by][3*(i-row_begin)]
// ---------------------------------------------------------------------------
template <bool WALL, bool SELF>
__device__ __forceinline__ void sweep_tile(const RblParams &P, const JBlob *sj, double xi,
                                           double yi, double zi, int self_jj, double &ux,
                                           double &uy, double &uz, unsigned &flags, const RblWallK &K)
{
#pragma unroll 4
  for (int jj = 0; jj < TB; ++jj) {
    const JBlob b = sj[jj];
    rbl_pair_accum<WALL, SELF, true>(P, xi, yi, zi, b.x, b.y, b.z, b.fx, b.fy, b.fz,
                                     SELF && (jj == self_jj), ux, uy, uz, flags, K);
  }
}

template <bool WALL>
__global__ __launch_bounds__(TB) void k_apply_M(const double *__restrict__ r,
                                                const double *__restrict__ F,
                                                double *__restrict__ out,
                                                double *__restrict__ part, long N,
                                                long row_begin, long row_end, long jchunk,
                                                int jsplit, RblParams P, unsigned *err)
{
  __shared__ JBlob sj[TB];
  const int t = threadIdx.x;
  const long i_tile0 = row_begin + (long)blockIdx.x * TB;
  const long i = i_tile0 + t;
  const bool valid = i < row_end;
  const long ic = valid ? i : row_end - 1;
  const double zi_phys = r[3 * ic + 2];
  const double xi = r[3 * ic] * P.inv_a, yi = r[3 * ic + 1] * P.inv_a, zi = zi_phys * P.inv_a;   // radius-scaled
  const RblParams Pu = unit_params(P);
  const RblWallK WK = rbl_wall_k_resident();
  unsigned flags = 0;
  if (WALL && zi < 0.0) flags |= RBL_FLAG_BELOW_WALL;

  const long j_begin = (long)blockIdx.y * jchunk;
  const long j_end = (j_begin + jchunk < N) ? j_begin + jchunk : N;
  double ux = 0.0, uy = 0.0, uz = 0.0;

  for (long j0 = j_begin; j0 < j_end; j0 += TB) {
    // ---- stage one j-tile (positions + damped forces) into LDS -------------
    const long j = j0 + t;
    JBlob b;
    if (j < j_end) {
      b.x = r[3 * j]; b.y = r[3 * j + 1]; b.z = r[3 * j + 2];
      double d = 1.0;
      if (WALL) {
        if (b.z < 0.0) flags |= RBL_FLAG_BELOW_WALL;
        d = damp_of(P, b.z);
      }
      b.x *= P.inv_a; b.y *= P.inv_a; b.z *= P.inv_a;
      b.fx = d * F[3 * j]; b.fy = d * F[3 * j + 1]; b.fz = d * F[3 * j + 2];
    } else {  // padding: zero force, far away, above the wall
      b.x = 1.0e15; b.y = 1.0e15; b.z = 1.0; b.fx = 0.0; b.fy = 0.0; b.fz = 0.0;
    }
    __syncthreads();  // previous tile fully consumed
    sj[t] = b;
    __syncthreads();
    // ---- sweep: does this tile contain any of the block's own rows? --------
    const bool diag = (j0 < i_tile0 + TB) && (j0 + TB > i_tile0);  // block-uniform
    if (diag) {
      const long sjj = i - j0;  // index of i itself inside the tile (may be out of range)
      const int self_jj = (valid && sjj >= 0 && sjj < TB) ? (int)sjj : -1;
      sweep_tile<WALL, true>(Pu, sj, xi, yi, zi, self_jj, ux, uy, uz, flags, WK);
    } else {
      sweep_tile<WALL, false>(Pu, sj, xi, yi, zi, -1, ux, uy, uz, flags, WK);
    }
  }

  if (valid) {
    const long o = 3 * (i - row_begin);
    if (jsplit == 1) {
      const double sc = WALL ? P.nf * damp_of(P, zi_phys) : P.nf;
      out[o] = sc * ux; out[o + 1] = sc * uy; out[o + 2] = sc * uz;
      if (!(isfinite(ux) && isfinite(uy) && isfinite(uz))) flags |= RBL_FLAG_NONFINITE;
    } else {
      double *p = part + (size_t)blockIdx.y * (size_t)(3 * (row_end - row_begin)) + o;
      p[0] = ux; p[1] = uy; p[2] = uz;
    }
  } else {
    flags &= ~RBL_FLAG_OVERLAP;  // clamped duplicate row: ignore its pair checks
  }
  if (flags) atomicOr(err, flags);
}

template <bool WALL>
__global__ __launch_bounds__(256) void k_reduce_parts(const double *__restrict__ part,
                                                      const double *__restrict__ r,
                                                      double *__restrict__ out, long row_begin,
                                                      long nrows, int jsplit, RblParams P,
                                                      unsigned *err)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over 3*nrows
  if (idx >= 3 * nrows) return;
  double s = 0.0;
  for (int k = 0; k < jsplit; ++k) s += part[(size_t)k * (size_t)(3 * nrows) + idx];
  double sc = P.nf;
  if (WALL) sc *= damp_of(P, r[3 * (row_begin + idx / 3) + 2]);
  out[idx] = sc * s;
  if (!isfinite(s)) atomicOr(err, (unsigned)RBL_FLAG_NONFINITE);
}


// ---------------------------------------------------------------------------
// Symmetric matrix-free matvec (variant 1): every UNORDERED pair is evaluated once.
// Work unit = one wavefront = (I-tile of 64 rows) x (chunk of C column tiles, J >= I).
//   J == I : ordered sweep with the index-equality self term (broadcast LDS reads)
//   J  > I : "systolic" sweep -- at step s lane l pairs its row i with column
//            j = (l+s)&63, so the 64 lanes always touch 64 DIFFERENT j: j-data is read
//            from LDS conflict-free and M_ji F_i is added to the j accumulator in LDS
//            with ds_add_f64 (one lane per address per step -> deterministic order).
// Row sums go to slabI[chunk][i], column sums to slabJ[I-row][j]; k_reduce_sym adds
// them in a fixed order (no global atomics -> bitwise reproducible).
// (i_first, i_step) selects the I-tiles of this launch (multi-GPU: rank, world size; see sym_row_of).
// ---------------------------------------------------------------------------
constexpr int TS = 64;

// Row super-tiles of launch (i_first, i_step) -- rank, world size with several GPUs.  Row I carries T - I tile pairs, so
// plain striding (I = i_first + e i_step) leaves rank 0 with (i_step - 1) / i_step of a row more than the last rank in
// EVERY block of i_step rows (8.8 % at 8 ranks, cfg 3).  Rows are dealt in UNITS of sw consecutive super-tiles (sw = the
// waves of a workgroup: the rows one workgroup sweeps in lock step; round 3 -- with single super-tiles as units the four
// waves of a shard's workgroup sat up to 8 i_step tiles apart and idled through each other's regions, so shards ran one
// wave per workgroup, each staging its own column tiles: 2.9 ms per rank at 8 ranks against 20.5 / 8 = 2.56).  The ge-th
// owned unit is taken from the ge-th block of i_step consecutive units, offset i_first in even blocks and
// i_step - 1 - i_first in odd ones: two consecutive blocks give every rank the same work.  Increasing in e (the
// column-sum slabs rely on it).  sw = 1 or i_step = 1: the plain maps.
__device__ __forceinline__ int sym_unit_of(int ge, int i_first, int i_step)
{
  return ge * i_step + ((ge & 1) ? i_step - 1 - i_first : i_first);
}
__device__ __forceinline__ int sym_row_of(int e, int i_first, int i_step, int sw)
{
  const int ge = e / sw;
  return sym_unit_of(ge, i_first, i_step) * sw + (e - ge * sw);
}
__device__ __forceinline__ bool sym_unit_owned(int G, int i_first, int i_step) { return sym_unit_of(G / i_step, i_first, i_step) == G; }
__device__ __forceinline__ bool sym_row_owned(int I, int i_first, int i_step, int sw) { return sym_unit_owned(I / sw, i_first, i_step); }
__device__ __forceinline__ int sym_rows_below(int Ilim, int i_first, int i_step, int sw)   // owned rows I < Ilim
{
  const int Gl = Ilim / sw, rem = Ilim - Gl * sw;                     // units 0 .. Gl-1 lie wholly below Ilim
  const int eb = Gl / i_step;                                         // blocks 0 .. eb-1 lie wholly below unit Gl
  int n = (eb + (sym_unit_of(eb, i_first, i_step) < Gl ? 1 : 0)) * sw;
  if (rem > 0 && sym_unit_owned(Gl, i_first, i_step)) n += rem;
  return n;
}

// Bounding box of every 64-blob tile in radius-scaled coordinates: bbox[tile] = {min x,y,z, max x,y,z}.
// Lets the symmetric kernel prove "no pair of this tile pair is closer than 2a" and run the sweep without
// the per-pair overlap test (blobs of different bodies are never that close).
__global__ __launch_bounds__(TS) void k_tile_bbox(const double *__restrict__ r, long N, double inv_a,
                                                  double *__restrict__ bbox)
{
  const long idx = (long)blockIdx.x * TS + threadIdx.x;
  double lo[3], hi[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double v = (idx < N) ? r[3 * idx + d] * inv_a : 0.0;
    lo[d] = (idx < N) ? v : 1.0e300;
    hi[d] = (idx < N) ? v : -1.0e300;
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      lo[d] = fmin(lo[d], __shfl_xor(lo[d], m, TS));
      hi[d] = fmax(hi[d], __shfl_xor(hi[d], m, TS));
    }
  if (threadIdx.x == 0) {
    double *b = bbox + 6 * (size_t)blockIdx.x;
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = hi[0]; b[4] = hi[1]; b[5] = hi[2];
  }
}

// farmap[S][J] = 1 when every blob of tile J is farther than 2a from every blob of row super-tile S
// (NI consecutive tiles): one byte per (super-tile, tile), read as a wave-uniform value by the matvec kernel.
__global__ __launch_bounds__(256) void k_tile_far(const double *__restrict__ bbox, int T, int NI,
                                                  unsigned char *__restrict__ farmap, unsigned *queue, double gap_ratio)
{
  if (queue && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *queue = 0u;   // work queue of the pair kernel that follows
  // bit 0: every blob of column tile J is farther than 2a from every row of super-tile S (no pair can overlap);
  // bit 1: ... and the relaxed (single-precision) sweep of the pair is accurate to ~1e-6 of every separation.  That sweep
  //        takes coordinates relative to the first blob of tile J: |x_j - o| <= d_J (diagonal of J's box), |x_i - o| <= gap +
  //        d_I + d_J, so rounding them to fp32 moves a separation by at most eps/2 (gap + d_I + 2 d_J), i.e. by
  //        eps/2 (1 + (d_I + 2 d_J) / gap) of itself (eps = 1.2e-7): bounded by 6e-8 (1 + gap_ratio) when d_I + 2 d_J <= gap_ratio gap
  //        (RBL_OPT_RELAXED_GAP_RATIO; the pair entries amplify a separation error ~3x: 1/r and 1/r^3 terms).  Tiles
  //        that straddle two far-apart bodies, or suspensions thousands of radii wide, fail the test for the pairs where
  //        it matters and those are swept in fp64.
  const int J = blockIdx.x * 256 + threadIdx.x, S = blockIdx.y;
  if (J >= T) return;
  double gap2 = 1.0e300;
  const double *bj = bbox + 6 * (size_t)J;
  double lo[3] = {1.0e300, 1.0e300, 1.0e300}, hi[3] = {-1.0e300, -1.0e300, -1.0e300};
  for (int a = 0; a < NI; ++a) {
    const int I = NI * S + a;
    if (I >= T) break;
    const double *bi = bbox + 6 * (size_t)I;
    double g2 = 0.0;
    for (int d = 0; d < 3; ++d) {
      const double g = fmax(fmax(bj[d] - bi[3 + d], bi[d] - bj[3 + d]), 0.0);
      g2 = __builtin_fma(g, g, g2);
      lo[d] = fmin(lo[d], bi[d]); hi[d] = fmax(hi[d], bi[3 + d]);
    }
    gap2 = fmin(gap2, g2);
  }
  double dI2 = 0.0, dJ2 = 0.0;
  for (int d = 0; d < 3; ++d) {
    dI2 = __builtin_fma(hi[d] - lo[d], hi[d] - lo[d], dI2);
    dJ2 = __builtin_fma(bj[3 + d] - bj[d], bj[3 + d] - bj[d], dJ2);
  }
  const bool far = gap2 > 4.0001;
  const bool f32ok = far && (sqrt(dI2) + 2.0 * sqrt(dJ2) <= gap_ratio * sqrt(gap2));
  farmap[(size_t)S * T + J] = far ? (f32ok ? 3 : 1) : 0;
}
#ifndef RBL_SYM_UNROLL
#define RBL_SYM_UNROLL 2
#endif
typedef double double2_t __attribute__((ext_vector_type(2)));

// ---- slab layout -------------------------------------------------------------------------------------------
// A workgroup of the symmetric kernels is SW wavefronts = SW consecutive owned row super-tiles ("row group" g) x one
// chunk c of C column tiles.  It leaves
//   row sums     slabI[c]  : for every row blob this chunk can see, i.e. blobs [0, HI(c)),  HI(c) = 64 min((c+1) C + NI-1, T)
//                            (a super-tile whose first tile is the chunk's last column still owns NI-1 more row tiles)
//   column sums  slabJ[g]  : ONE set per row group (the SW waves' LDS accumulators are added in wave order before the
//                            write), for the column blobs at or after the group's first row, [RJ(g), Npad)
// With one rank (i_step == 1) both are stored as triangles (tri = 1): prefix offsets in closed form,
//   offI(c) = 64 [C c (c+1) / 2 + (NI-1) c],      offJ(g) = g Npad - RJ1 g (g-1) / 2,   RJ(g) = RJ1 g,   RJ1 = 64 NI SW;
// row shards of a multi-GPU launch own every i_step-th super-tile, their slabs stay rectangular (and are 1/i_step of
// the single-rank size anyway).  nrhs vectors lie back to back inside a slab.  All in units of blobs (x 3 doubles).
// SW = 4 for large systems (two rows per lane), 1 otherwise: with few tiles the lock-step of a multi-wave workgroup and
// its coarser work units cost more than the slabs save (8 100 blobs: 0.080 ms with SW = 1, 0.108 ms with SW = 4).  The
// rows of a multi-GPU shard are dealt in units of SW consecutive super-tiles (sym_row_of), so a shard's workgroup sweeps
// four NEIGHBOURING super-tiles exactly like a single-rank one (round 2 dealt single super-tiles: waves up to 8 i_step
// tiles apart, 4.8 ms per rank at 8 ranks with SW = 4, which is why shards ran SW = 1 at 2.7-2.9 ms then).
#ifndef RBL_SYM_WAVES
#define RBL_SYM_WAVES 4
#endif
constexpr int SW_LARGE = RBL_SYM_WAVES;

struct SymLayout {
  long Npad;
  int T, NI, C, nch, rowsI, rowsG, tri, nrhs, i_first, i_step, SW;
};

__host__ __device__ __forceinline__ long sym_HI(const SymLayout &L, int c)
{
  if (!L.tri) return L.Npad;
  const long h = (long)(c + 1) * L.C + (L.NI - 1);
  return (h < L.T ? h : (long)L.T) * TS;
}
__host__ __device__ __forceinline__ long sym_offI(const SymLayout &L, int c)
{
  return L.tri ? (long)TS * ((long)L.C * ((long)c * (c + 1) / 2) + (long)(L.NI - 1) * c) : (long)c * L.Npad;
}
__host__ __device__ __forceinline__ long sym_RJ(const SymLayout &L, int g) { return L.tri ? (long)TS * L.NI * L.SW * g : 0; }
__host__ __device__ __forceinline__ long sym_offJ(const SymLayout &L, int g)
{
  return L.tri ? (long)g * L.Npad - (long)TS * L.NI * L.SW * ((long)g * (g - 1) / 2) : (long)g * L.Npad;
}
// doubles: address of (vector v, blob b) in chunk c's row-sum slab / group g's column-sum slab
__device__ __forceinline__ size_t sym_idxI(const SymLayout &L, int c, int v, long b)
{
  return ((size_t)sym_offI(L, c) * L.nrhs + (size_t)v * sym_HI(L, c) + (size_t)b) * 3;
}
__device__ __forceinline__ size_t sym_idxJ(const SymLayout &L, int g, int v, long b)
{
  const long rj = sym_RJ(L, g);
  return ((size_t)sym_offJ(L, g) * L.nrhs + (size_t)v * (L.Npad - rj) + (size_t)(b - rj)) * 3;
}

__device__ __forceinline__ double sym_first_lane(double v)      // lane 0's value in every lane (a wavefront-uniform double)
{
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Wave timeline of the symmetric kernel for tools/wave_trace.hip (never in the product build): every workgroup leaves
// {start, end} of its first wave on the 100 MHz constant clock, its HW_ID and XCC_ID in g_wave_trace[4 * workgroup].
#ifdef RBL_WAVE_TRACE
#define RBL_WT_BEGIN const unsigned long long wt_begin = wall_clock64();
#define RBL_WT_END(unit)                                                                                                  \
  if (::g_wave_trace && threadIdx.x == 0) {                                                                               \
    unsigned long long *wt = ::g_wave_trace + 4 * (size_t)(unit);                                                        \
    wt[0] = wt_begin; wt[1] = wall_clock64();                                                                             \
    wt[2] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); wt[3] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20); \
  }
#else
#define RBL_WT_BEGIN
#define RBL_WT_END(unit)
#endif

// PREC = 1 (relaxed product, two rows per lane only): tile pairs the far map proves free of overlaps AND safe for single
// precision (k_tile_far, bit 1) are swept in packed single precision (rbl_pair_sym_pk), coordinates relative to the first
// blob of the column tile (round 3; one origin per workgroup let the error grow with the extent of the suspension); their
// per-tile sums are added to the double accumulators, so single precision only ever sums 64 x NI terms.  Diagonal, near
// and unsafe tiles stay fp64.
// (the four-wave wall instance is held to 168 VGPRs -- three waves per SIMD instead of two at the 186 the compiler takes
// unasked: no spill, 2 % faster at cfg 3 on one box, A/B gpurun_out/r03k/ab_wpe3.txt; HIP's second launch bound is the
// minimum number of waves per SIMD.  The other instances are left alone: the relaxed one would spill, the small ones
// already fit four.)
template <bool WALL, int NI, int SW, int PREC>
__global__ __launch_bounds__(TS *SW, (WALL && NI == 2 && SW > 1 && PREC == 0) ? 3 : 1) void k_apply_M_sym(const double *__restrict__ r,
                                                        const double *__restrict__ F,
                                                        double *__restrict__ slabI,
                                                        double *__restrict__ slabJ, long N, SymLayout L, RblParams P,
                                                        unsigned *err, const unsigned char *__restrict__ farmap,
                                                        unsigned *queue)
{
  // A lane owns NI rows (row "super-tile" I = tiles NI*I .. NI*I+NI-1): the j data read from
  // LDS and the ds_add of M_ji F_i are shared by NI pair evaluations.  The SW waves of the workgroup own SW
  // consecutive owned super-tiles and walk the same column tiles in step: ONE staged j tile, per-wave accumulators.
  static_assert(PREC == 0 || NI == 2, "the packed single-precision sweep carries the two rows of a lane");
  __shared__ double2_t sP0[TS], sP1[TS], sP2[TS];  // (x,y) (z,fx) (fy,fz) of the j tile
  __shared__ double sU[SW][3][TS];                  // M_ji F_i sums for the j tile, one set per wave
  __shared__ float sPf[PREC ? 6 : 1][TS];           // relaxed product: the j tile in single precision, origin-relative
  const int lane = threadIdx.x & (TS - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int T = L.T, C = L.C;
  unsigned flags = 0;
  __shared__ double sO[3];                          // relaxed product: origin of the single-precision coordinates = first blob of the j tile
  // All pair arithmetic of this kernel runs in coordinates divided by the blob radius (the mobility entries
  // are functions of r/a only): Pu is the a = 1 parameter set, positions are scaled once when loaded.
  const RblParams Pu = unit_params(P);
  const RblWallK WK = rbl_wall_k_resident();
  // one work unit: row group g x chunk c (everything below runs once per unit; `return` ends the unit)
  auto sweep_unit = [&](const int c, const int g, const unsigned unit) {
  RBL_WT_BEGIN
  const int It00 = NI * sym_row_of(SW * g, L.i_first, L.i_step, SW);   // first tile of the group (wave 0's)
  if (It00 >= T) return;
  int J0 = c * C;
  const int J1 = (J0 + C < T) ? J0 + C : T;
  if (J0 < It00) J0 = It00;
  if (J0 >= J1) return;                                            // workgroup-uniform
  const int e = SW * g + wave;
  const int I = sym_row_of(e, L.i_first, L.i_step, SW);                // this wave's super-tile
  const bool wlive = e < L.rowsI && NI * I < T;
  const int It0 = wlive ? NI * I : (1 << 30);                      // a wave without rows never sweeps, only keeps step
  auto load_blob = [&](long idx, double &x, double &y, double &z, double &fx, double &fy, double &fz) {
    if (idx < N) {
      x = r[3 * idx]; y = r[3 * idx + 1]; z = r[3 * idx + 2];
      double d = 1.0;
      if (WALL) {
        if (z < 0.0) flags |= RBL_FLAG_BELOW_WALL;
        d = damp_of(P, z);
      }
      x *= P.inv_a; y *= P.inv_a; z *= P.inv_a;
      fx = d * F[3 * idx]; fy = d * F[3 * idx + 1]; fz = d * F[3 * idx + 2];
    } else {  // padding blob: zero force, far from everything (and from every other pad)
      x = 1.0e15 * (double)(2 + (idx - N)); y = 0.0; z = 1.0; fx = 0.0; fy = 0.0; fz = 0.0;
    }
  };

  double xi[NI], yi[NI], zi[NI], Fix[NI], Fiy[NI], Fiz[NI], uix[NI], uiy[NI], uiz[NI];
#pragma unroll
  for (int a = 0; a < NI; ++a) {
    load_blob(wlive ? (long)(It0 + a) * TS + lane : N + 1 + a, xi[a], yi[a], zi[a], Fix[a], Fiy[a], Fiz[a]);
    uix[a] = 0.0; uiy[a] = 0.0; uiz[a] = 0.0;
  }
  rbl_f2 Fx2 = {0, 0}, Fy2 = {0, 0}, Fz2 = {0, 0};
  if (PREC) {
    const int a1 = NI - 1;
    Fx2 = (rbl_f2){(float)Fix[0], (float)Fix[a1]};
    Fy2 = (rbl_f2){(float)Fiy[0], (float)Fiy[a1]};
    Fz2 = (rbl_f2){(float)Fiz[0], (float)Fiz[a1]};
  }

  for (int J = J0; J < J1; ++J) {
    const long j = (long)J * TS + lane;
    double xj = 0, yj = 0, zj = 0, Fjx = 0, Fjy = 0, Fjz = 0;
    if (wave == 0) load_blob(j, xj, yj, zj, Fjx, Fjy, Fjz);
    const bool sweeps = J >= It0;                                  // wave-uniform
    // wave-uniform: every blob of tile J is farther than 2a from every owned row (k_tile_far; bit 1: single precision is safe)
    const int fmap = (sweeps && farmap) ? __builtin_amdgcn_readfirstlane((int)farmap[(size_t)I * (size_t)T + J]) : 0;
    const bool far_tile = fmap != 0;
    __syncthreads();                                               // previous tile consumed, its column sums written
    if (wave == 0) {
      sP0[lane] = (double2_t){xj, yj};
      sP1[lane] = (double2_t){zj, Fjx};
      sP2[lane] = (double2_t){Fjy, Fjz};
      if (PREC) {
        // coordinates relative to the tile's first blob: what single precision then rounds is a separation-sized number
        const double ox = sym_first_lane(xj), oy = sym_first_lane(yj), oz = sym_first_lane(zj);
        if (lane == 0) { sO[0] = ox; sO[1] = oy; sO[2] = oz; }
        sPf[0][lane] = (float)(xj - ox); sPf[PREC ? 1 : 0][lane] = (float)(yj - oy); sPf[PREC ? 2 : 0][lane] = (float)(zj - oz);
        sPf[PREC ? 3 : 0][lane] = (float)Fjx; sPf[PREC ? 4 : 0][lane] = (float)Fjy; sPf[PREC ? 5 : 0][lane] = (float)Fjz;
      }
    }
    sU[wave][0][lane] = 0.0; sU[wave][1][lane] = 0.0; sU[wave][2][lane] = 0.0;
    __syncthreads();
    if (!sweeps) {
      // the tile lies before this wave's rows: pairs belong to an earlier wave
    } else if (PREC && (fmap & 2) && J >= It0 + NI) {   // relaxed product: packed single precision, both rows of the lane at once
      const double ox = sO[0], oy = sO[1], oz = sO[2];
      const int a1 = NI - 1;
      const rbl_f2 xi2 = {(float)(xi[0] - ox), (float)(xi[a1] - ox)}, yi2 = {(float)(yi[0] - oy), (float)(yi[a1] - oy)},
                   zi2 = {(float)(zi[0] - oz), (float)(zi[a1] - oz)};
      const float two_z0 = (float)(2.0 * oz);
      rbl_f2 ax = {0, 0}, ay = {0, 0}, az = {0, 0};
#pragma unroll 2
      for (int s = 0; s < TS; ++s) {
        const int jj = (lane + s) & (TS - 1);
        rbl_f2 vx = {0, 0}, vy = {0, 0}, vz = {0, 0};
        rbl_pair_sym_pk<WALL>(xi2, yi2, zi2, Fx2, Fy2, Fz2, sPf[0][jj], sPf[PREC ? 1 : 0][jj], sPf[PREC ? 2 : 0][jj],
                              sPf[PREC ? 3 : 0][jj], sPf[PREC ? 4 : 0][jj], sPf[PREC ? 5 : 0][jj], two_z0, ax, ay, az, vx, vy, vz);
        // column sums in double: ds_add_f32 issues ~23x slower than ds_add_f64 on gfx950 (tools/peak_fp32pk.hip: 3.2 vs 74 G
        // wave-instructions/s), three conversions per step are far cheaper
        __hip_atomic_fetch_add(&sU[wave][0][jj], (double)(vx.x + vx.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sU[wave][1][jj], (double)(vy.x + vy.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sU[wave][2][jj], (double)(vz.x + vz.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      uix[0] += (double)ax.x; uiy[0] += (double)ay.x; uiz[0] += (double)az.x;
      uix[NI - 1] += (double)ax.y; uiy[NI - 1] += (double)ay.y; uiz[NI - 1] += (double)az.y;
    } else if (J >= It0 + NI) {  // every owned row tile lies strictly before J: fused symmetric sweep
      auto sweep = [&](auto nearchk) {
        unsigned off16 = (unsigned)lane * 16u;       // byte offset of column jj in the 16-B arrays, carried (jj*8 = off16/2)
#pragma unroll RBL_SYM_UNROLL
        for (int s = 0; s < TS; ++s) {
          const double2_t pa = *(const double2_t *)((const char *)sP0 + off16), pb = *(const double2_t *)((const char *)sP1 + off16),
                          pd = *(const double2_t *)((const char *)sP2 + off16);
          const int jj = (int)(off16 >> 4);
          off16 = (off16 + 16u) & (unsigned)(TS * 16 - 16);
          double vx = 0.0, vy = 0.0, vz = 0.0;
#pragma unroll
          for (int a = 0; a < NI; ++a)
            rbl_pair_sym<WALL, true, decltype(nearchk)::value>(Pu, xi[a], yi[a], zi[a], Fix[a], Fiy[a], Fiz[a], pa.x,
                                                               pa.y, pb.x, pb.y, pd.x, pd.y, uix[a], uiy[a], uiz[a],
                                                               vx, vy, vz, flags, WK);
          __hip_atomic_fetch_add(&sU[wave][0][jj], vx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&sU[wave][1][jj], vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&sU[wave][2][jj], vz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
      if (far_tile) sweep(std::false_type{});   // no pair can overlap: sweep without the per-pair test
      else sweep(std::true_type{});
    } else {              // J is one of the owned row tiles: per sub-tile diagonal / symmetric / skip
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        if (J == It0 + a) {
#pragma unroll 4
          for (int jj = 0; jj < TS; ++jj) {
            const double2_t pa = sP0[jj], pb = sP1[jj], pd = sP2[jj];
            rbl_pair_accum<WALL, true, true>(Pu, xi[a], yi[a], zi[a], pa.x, pa.y, pb.x, pb.y, pd.x, pd.y, jj == lane,
                                       uix[a], uiy[a], uiz[a], flags);
          }
        } else if (J > It0 + a) {
          for (int s = 0; s < TS; ++s) {
            const int jj = (lane + s) & (TS - 1);
            const double2_t pa = sP0[jj], pb = sP1[jj], pd = sP2[jj];
            double vx = 0.0, vy = 0.0, vz = 0.0;
            rbl_pair_sym<WALL, true>(Pu, xi[a], yi[a], zi[a], Fix[a], Fiy[a], Fiz[a], pa.x, pa.y, pb.x, pb.y, pd.x,
                               pd.y, uix[a], uiy[a], uiz[a], vx, vy, vz, flags, WK);
            __hip_atomic_fetch_add(&sU[wave][0][jj], vx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&sU[wave][1][jj], vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&sU[wave][2][jj], vz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    if (J > It00) {  // some owned row tile of the group precedes J: column sums exist.  Waves added in fixed order.
      __syncthreads();
      for (int q = threadIdx.x; q < 3 * TS; q += TS * SW) {
        const int l = q / 3, k = q - 3 * l;
        double sum = sU[0][k][l];
#pragma unroll
        for (int w = 1; w < SW; ++w) sum += sU[w][k][l];
        slabJ[sym_idxJ(L, g, 0, (long)J * TS + l) + k] = sum;
      }
    }
  }
  if (wlive && c * C + C > It0) {   // this wave swept at least one tile of the chunk: its row sums (k_reduce_sym reads chunks >= It0 / C)
#pragma unroll
    for (int a = 0; a < NI; ++a) {
      if (It0 + a < T) {
        double *p = slabI + sym_idxI(L, c, 0, (long)(It0 + a) * TS + lane);
        p[0] = uix[a]; p[1] = uiy[a]; p[2] = uiz[a];
      }
    }
  }
  RBL_WT_END(unit)
  };
  if (!queue) {
    // one unit per workgroup.  Workgroups go to the 8 XCDs round-robin in launch order: rotate the row group with the chunk,
    // or a group count that is a multiple of 8 pins every group (and its triangular share of the work) to one XCD
    sweep_unit((int)blockIdx.y, (int)((blockIdx.x + blockIdx.y) % gridDim.x), blockIdx.y * gridDim.x + blockIdx.x);
  } else {
    // WORK QUEUE (large systems): a fixed set of resident workgroups draws units from one counter until it runs dry, so an
    // XCD that runs faster -- they differ by up to 5 % under fp64 load, and launch-order dispatch deals every XCD the same
    // number of workgroups -- simply takes more units, and nobody idles for longer than one unit at the end.  The slabs are
    // addressed by unit, not by workgroup: results do not depend on who swept what.  Long chunks first.
    __shared__ unsigned s_unit;
    const unsigned n_units = (unsigned)L.rowsG * (unsigned)L.nch;
    for (;;) {
      __syncthreads();                                   // the previous unit's LDS is dead
      if (threadIdx.x == 0) s_unit = atomicAdd(queue, 1u);
      __syncthreads();
      const unsigned u = s_unit;
      if (u >= n_units) break;
      const int by = (int)(u / (unsigned)L.rowsG), bx = (int)(u - (unsigned)by * (unsigned)L.rowsG);
      sweep_unit(L.nch - 1 - by, (bx + by) % L.rowsG, u);
    }
  }
  if (flags) atomicOr(err, flags);
}

// ---------------------------------------------------------------------------
// Mid-size systems (one row per lane: fewer than 128 blob tiles per rank -- BASELINE configs[1], 8 100 blobs): the same
// work decomposition and the SAME slabs as k_apply_M_sym<WALL, 1, 1, 0>, re-shaped for what limits that size (round 4):
//   * a work unit belongs to ONE WAVE, and IW independent waves share a workgroup (own LDS image each, no workgroup
//     barrier -- a wave's DS instructions execute in order).  Single-wave workgroups stop at 16 per CU = 4 waves per SIMD
//     (profiles/r03_cfg2_wave_timeline.md); with four waves per workgroup eight waves per SIMD are resident, which is what
//     hides the staging loads and the fp64 latency chain of a unit that lasts only 5 us of issue;
//   * the column sums M_ji F_i stay in VGPRs and ROTATE with the column index: at step s lane l pairs its row with column
//     (l + s) & 63, so the accumulator of that column moves one lane down per step -- `v_mov_b32_dpp wave_rol:1`, six dword
//     moves, folded into the pair's own FMA chain (the rotated sum is the addend).  The three ds_add_f64 per step of the
//     LDS form (24 of its 47 LDS clocks, as much LDS time as VALU time at one row per lane) are gone, and after 64 steps
//     every accumulator is back in its own lane: the 64 sums go to the slab straight from registers;
//   * only the existing units are launched (closed-form index -> (row tile, chunk)), never the dead half of the rectangle.
// Measured at cfg 2 (tools/bench_midsize.py, one box, alternating): 70.1 us per product against 73.9-75.4 (-6 %); PMC passes of both
// kernels in profiles/r04_cfg2_kernel_pmc.md.
// Sums per column are formed in step order, like the LDS atomics before them: bitwise reproducible run to run.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_rol1(double v)            // lane l takes lane (l + 1) & 63's value
{
  // (mov_dpp: no `old` operand to initialise -- every lane of a wave rotate has a source; update_dpp(0, ...) cost a v_mov per half)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x134, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x134, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// live units of row tile e (i_step == 1, NI == 1): chunks e / C .. nch - 1.  units before row e:
__host__ __device__ __forceinline__ long symw_prefix(int e, int C, int nch)
{
  const long q = e / C, rem = e - q * C;
  return (long)e * nch - ((long)C * (q * (q - 1) / 2) + q * rem);
}

#ifndef RBL_SYMW_WAVES_FREE
#define RBL_SYMW_WAVES_FREE 6     // waves per SIMD the register allocator is held to (HIP's second launch bound): 80 VGPRs, no spill
#endif
#ifndef RBL_SYMW_WAVES_WALL
#define RBL_SYMW_WAVES_WALL 4     // 116 VGPRs (80 would spill 136 bytes per lane)
#endif
#ifndef RBL_SYMW_UNROLL
#define RBL_SYMW_UNROLL 2
#endif
#ifndef RBL_SYMW_LDSACC
#define RBL_SYMW_LDSACC 0         // column-sum components kept in LDS (atomics) instead of rotating registers
#endif
#ifndef RBL_SYMW_IW
#define RBL_SYMW_IW 4             // independent waves (= work units in flight) per workgroup
#endif
#ifndef RBL_SYMW2_WAVES_FREE
#define RBL_SYMW2_WAVES_FREE 4    // two rows per lane: waves per SIMD the allocator is held to (free space / wall)
#endif
#ifndef RBL_SYMW2_WAVES_WALL
#define RBL_SYMW2_WAVES_WALL 3
#endif
// live units of row super-tile e with NI rows per lane: chunks (NI e) / C .. nch - 1.  Closed form for NI = 1, and for NI = 2 with
// C = 1 or C even (floor(2 e / C) = floor(e / (C / 2)); sym_geometry keeps C that way for these kernels)
__host__ __device__ __forceinline__ long symw_prefix_ni(int e, int C, int nch, int ni)
{
  if (ni == 1) return symw_prefix(e, C, nch);
  if (C == 1) return (long)e * nch - (long)e * (e - 1);
  return symw_prefix(e, C / 2, nch);
}
// NI = 2 (round 4, 8 200 - 20 480 blobs and whatever sym_geometry sends here): a lane owns the same two row tiles as in
// k_apply_M_sym<WALL, 2, 1, 0> and the SAME slabs come out; the column data read from LDS and the three travelling column sums
// (six DPP moves per step) are shared by the two pair evaluations of a step.
template <bool WALL, int NI, int IW>
__global__ __launch_bounds__(TS *IW, NI == 2 ? (WALL ? RBL_SYMW2_WAVES_WALL : RBL_SYMW2_WAVES_FREE) : (WALL ? RBL_SYMW_WAVES_WALL : RBL_SYMW_WAVES_FREE))
void k_apply_M_symw(const double *__restrict__ r, const double *__restrict__ F,
                                                        double *__restrict__ slabI, double *__restrict__ slabJ, long N,
                                                        SymLayout L, RblParams P, unsigned *err, long n_units)
{
  __shared__ double2_t sP0[IW][TS], sP1[IW][TS], sP2[IW][TS];   // (x,y) (z,fx) (fy,fz) of a wave's current column tile
  // RBL_SYMW_LDSACC of the three column-sum components go through LDS atomics instead of the rotating registers: the DPP
  // moves are VALU work, the atomics LDS work, and the kernel is bound by whichever pipe carries more (measured)
  constexpr int NL = RBL_SYMW_LDSACC;
  __shared__ double sU[IW][NL > 0 ? NL : 1][TS];
  const int lane = threadIdx.x & (TS - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int T = L.T, C = L.C;
  unsigned flags = 0;
  const RblParams Pu = unit_params(P);
  const RblWallK WK = rbl_wall_k_resident();
  // one work unit of this wave (`return` ends the unit)
  auto sweep_unit = [&](const long u) {
  int e, c;
  if (L.tri) {                       // u -> (row super-tile, chunk) through the closed-form prefix count (scalar work)
    int lo = 0, hi = L.rowsI;        // largest e with prefix(e) <= u
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (symw_prefix_ni(mid, C, L.nch, NI) <= u) lo = mid; else hi = mid;
    }
    e = lo; c = (NI * e) / C + (int)(u - symw_prefix_ni(e, C, L.nch, NI));
  } else {                           // a multi-GPU shard: the rectangle (row, chunk), dead units leave at once
    c = (int)(u / L.rowsI);
    e = (int)((u - (long)c * L.rowsI + c) % L.rowsI);
  }
  const int I = sym_row_of(e, L.i_first, L.i_step, 1);
  const int It0 = NI * I;            // first row tile of the lane's NI
  if (It0 >= T) return;
  int J0 = c * C;
  const int J1 = (J0 + C < T) ? J0 + C : T;
  if (J0 < It0) J0 = It0;
  if (J0 >= J1) return;
  auto load_blob = [&](long idx, double &x, double &y, double &z, double &fx, double &fy, double &fz) {
    if (idx < N) {
      x = r[3 * idx]; y = r[3 * idx + 1]; z = r[3 * idx + 2];
      double d = 1.0;
      if (WALL) {
        if (z < 0.0) flags |= RBL_FLAG_BELOW_WALL;
        d = damp_of(P, z);
      }
      x *= P.inv_a; y *= P.inv_a; z *= P.inv_a;
      fx = d * F[3 * idx]; fy = d * F[3 * idx + 1]; fz = d * F[3 * idx + 2];
    } else {  // padding blob: zero force, far from everything (and from every other pad)
      x = 1.0e15 * (double)(2 + (idx - N)); y = 0.0; z = 1.0; fx = 0.0; fy = 0.0; fz = 0.0;
    }
  };
  double xi[NI], yi[NI], zi[NI], Fix[NI], Fiy[NI], Fiz[NI], uix[NI], uiy[NI], uiz[NI];
#pragma unroll
  for (int a = 0; a < NI; ++a) {
    load_blob((long)(It0 + a) * TS + lane, xi[a], yi[a], zi[a], Fix[a], Fiy[a], Fiz[a]);   // (a row tile beyond T is padding: zero force)
    uix[a] = 0.0; uiy[a] = 0.0; uiz[a] = 0.0;
  }
  for (int J = J0; J < J1; ++J) {
    {
      double xj, yj, zj, Fjx, Fjy, Fjz;
      load_blob((long)J * TS + lane, xj, yj, zj, Fjx, Fjy, Fjz);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // the previous tile's reads are done (in-order DS queue)
      sP0[wave][lane] = (double2_t){xj, yj};
      sP1[wave][lane] = (double2_t){zj, Fjx};
      sP2[wave][lane] = (double2_t){Fjy, Fjz};
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    double ax = 0.0, ay = 0.0, az = 0.0;                            // column sums of column (lane + s) & 63, travelling
    if (J < It0 + NI) {              // J is one of the lane's own row tiles: the ordered diagonal sweep of that row tile ...
#pragma unroll
      for (int a = 0; a < NI; ++a)
        if (J == It0 + a) {
#pragma unroll 2
          for (int jj = 0; jj < TS; ++jj) {
            const double2_t pa = sP0[wave][jj], pb = sP1[wave][jj], pd = sP2[wave][jj];
            rbl_pair_accum<WALL, true, true>(Pu, xi[a], yi[a], zi[a], pa.x, pa.y, pb.x, pb.y, pd.x, pd.y, jj == lane, uix[a], uiy[a], uiz[a], flags);
          }
        }
      if (NI == 1 || J == It0) continue;                             // (no earlier row tile of the lane: no column sums)
      // ... and (NI = 2, J = It0 + 1) the symmetric sweep of row tile It0 against it
      unsigned off16 = (unsigned)lane * 16u;
      const char *b0 = (const char *)sP0[wave], *b1 = (const char *)sP1[wave], *b2 = (const char *)sP2[wave];
#pragma unroll 2
      for (int s = 0; s < TS; ++s) {
        const double2_t pa = *(const double2_t *)(b0 + off16), pb = *(const double2_t *)(b1 + off16), pd = *(const double2_t *)(b2 + off16);
        off16 = (off16 + 16u) & (unsigned)(TS * 16 - 16);
        double vx = ax, vy = ay, vz = az;
        rbl_pair_sym<WALL, true, true>(Pu, xi[0], yi[0], zi[0], Fix[0], Fiy[0], Fiz[0], pa.x, pa.y, pb.x, pb.y, pd.x, pd.y, uix[0], uiy[0], uiz[0], vx, vy, vz, flags, WK);
        ax = wave_rol1(vx); ay = wave_rol1(vy); az = wave_rol1(vz);
      }
      double *q = slabJ + sym_idxJ(L, e, 0, (long)J * TS + lane);
      q[0] = ax; q[1] = ay; q[2] = az;
      continue;
    }
    if (NL > 0) {
#pragma unroll
      for (int k = 0; k < NL; ++k) sU[wave][k][lane] = 0.0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    unsigned off16 = (unsigned)lane * 16u;
    const char *b0 = (const char *)sP0[wave], *b1 = (const char *)sP1[wave], *b2 = (const char *)sP2[wave];
#pragma unroll RBL_SYMW_UNROLL
    for (int s = 0; s < TS; ++s) {
      const double2_t pa = *(const double2_t *)(b0 + off16), pb = *(const double2_t *)(b1 + off16), pd = *(const double2_t *)(b2 + off16);
      const int jj = (int)(off16 >> 4);
      off16 = (off16 + 16u) & (unsigned)(TS * 16 - 16);
      double vx = (NL > 0) ? 0.0 : ax, vy = (NL > 1) ? 0.0 : ay, vz = (NL > 2) ? 0.0 : az;
#pragma unroll
      for (int a = 0; a < NI; ++a)
        rbl_pair_sym<WALL, true, true>(Pu, xi[a], yi[a], zi[a], Fix[a], Fiy[a], Fiz[a], pa.x, pa.y, pb.x, pb.y, pd.x, pd.y, uix[a], uiy[a], uiz[a], vx, vy, vz, flags, WK);
      if (NL > 0) __hip_atomic_fetch_add(&sU[wave][0][jj], vx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); else ax = wave_rol1(vx);
      if (NL > 1) __hip_atomic_fetch_add(&sU[wave][NL > 1 ? 1 : 0][jj], vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); else ay = wave_rol1(vy);
      if (NL > 2) __hip_atomic_fetch_add(&sU[wave][NL > 2 ? 2 : 0][jj], vz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); else az = wave_rol1(vz);
    }
    if (NL > 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      ax = sU[wave][0][lane];
      if (NL > 1) ay = sU[wave][NL > 1 ? 1 : 0][lane];
      if (NL > 2) az = sU[wave][NL > 2 ? 2 : 0][lane];
    }
    double *q = slabJ + sym_idxJ(L, e, 0, (long)J * TS + lane);     // 64 rotations: lane l holds column l again
    q[0] = ax; q[1] = ay; q[2] = az;
  }
#pragma unroll
  for (int a = 0; a < NI; ++a)
    if (It0 + a < T) {
      double *p_ = slabI + sym_idxI(L, c, 0, (long)(It0 + a) * TS + lane);
      p_[0] = uix[a]; p_[1] = uiy[a]; p_[2] = uiz[a];
    }
  };
  // (A work queue -- a resident set of waves drawing units from one counter, or from eight per-XCD counters -- was measured and
  // dropped: 179 and 119 us per product at cfg 2 against 70 with one unit per wave, gpurun_out/r04f, r04g: a wave that sweeps
  // units back to back pays every unit's staging latency in series.)
  const long u = (long)blockIdx.x * IW + wave;
  if (u < n_units) sweep_unit(u);
  if (flags) atomicOr(err, flags);
}

// ---------------------------------------------------------------------------
// The wave-unit kernel for TWO force vectors (the Lanczos pair of mid-size systems, end of round 4): k_apply_M_sym2<WALL, NI, 1>'s
// decomposition and slabs, k_apply_M_symw's execution -- a unit per wave, the SIX column sums of a step (three per vector) rotating
// through the lanes in registers instead of six ds_add_f64 (the one-row form of the round-3 kernel is LDS-bound on them: 11 LDS
// instructions per pair step).  Vector v of F / the slabs as in k_apply_M_sym2.
// ---------------------------------------------------------------------------
#ifndef RBL_SYMW2V_WAVES
#define RBL_SYMW2V_WAVES(WALL, NI) ((NI) == 2 ? ((WALL) ? 2 : 3) : ((WALL) ? 3 : 4))
#endif
template <bool WALL, int NI, int IW>
__global__ __launch_bounds__(TS *IW, RBL_SYMW2V_WAVES(WALL, NI))
void k_apply_M_symw2v(const double *__restrict__ r, const double *__restrict__ F, double *__restrict__ slabI, double *__restrict__ slabJ,
                      long N, SymLayout L, RblParams P, unsigned *err, long n_units)
{
  __shared__ double2_t sP0[IW][TS], sP1[IW][TS], sP2[IW][TS], sP3[IW][TS], sP4[IW][TS];   // (x,y) (z,f0x) (f0y,f0z) (f1x,f1y) (f1z,-)
  const int lane = threadIdx.x & (TS - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int T = L.T, C = L.C;
  const long n3 = 3 * N;
  unsigned flags = 0;
  const RblParams Pu = unit_params(P);
  const RblWallK WK = rbl_wall_k_resident();
  auto sweep_unit = [&](const long u) {
  int e, c;
  if (L.tri) {
    int lo = 0, hi = L.rowsI;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (symw_prefix_ni(mid, C, L.nch, NI) <= u) lo = mid; else hi = mid;
    }
    e = lo; c = (NI * e) / C + (int)(u - symw_prefix_ni(e, C, L.nch, NI));
  } else {
    c = (int)(u / L.rowsI);
    e = (int)((u - (long)c * L.rowsI + c) % L.rowsI);
  }
  const int I = sym_row_of(e, L.i_first, L.i_step, 1);
  const int It0 = NI * I;
  if (It0 >= T) return;
  int J0 = c * C;
  const int J1 = (J0 + C < T) ? J0 + C : T;
  if (J0 < It0) J0 = It0;
  if (J0 >= J1) return;
  auto load_blob = [&](long idx, double &x, double &y, double &z, RblV3 &f0, RblV3 &f1) {
    if (idx < N) {
      x = r[3 * idx]; y = r[3 * idx + 1]; z = r[3 * idx + 2];
      double d = 1.0;
      if (WALL) {
        if (z < 0.0) flags |= RBL_FLAG_BELOW_WALL;
        d = damp_of(P, z);
      }
      x *= P.inv_a; y *= P.inv_a; z *= P.inv_a;
      f0 = RblV3{d * F[3 * idx], d * F[3 * idx + 1], d * F[3 * idx + 2]};
      f1 = RblV3{d * F[n3 + 3 * idx], d * F[n3 + 3 * idx + 1], d * F[n3 + 3 * idx + 2]};
    } else {
      x = 1.0e15 * (double)(2 + (idx - N)); y = 0.0; z = 1.0; f0 = RblV3{0.0, 0.0, 0.0}; f1 = RblV3{0.0, 0.0, 0.0};
    }
  };
  double xi[NI], yi[NI], zi[NI];
  RblV3 Fi0[NI], Fi1[NI], ui0[NI], ui1[NI];
#pragma unroll
  for (int a = 0; a < NI; ++a) {
    load_blob((long)(It0 + a) * TS + lane, xi[a], yi[a], zi[a], Fi0[a], Fi1[a]);
    ui0[a] = RblV3{0.0, 0.0, 0.0}; ui1[a] = RblV3{0.0, 0.0, 0.0};
  }
  for (int J = J0; J < J1; ++J) {
    {
      double xj, yj, zj; RblV3 g0, g1;
      load_blob((long)J * TS + lane, xj, yj, zj, g0, g1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      sP0[wave][lane] = (double2_t){xj, yj};
      sP1[wave][lane] = (double2_t){zj, g0.x};
      sP2[wave][lane] = (double2_t){g0.y, g0.z};
      sP3[wave][lane] = (double2_t){g1.x, g1.y};
      sP4[wave][lane] = (double2_t){g1.z, 0.0};
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    RblV3 a0{0.0, 0.0, 0.0}, a1{0.0, 0.0, 0.0};                    // the six travelling column sums
    const char *b0 = (const char *)sP0[wave], *b1 = (const char *)sP1[wave], *b2 = (const char *)sP2[wave],
               *b3 = (const char *)sP3[wave], *b4 = (const char *)sP4[wave];
    auto sym_sweep = [&](const int alo, const int ahi) {           // rows alo .. ahi - 1 of the lane against tile J, column sums rotating
      unsigned off16 = (unsigned)lane * 16u;
#pragma unroll 2
      for (int s = 0; s < TS; ++s) {
        const double2_t pa = *(const double2_t *)(b0 + off16), pb = *(const double2_t *)(b1 + off16), pd = *(const double2_t *)(b2 + off16),
                        pe = *(const double2_t *)(b3 + off16), pf = *(const double2_t *)(b4 + off16);
        off16 = (off16 + 16u) & (unsigned)(TS * 16 - 16);
        RblV3 v0 = a0, v1 = a1;
#pragma unroll
        for (int a = 0; a < NI; ++a)
          if (a >= alo && a < ahi)
            rbl_pair_sym2<WALL, true, true>(Pu, xi[a], yi[a], zi[a], Fi0[a], Fi1[a], pa.x, pa.y, pb.x, RblV3{pb.y, pd.x, pd.y},
                                            RblV3{pe.x, pe.y, pf.x}, ui0[a], ui1[a], v0, v1, flags, WK);
        a0 = RblV3{wave_rol1(v0.x), wave_rol1(v0.y), wave_rol1(v0.z)};
        a1 = RblV3{wave_rol1(v1.x), wave_rol1(v1.y), wave_rol1(v1.z)};
      }
    };
    if (J < It0 + NI) {              // one of the lane's own row tiles: its ordered diagonal sweep, one vector after the other ...
#pragma unroll
      for (int a = 0; a < NI; ++a)
        if (J == It0 + a) {
#pragma unroll 2
          for (int jj = 0; jj < TS; ++jj) {
            const double2_t pa = sP0[wave][jj], pb = sP1[wave][jj], pd = sP2[wave][jj], pe = sP3[wave][jj], pf = sP4[wave][jj];
            rbl_pair_accum<WALL, true, true>(Pu, xi[a], yi[a], zi[a], pa.x, pa.y, pb.x, pb.y, pd.x, pd.y, jj == lane,
                                             ui0[a].x, ui0[a].y, ui0[a].z, flags);
            rbl_pair_accum<WALL, true, true>(Pu, xi[a], yi[a], zi[a], pa.x, pa.y, pb.x, pe.x, pe.y, pf.x, jj == lane,
                                             ui1[a].x, ui1[a].y, ui1[a].z, flags);
          }
        }
      if (NI == 1 || J == It0) continue;
      sym_sweep(0, 1);               // ... and (NI = 2, J = It0 + 1) row tile It0 against it
    } else
      sym_sweep(0, NI);
    double *q0 = slabJ + sym_idxJ(L, e, 0, (long)J * TS + lane), *q1 = slabJ + sym_idxJ(L, e, 1, (long)J * TS + lane);
    q0[0] = a0.x; q0[1] = a0.y; q0[2] = a0.z;
    q1[0] = a1.x; q1[1] = a1.y; q1[2] = a1.z;
  }
#pragma unroll
  for (int a = 0; a < NI; ++a)
    if (It0 + a < T) {
      double *p0 = slabI + sym_idxI(L, c, 0, (long)(It0 + a) * TS + lane), *p1 = slabI + sym_idxI(L, c, 1, (long)(It0 + a) * TS + lane);
      p0[0] = ui0[a].x; p0[1] = ui0[a].y; p0[2] = ui0[a].z;
      p1[0] = ui1[a].x; p1[1] = ui1[a].y; p1[2] = ui1[a].z;
    }
  };
  const long u = (long)blockIdx.x * IW + wave;
  if (u < n_units) sweep_unit(u);
  if (flags) atomicOr(err, flags);
}

// ---------------------------------------------------------------------------
// The symmetric product for TWO force vectors at once (F, out: [2][3N]): same work decomposition, the pair
// coefficients are evaluated once for both (rbl_pair_sym2).  Slabs hold the two vectors back to back.
// ---------------------------------------------------------------------------
#ifndef RBL_SYM2_MIN_WAVES
#define RBL_SYM2_MIN_WAVES 1      // waves per SIMD the two-vector wall instance is held to.  Measured with 3 (tools/build_variant.sh s2w3
                                  // -DRBL_SYM2_MIN_WAVES=3: 168 VGPRs + 232 B of scratch instead of 235 VGPRs): Brownian step 608.6-609.1 ms
                                  // against 608.9-609.3 -- no change, left at the compiler's choice
#endif
template <bool WALL, int NI, int SW, int PREC>
__global__ __launch_bounds__(TS *SW, (WALL && NI == 2 && SW > 1 && PREC == 0) ? RBL_SYM2_MIN_WAVES : 1) void k_apply_M_sym2(const double *__restrict__ r, const double *__restrict__ F,
                                                         double *__restrict__ slabI, double *__restrict__ slabJ, long N,
                                                         SymLayout L, RblParams P, unsigned *err,
                                                         const unsigned char *__restrict__ farmap, unsigned *queue)
{
  static_assert(PREC == 0 || NI == 2, "the packed single-precision sweep carries the two rows of a lane");
  __shared__ double2_t sP0[TS], sP1[TS], sP2[TS], sP3[TS], sP4[TS];  // (x,y) (z,f0x) (f0y,f0z) (f1x,f1y) (f1z,-)
  __shared__ double sU[SW][2][3][TS];
  __shared__ float sPf[PREC ? 9 : 1][TS];           // relaxed product: x y z f0 f1 in single precision, origin-relative
  const int lane = threadIdx.x & (TS - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int T = L.T, C = L.C;
  const long n3 = 3 * N;
  unsigned flags = 0;
  const RblParams Pu = unit_params(P);
  const RblWallK WK = rbl_wall_k_resident();
  __shared__ double sO[3];                          // relaxed product: origin = first blob of the j tile (see k_apply_M_sym)
  auto sweep_unit = [&](const int c, const int g) {     // one work unit (see k_apply_M_sym)
  const int It00 = NI * sym_row_of(SW * g, L.i_first, L.i_step, SW);
  if (It00 >= T) return;
  int J0 = c * C;
  const int J1 = (J0 + C < T) ? J0 + C : T;
  if (J0 < It00) J0 = It00;
  if (J0 >= J1) return;
  const int e = SW * g + wave;
  const int I = sym_row_of(e, L.i_first, L.i_step, SW);
  const bool wlive = e < L.rowsI && NI * I < T;
  const int It0 = wlive ? NI * I : (1 << 30);
  auto load_blob = [&](long idx, double &x, double &y, double &z, RblV3 &f0, RblV3 &f1) {
    if (idx < N) {
      x = r[3 * idx]; y = r[3 * idx + 1]; z = r[3 * idx + 2];
      double d = 1.0;
      if (WALL) {
        if (z < 0.0) flags |= RBL_FLAG_BELOW_WALL;
        d = damp_of(P, z);
      }
      x *= P.inv_a; y *= P.inv_a; z *= P.inv_a;
      f0 = RblV3{d * F[3 * idx], d * F[3 * idx + 1], d * F[3 * idx + 2]};
      f1 = RblV3{d * F[n3 + 3 * idx], d * F[n3 + 3 * idx + 1], d * F[n3 + 3 * idx + 2]};
    } else {
      x = 1.0e15 * (double)(2 + (idx - N)); y = 0.0; z = 1.0;
      f0 = RblV3{0.0, 0.0, 0.0}; f1 = f0;
    }
  };
  double xi[NI], yi[NI], zi[NI];
  RblV3 Fi0[NI], Fi1[NI], ui0[NI], ui1[NI];
#pragma unroll
  for (int a = 0; a < NI; ++a) {
    load_blob(wlive ? (long)(It0 + a) * TS + lane : N + 1 + a, xi[a], yi[a], zi[a], Fi0[a], Fi1[a]);
    ui0[a] = RblV3{0.0, 0.0, 0.0}; ui1[a] = ui0[a];
  }
  rbl_f2 F0x = {0, 0}, F0y = {0, 0}, F0z = {0, 0}, F1x = {0, 0}, F1y = {0, 0}, F1z = {0, 0};
  if (PREC) {
    const int a1 = NI - 1;
    F0x = (rbl_f2){(float)Fi0[0].x, (float)Fi0[a1].x}; F0y = (rbl_f2){(float)Fi0[0].y, (float)Fi0[a1].y}; F0z = (rbl_f2){(float)Fi0[0].z, (float)Fi0[a1].z};
    F1x = (rbl_f2){(float)Fi1[0].x, (float)Fi1[a1].x}; F1y = (rbl_f2){(float)Fi1[0].y, (float)Fi1[a1].y}; F1z = (rbl_f2){(float)Fi1[0].z, (float)Fi1[a1].z};
  }
  for (int J = J0; J < J1; ++J) {
    const long j = (long)J * TS + lane;
    double xj = 0, yj = 0, zj = 0;
    RblV3 Fj0{0, 0, 0}, Fj1{0, 0, 0};
    if (wave == 0) load_blob(j, xj, yj, zj, Fj0, Fj1);
    const bool sweeps = J >= It0;
    const int fmap = (sweeps && farmap) ? __builtin_amdgcn_readfirstlane((int)farmap[(size_t)I * (size_t)T + J]) : 0;
    const bool far_tile = fmap != 0;
    __syncthreads();
    double ox = 0.0, oy = 0.0, oz = 0.0;
    if (PREC && wave == 0) {
      ox = sym_first_lane(xj); oy = sym_first_lane(yj); oz = sym_first_lane(zj);
      if (lane == 0) { sO[0] = ox; sO[1] = oy; sO[2] = oz; }
    }
    if (wave == 0) {
      sP0[lane] = (double2_t){xj, yj};
      sP1[lane] = (double2_t){zj, Fj0.x};
      sP2[lane] = (double2_t){Fj0.y, Fj0.z};
      sP3[lane] = (double2_t){Fj1.x, Fj1.y};
      sP4[lane] = (double2_t){Fj1.z, 0.0};
      if (PREC) {
        sPf[0][lane] = (float)(xj - ox); sPf[PREC ? 1 : 0][lane] = (float)(yj - oy); sPf[PREC ? 2 : 0][lane] = (float)(zj - oz);
        sPf[PREC ? 3 : 0][lane] = (float)Fj0.x; sPf[PREC ? 4 : 0][lane] = (float)Fj0.y; sPf[PREC ? 5 : 0][lane] = (float)Fj0.z;
        sPf[PREC ? 6 : 0][lane] = (float)Fj1.x; sPf[PREC ? 7 : 0][lane] = (float)Fj1.y; sPf[PREC ? 8 : 0][lane] = (float)Fj1.z;
      }
    }
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int d = 0; d < 3; ++d) sU[wave][v][d][lane] = 0.0;
    __syncthreads();
    auto pair_step = [&](int jj, int a, auto nearchk) {
      const double2_t pa = sP0[jj], pb = sP1[jj], pd = sP2[jj], pe = sP3[jj], pf = sP4[jj];
      RblV3 v0{0.0, 0.0, 0.0}, v1{0.0, 0.0, 0.0};
      rbl_pair_sym2<WALL, true, decltype(nearchk)::value>(Pu, xi[a], yi[a], zi[a], Fi0[a], Fi1[a], pa.x, pa.y, pb.x,
                                                          RblV3{pb.y, pd.x, pd.y}, RblV3{pe.x, pe.y, pf.x}, ui0[a],
                                                          ui1[a], v0, v1, flags, WK);
      __hip_atomic_fetch_add(&sU[wave][0][0][jj], v0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&sU[wave][0][1][jj], v0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&sU[wave][0][2][jj], v0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&sU[wave][1][0][jj], v1.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&sU[wave][1][1][jj], v1.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&sU[wave][1][2][jj], v1.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if (!sweeps) {
    } else if (PREC && (fmap & 2) && J >= It0 + NI) {   // relaxed product: packed single precision, coefficients once for both vectors
      const double qx = sO[0], qy = sO[1], qz = sO[2];
      const int a1 = NI - 1;
      const rbl_f2 xi2 = {(float)(xi[0] - qx), (float)(xi[a1] - qx)}, yi2 = {(float)(yi[0] - qy), (float)(yi[a1] - qy)},
                   zi2 = {(float)(zi[0] - qz), (float)(zi[a1] - qz)};
      const float two_z0 = (float)(2.0 * qz);
      rbl_f2 a0x = {0, 0}, a0y = {0, 0}, a0z = {0, 0}, a1x = {0, 0}, a1y = {0, 0}, a1z = {0, 0};
#pragma unroll 2
      for (int s = 0; s < TS; ++s) {
        const int jj = (lane + s) & (TS - 1);
        const RblPkCoef K = rbl_pk_coef<WALL>(xi2, yi2, zi2, sPf[0][jj], sPf[PREC ? 1 : 0][jj], sPf[PREC ? 2 : 0][jj], two_z0);
        rbl_f2 v0x = {0, 0}, v0y = {0, 0}, v0z = {0, 0}, v1x = {0, 0}, v1y = {0, 0}, v1z = {0, 0};
        rbl_pk_apply<WALL, false>(K, rbl_splat(sPf[PREC ? 3 : 0][jj]), rbl_splat(sPf[PREC ? 4 : 0][jj]), rbl_splat(sPf[PREC ? 5 : 0][jj]), a0x, a0y, a0z);
        rbl_pk_apply<WALL, false>(K, rbl_splat(sPf[PREC ? 6 : 0][jj]), rbl_splat(sPf[PREC ? 7 : 0][jj]), rbl_splat(sPf[PREC ? 8 : 0][jj]), a1x, a1y, a1z);
        rbl_pk_apply<WALL, true>(K, F0x, F0y, F0z, v0x, v0y, v0z);
        rbl_pk_apply<WALL, true>(K, F1x, F1y, F1z, v1x, v1y, v1z);
        __hip_atomic_fetch_add(&sU[wave][0][0][jj], (double)(v0x.x + v0x.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sU[wave][0][1][jj], (double)(v0y.x + v0y.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sU[wave][0][2][jj], (double)(v0z.x + v0z.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sU[wave][1][0][jj], (double)(v1x.x + v1x.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sU[wave][1][1][jj], (double)(v1y.x + v1y.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sU[wave][1][2][jj], (double)(v1z.x + v1z.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      ui0[0].x += (double)a0x.x; ui0[0].y += (double)a0y.x; ui0[0].z += (double)a0z.x;
      ui0[NI - 1].x += (double)a0x.y; ui0[NI - 1].y += (double)a0y.y; ui0[NI - 1].z += (double)a0z.y;
      ui1[0].x += (double)a1x.x; ui1[0].y += (double)a1y.x; ui1[0].z += (double)a1z.x;
      ui1[NI - 1].x += (double)a1x.y; ui1[NI - 1].y += (double)a1y.y; ui1[NI - 1].z += (double)a1z.y;
    } else if (J >= It0 + NI) {
      // both rows of the lane meet column jj: their contributions to its sums are added in registers (the pair routine accumulates
      // onto what it is handed: no instruction more) and go to LDS once -- 6 atomics a column instead of 12; at two waves a SIMD the
      // two-vector kernel waits for its LDS 5.7 x as long as the one-vector one (profiles/r05_two_vector_pmc.md)
      auto sweep = [&](auto nearchk) {
#pragma unroll 2
        for (int s = 0; s < TS; ++s) {
          const int jj = (lane + s) & (TS - 1);
          const double2_t pa = sP0[jj], pb = sP1[jj], pd = sP2[jj], pe = sP3[jj], pf = sP4[jj];
          RblV3 v0{0.0, 0.0, 0.0}, v1{0.0, 0.0, 0.0};
#pragma unroll
          for (int a = 0; a < NI; ++a)
            rbl_pair_sym2<WALL, true, decltype(nearchk)::value>(Pu, xi[a], yi[a], zi[a], Fi0[a], Fi1[a], pa.x, pa.y, pb.x,
                                                                RblV3{pb.y, pd.x, pd.y}, RblV3{pe.x, pe.y, pf.x}, ui0[a],
                                                                ui1[a], v0, v1, flags, WK);
          __hip_atomic_fetch_add(&sU[wave][0][0][jj], v0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&sU[wave][0][1][jj], v0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&sU[wave][0][2][jj], v0.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&sU[wave][1][0][jj], v1.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&sU[wave][1][1][jj], v1.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_fetch_add(&sU[wave][1][2][jj], v1.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
      if (far_tile) sweep(std::false_type{});
      else sweep(std::true_type{});
    } else {
#pragma unroll
      for (int a = 0; a < NI; ++a) {
        if (J == It0 + a) {       // diagonal tile: ordered pairs with the self term, one vector after the other
          for (int jj = 0; jj < TS; ++jj) {
            const double2_t pa = sP0[jj], pb = sP1[jj], pd = sP2[jj], pe = sP3[jj], pf = sP4[jj];
            rbl_pair_accum<WALL, true, true>(Pu, xi[a], yi[a], zi[a], pa.x, pa.y, pb.x, pb.y, pd.x, pd.y, jj == lane,
                                             ui0[a].x, ui0[a].y, ui0[a].z, flags);
            rbl_pair_accum<WALL, true, true>(Pu, xi[a], yi[a], zi[a], pa.x, pa.y, pb.x, pe.x, pe.y, pf.x, jj == lane,
                                             ui1[a].x, ui1[a].y, ui1[a].z, flags);
          }
        } else if (J > It0 + a) {
          for (int s = 0; s < TS; ++s) pair_step((lane + s) & (TS - 1), a, std::true_type{});
        }
      }
    }
    if (J > It00) {
      __syncthreads();
      const int t = threadIdx.x;
      for (int q = t; q < 2 * 3 * TS; q += TS * SW) {
        const int v = q / (3 * TS), rem = q - v * 3 * TS, l = rem / 3, k = rem - 3 * l;
        double sum = sU[0][v][k][l];
#pragma unroll
        for (int w = 1; w < SW; ++w) sum += sU[w][v][k][l];
        slabJ[sym_idxJ(L, g, v, (long)J * TS + l) + k] = sum;
      }
    }
  }
  if (wlive && c * C + C > It0) {
#pragma unroll
    for (int a = 0; a < NI; ++a) {
      if (It0 + a < T) {
        double *p = slabI + sym_idxI(L, c, 0, (long)(It0 + a) * TS + lane);
        p[0] = ui0[a].x; p[1] = ui0[a].y; p[2] = ui0[a].z;
        double *p1 = slabI + sym_idxI(L, c, 1, (long)(It0 + a) * TS + lane);
        p1[0] = ui1[a].x; p1[1] = ui1[a].y; p1[2] = ui1[a].z;
      }
    }
  }
  };
  if (!queue) {
    sweep_unit((int)blockIdx.y, (int)((blockIdx.x + blockIdx.y) % gridDim.x));
  } else {                                              // work queue, as in k_apply_M_sym
    __shared__ unsigned s_unit;
    const unsigned n_units = (unsigned)L.rowsG * (unsigned)L.nch;
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) s_unit = atomicAdd(queue, 1u);
      __syncthreads();
      const unsigned u = s_unit;
      if (u >= n_units) break;
      const int by = (int)(u / (unsigned)L.rowsG), bx = (int)(u - (unsigned)by * (unsigned)L.rowsG);
      sweep_unit(L.nch - 1 - by, (bx + by) % L.rowsG);
    }
  }
  if (flags) atomicOr(err, flags);
}

// Slab reduction.  Block = 64 consecutive entries of U (all in one blob tile J, since a tile is 192 entries)
// x RG groups; group q adds the slab entries e = q, q+RG, ... (first the row-sum slabs of the chunks that
// cover J, then the column-sum slabs of the row groups before J), the RG partial sums are combined in LDS in
// fixed order.  Many short independent load streams instead of one long one per entry.
constexpr int RG = 16;

// (the 64-lane sum in its one fixed order: rbl_wave_sum64, rbl_internal.hpp)

#ifndef RBL_REDUCE_UNROLL
#define RBL_REDUCE_UNROLL 4     // slab entries in flight per thread of k_reduce_sym (cfg 3, one box, tools/build_variant.sh: 4: 134 us, 8: 145, 16: 159)
#endif
template <bool WALL>
__global__ __launch_bounds__(64 * RG) void k_reduce_sym(const double *__restrict__ slabI,
                                                       const double *__restrict__ slabJ,
                                                       const double *__restrict__ r,
                                                       double *__restrict__ out, long N, SymLayout L, RblParams P,
                                                       unsigned *err, RblSaddleFuse fuse)
{
  // blockIdx.y = right-hand side (out is [gridDim.y][3N])
  const int v = blockIdx.y;
  out += (size_t)v * (size_t)(3 * N);
  __shared__ double sh[RG][64];
  const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long idx = (long)blockIdx.x * 64 + tx;           // over 3*N
  const bool live = idx < 3 * N;
  const long idc = live ? idx : 3 * N - 1;
  const long j = idc / 3;
  const int k = (int)(idc - 3 * j);
  const int J = (int)(j / TS);
  const int NI = L.NI, C = L.C;
  const int Is = J / NI;                                 // super-tile owning row tile J
  const bool owned = sym_row_owned(Is, L.i_first, L.i_step, L.SW);   // this launch owned the rows of tile J
  const int c0 = (NI * Is) / C;
  const int nI = owned ? L.nch - c0 : 0;
  const int Ilim = (J + NI - 1) / NI;                    // super-tiles I with NI*I < J
  const int nE = sym_rows_below(Ilim, L.i_first, L.i_step, L.SW);   // owned ones among them
  const int nJ = (nE + L.SW - 1) / L.SW;                 // row groups whose first super-tile precedes J
  double s = 0.0;
  int e = q;
#pragma unroll RBL_REDUCE_UNROLL
  for (; e < nI; e += RG) s += slabI[sym_idxI(L, c0 + e, v, j) + k];
  e -= nI;
#pragma unroll RBL_REDUCE_UNROLL
  for (; e < nJ; e += RG) s += slabJ[sym_idxJ(L, e, v, j) + k];
  sh[q][tx] = s;
  __syncthreads();
  double wtop = 0.0;                                     // this thread's entry of the fused saddle product (q == 0)
  if (q == 0 && live) {
    double t = sh[0][tx];
#pragma unroll
    for (int w = 1; w < RG; ++w) t += sh[w][tx];
    double sc = P.nf;
    if (WALL) sc *= damp_of(P, r[3 * j + 2]);
    out[idx] = sc * t;
    if (!isfinite(t)) atomicOr(err, (unsigned)RBL_FLAG_NONFINITE);
    if (fuse.lever) {                                    // saddle epilogue: w = M lambda - K U  (k_saddle_tail's arithmetic)
      const int b = (int)(j / fuse.N_blb);
      const double *u = fuse.U + 6 * b, *om = u + 3, *l = fuse.lever + 3 * j;
      const double ku = (k == 0) ? u[0] + l[2] * om[1] - l[1] * om[2]
                      : (k == 1) ? u[1] + l[0] * om[2] - l[2] * om[0]
                                 : u[2] + l[1] * om[0] - l[0] * om[1];
      wtop = sc * t - ku;
      fuse.w[idx] = wtop;
    }
  }
  if (fuse.lever && blockIdx.x == 0 && blockIdx.y == 0)   // ... and its body rows: K^T lambda as the preconditioner left it
    for (int i = threadIdx.x; i < fuse.nb6; i += 64 * RG) fuse.w[3 * N + i] = fuse.ktl[i];
  if (fuse.dotK > 0) {                                     // (uniform) first Gram-Schmidt pass: this block's share of V_k . w, wave q the vectors q, q + RG, ...
    if (q == 0) sh[0][tx] = live ? wtop : 0.0;             // (the wave that read sh[1..] above is the one that overwrites sh[0])
    __syncthreads();
    const double wv = sh[0][tx];
    for (int kk = q; kk < fuse.dotK; kk += RG) {
      double a = live ? fuse.dotV[(size_t)kk * (size_t)fuse.dotStride + idx] * wv : 0.0;
      a = rbl_wave_sum64(a);
      if (tx == 0) fuse.dotPart[(size_t)kk * fuse.dotNp + blockIdx.x] = a;
    }
    if (blockIdx.x == 0)                                   // the body rows' share, last slot
      for (int kk = q; kk < fuse.dotK; kk += RG) {
        double a = 0.0;
        for (int i = tx; i < fuse.nb6; i += 64) a = __builtin_fma(fuse.dotV[(size_t)kk * (size_t)fuse.dotStride + 3 * N + i], fuse.ktl[i], a);
        a = rbl_wave_sum64(a);
        if (tx == 0) fuse.dotPart[(size_t)kk * fuse.dotNp + fuse.dotNp - 1] = a;
      }
  }
}

// ---------------------------------------------------------------------------
// Multi-RHS matvec on the fp64 matrix cores: up to 16 right-hand sides per pass.
// A wavefront owns 16 rows i; per step every lane evaluates ONE ordered pair
// (i = i0 + (lane&15), j = j0 + 4 s + (lane>>4)) -- exactly the A-operand layout of
// v_mfma_f64_16x16x4 (A[m = lane&15][k = lane>>4]) -- builds its 3x3 block on the VALU
// and feeds the nine entries to nine MFMAs
//        D_a[i][n] += M_ij[a][b] * F_b[j][n]        a,b in {x,y,z}, n = RHS index,
// whose B-operands F_b[j = lane>>4][n = lane&15] are the packed forces.  The pair
// coefficients cost ~100 VALU instructions per lane-step, the nine MFMAs 9 x 2048 flop:
// the kernel is matrix-core bound and 16 vectors cost ~2-3x one single-RHS product.
// Layouts: Fp[j][b][16] (damped, zero padded), Up[split][i][a][16] (raw sums).
// ---------------------------------------------------------------------------
typedef double double4m_t __attribute__((ext_vector_type(4)));
constexpr int MR = 16;      // RHS per pass = MFMA N dimension
constexpr int MTJ = 256;    // j tile staged in LDS (positions only)

template <bool WALL, bool SELF>
__device__ __forceinline__ void mrhs_tile(const RblParams &P, const double *sx, const double *sy,
                                          const double *sz, const double *__restrict__ Fp, long j0,
                                          long i, double xi, double yi, double zi, int l15, int l4,
                                          double4m_t &dx_, double4m_t &dy_, double4m_t &dz_,
                                          unsigned &flags, const RblWallK &K)
{
  const double *fp = Fp + ((size_t)(j0 + l4) * 3) * MR + l15;
#pragma unroll 2
  for (int s = 0; s < MTJ / 4; ++s) {
    const int jj = 4 * s + l4;
    const double f0 = fp[0], f1 = fp[MR], f2 = fp[2 * MR];
    fp += (size_t)4 * 3 * MR;
    double m[9];
    rbl_pair_block_fast<WALL, SELF, true>(P, xi, yi, zi, sx[jj], sy[jj], sz[jj], SELF && (j0 + jj == i), m, flags, K);
    dx_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[0], f0, dx_, 0, 0, 0);
    dx_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[1], f1, dx_, 0, 0, 0);
    dx_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[2], f2, dx_, 0, 0, 0);
    dy_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[3], f0, dy_, 0, 0, 0);
    dy_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[4], f1, dy_, 0, 0, 0);
    dy_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[5], f2, dy_, 0, 0, 0);
    dz_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[6], f0, dz_, 0, 0, 0);
    dz_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[7], f1, dz_, 0, 0, 0);
    dz_ = __builtin_amdgcn_mfma_f64_16x16x4f64(m[8], f2, dz_, 0, 0, 0);
  }
}

template <bool WALL>
__global__ __launch_bounds__(256) void k_apply_M_mrhs(const double *__restrict__ r,
                                                      const double *__restrict__ Fp,
                                                      double *__restrict__ Up, long N, long Npad,
                                                      long jchunk, RblParams P, unsigned *err)
{
  __shared__ double sx[MTJ], sy[MTJ], sz[MTJ];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long i_blk0 = (long)blockIdx.x * 64;
  const long i = i_blk0 + wave * 16 + l15;
  const long ic = i < N ? i : N - 1;
  const double xi = r[3 * ic] * P.inv_a, yi = r[3 * ic + 1] * P.inv_a, zi = r[3 * ic + 2] * P.inv_a;   // radius-scaled
  const RblParams Pu = unit_params(P);
  const RblWallK WK = rbl_wall_k_literal();   // (resident constants cost this kernel 1 %: register pressure)
  unsigned flags = 0;
  const long j_begin = (long)blockIdx.y * jchunk;
  const long j_end = (j_begin + jchunk < Npad) ? j_begin + jchunk : Npad;
  double4m_t ax = {0, 0, 0, 0}, ay = {0, 0, 0, 0}, az = {0, 0, 0, 0};
  for (long j0 = j_begin; j0 < j_end; j0 += MTJ) {
    const long j = j0 + t;
    double x, y, z;
    if (j < N) {
      x = r[3 * j] * P.inv_a; y = r[3 * j + 1] * P.inv_a; z = r[3 * j + 2] * P.inv_a;
      if (WALL && z < 0.0) flags |= RBL_FLAG_BELOW_WALL;
    } else {  // padding (its packed forces are zero)
      x = 1.0e15 * (double)(2 + (j - N)); y = 0.0; z = 1.0;
    }
    __syncthreads();
    sx[t] = x; sy[t] = y; sz[t] = z;
    __syncthreads();
    const bool diag = (j0 < i_blk0 + 64) && (j0 + MTJ > i_blk0);  // block-uniform
    if (diag)
      mrhs_tile<WALL, true>(Pu, sx, sy, sz, Fp, j0, i < N ? i : -1, xi, yi, zi, l15, l4, ax, ay, az, flags, WK);
    else
      mrhs_tile<WALL, false>(Pu, sx, sy, sz, Fp, j0, -1, xi, yi, zi, l15, l4, ax, ay, az, flags, WK);
  }
  // D layout of v_mfma_f64_16x16x4: row = (lane>>4) + 4 v, col = lane & 15
  double *up = Up + (size_t)blockIdx.y * (size_t)Npad * 3 * MR;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const long io = i_blk0 + wave * 16 + l4 + 4 * v;
    if (io < N) {
      double *p = up + ((size_t)io * 3) * MR + l15;
      p[0] = ax[v]; p[MR] = ay[v]; p[2 * MR] = az[v];
    }
  }
  if (i >= N) flags &= ~RBL_FLAG_OVERLAP;
  if (flags) atomicOr(err, flags);
}

// pack nrhs column-major RHS (n3 x nrhs) into Fp[j][b][16], damped, zero padded
template <bool WALL>
__global__ void k_pack_rhs(const double *__restrict__ F, const double *__restrict__ r, long N, long Npad,
                           int nrhs, RblParams P, double *__restrict__ Fp, long ldF)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over Npad*3*16
  if (idx >= Npad * 3 * MR) return;
  const int n = (int)(idx % MR);
  const long jb = idx / MR;  // j*3 + b
  const long j = jb / 3;
  double v = 0.0;
  if (j < N && n < nrhs) {
    v = F[(size_t)n * (size_t)ldF + jb];
    if (WALL) v *= damp_of(P, r[3 * j + 2]);
  }
  Fp[idx] = v;
}

template <bool WALL>
__global__ void k_unpack_rhs(const double *__restrict__ Up, const double *__restrict__ r, long N, long Npad,
                             int nrhs, int nsplit, RblParams P, double *__restrict__ out, unsigned *err, long ldO)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over N*3*nrhs, ia fastest
  if (idx >= 3 * N * nrhs) return;
  const long ia = idx % (3 * N);
  const int n = (int)(idx / (3 * N));
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += Up[(size_t)k * (size_t)Npad * 3 * MR + (size_t)ia * MR + n];
  double sc = P.nf;
  if (WALL) sc *= damp_of(P, r[3 * (ia / 3) + 2]);
  out[(size_t)n * (size_t)ldO + ia] = sc * s;
  if (!isfinite(s)) atomicOr(err, (unsigned)RBL_FLAG_NONFINITE);
}

// ---------------------------------------------------------------------------
// Dense assembly, column-major n3 x n3 (ld = n3).  Block = 256 consecutive i
// (768 consecutive rows) x JB consecutive j (3*JB columns).  Each 3x3 block is
// computed with the reference's (min,max) roles and transposed when i > j, so
// the matrix equals the reference's entry for entry.  Columns are written from
// LDS as contiguous 8-B-per-lane runs.
// ---------------------------------------------------------------------------
constexpr int JB = 16;

template <bool WALL>
__global__ __launch_bounds__(256) void k_build_M(const double *__restrict__ r,
                                                 double *__restrict__ M, long N,
                                                 int scale_damp, RblParams P, unsigned *err,
                                                 long strideR, long strideM, int lower_tiles)
{
  // lower_tiles > 0 (the per-body matrices a factorisation follows): blocks that lie entirely ABOVE the diagonal tiles of that
  // edge are not written -- a factorisation reads the lower triangle and whole diagonal tiles only: half the bytes
  if (lower_tiles > 0 && 3 * ((long)blockIdx.x * 256 + 256) <= ((3 * (long)blockIdx.y * JB) / lower_tiles) * lower_tiles) return;
  r += (size_t)blockIdx.z * (size_t)strideR;   // batched: one blob set / one matrix per blockIdx.z
  M += (size_t)blockIdx.z * (size_t)strideM;
  __shared__ double col[3][768];
  const int t = threadIdx.x;
  const long i0 = (long)blockIdx.x * 256;
  const long i = i0 + t;
  const bool valid = i < N;
  const long ic = valid ? i : N - 1;
  const long n3 = 3 * N;
  const double xi = r[3 * ic], yi = r[3 * ic + 1], zi = r[3 * ic + 2];
  const double di = scale_damp ? damp_of(P, zi) : 1.0;
  const long nrow_blk = ((N - i0) < 256 ? (N - i0) : 256) * 3;
  unsigned flags = 0;
  const long jb0 = (long)blockIdx.y * JB;
  for (int jj = 0; jj < JB; ++jj) {
    const long j = jb0 + jj;
    if (j >= N) break;  // block-uniform
    const double xj = r[3 * j], yj = r[3 * j + 1], zj = r[3 * j + 2];
    double b[9];
    const bool self = (ic == j);
    if (ic <= j) {
      rbl_block_ref(P, WALL, xi, yi, zi, xj, yj, zj, self, b, flags);
    } else {  // stored block is (j,i); ours is its transpose (c_rigid_obj.cpp:451)
      double bt[9];
      rbl_block_ref(P, WALL, xj, yj, zj, xi, yi, zi, false, bt, flags);
      b[0] = bt[0]; b[1] = bt[3]; b[2] = bt[6];
      b[3] = bt[1]; b[4] = bt[4]; b[5] = bt[7];
      b[6] = bt[2]; b[7] = bt[5]; b[8] = bt[8];
    }
    const double dj = scale_damp ? damp_of(P, zj) : 1.0;
    {
#pragma clang fp contract(off)
      for (int k = 0; k < 9; ++k) {
        double v = b[k] * P.nf;                   // Mob *= norm_fact  (:456)
        if (scale_damp) v = (di * v) * dj;        // B * Mob * B       (:669)
        b[k] = v;
      }
    }
    __syncthreads();
    for (int q = 0; q < 3; ++q)
      for (int p = 0; p < 3; ++p) col[q][3 * t + p] = b[3 * p + q];
    __syncthreads();
    for (int q = 0; q < 3; ++q) {
      double *dst = M + (size_t)(3 * j + q) * (size_t)n3 + (size_t)(3 * i0);
      for (int e = t; e < nrow_blk; e += 256) dst[e] = col[q][e];
    }
  }
  if (!valid) flags &= ~RBL_FLAG_OVERLAP;
  if (flags) atomicOr(err, flags);
}

// ---------------------------------------------------------------------------
__global__ void k_blob_positions(const double *__restrict__ X, const double *__restrict__ Q,
                                 const double *__restrict__ cfg, int N_blb, int body_begin,
                                 long n_out_blobs, double *__restrict__ out)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_out_blobs) return;
  const int b = body_begin + (int)(idx / N_blb);
  const int k = (int)(idx % N_blb);
  const double w = Q[4 * b], x = Q[4 * b + 1], y = Q[4 * b + 2], z = Q[4 * b + 3];
  // unit quaternion -> rotation (same expansion Eigen uses; reference :258)
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double c0 = cfg[3 * k], c1 = cfg[3 * k + 1], c2 = cfg[3 * k + 2];
  {
#pragma clang fp contract(off)
    out[3 * idx] = c0 * (1 - (tyy + tzz)) + c1 * (txy - twz) + c2 * (txz + twy) + X[3 * b];
    out[3 * idx + 1] = c0 * (txy + twz) + c1 * (1 - (txx + tzz)) + c2 * (tyz - twx) + X[3 * b + 1];
    out[3 * idx + 2] = c0 * (txz - twy) + c1 * (tyz + twx) + c2 * (1 - (txx + tyy)) + X[3 * b + 2];
  }
}

// ---------------------------------------------------------------------------
__global__ void k_pair_blocks(const double *__restrict__ ri, const double *__restrict__ rj,
                              const int32_t *__restrict__ ii, const int32_t *__restrict__ jj,
                              long n, int wall, int mode, RblParams P, double *__restrict__ out,
                              unsigned *err)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  unsigned flags = 0;
  const double xi = ri[3 * k], yi = ri[3 * k + 1], zi = ri[3 * k + 2];
  const double xj = rj[3 * k], yj = rj[3 * k + 1], zj = rj[3 * k + 2];
  const int i = ii[k], j = jj[k];
  double b[9];
  if (mode == 0) {
    if (i <= j) {
      rbl_block_ref(P, wall != 0, xi, yi, zi, xj, yj, zj, i == j, b, flags);
    } else {
      double bt[9];
      rbl_block_ref(P, wall != 0, xj, yj, zj, xi, yi, zi, false, bt, flags);
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) b[3 * p + q] = bt[3 * q + p];
    }
  } else {
    // fast path: columns of the block are its action on unit forces
    for (int c = 0; c < 3; ++c) {
      double ux = 0, uy = 0, uz = 0;
      const double fx = c == 0, fy = c == 1, fz = c == 2;
      if (wall)
        rbl_pair_accum<true, true>(P, xi, yi, zi, xj, yj, zj, fx, fy, fz, i == j, ux, uy, uz, flags);
      else
        rbl_pair_accum<false, true>(P, xi, yi, zi, xj, yj, zj, fx, fy, fz, i == j, ux, uy, uz, flags);
      b[c] = ux; b[3 + c] = uy; b[6 + c] = uz;
    }
    if (wall && (zi < 0.0 || zj < 0.0)) flags |= RBL_FLAG_BELOW_WALL;
  }
  for (int e = 0; e < 9; ++e) out[9 * k + e] = b[e] * P.nf;
  if (flags) atomicOr(err, flags);
}

// ---------------------------------------------------------------------------
// Counter-based normal generator: Philox-4x32-10 -> Box-Muller, two doubles
// from each 128-bit block.  Reproducible for (seed, offset) on any grid.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void k_normal(uint64_t seed, uint64_t offset, long n, double *__restrict__ out)
{
  const long pair = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one pair of outputs
  if (2 * pair >= n) return;
  const uint64_t ctr = offset + (uint64_t)pair;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x52424C31u /* "RBL1" */, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint64_t a = ((uint64_t)c[0] << 32) | c[1];
  const uint64_t b = ((uint64_t)c[2] << 32) | c[3];
  const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);  // (0,1)
  const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincos(6.283185307179586476925 * u2, &sn, &cs);
  out[2 * pair] = rad * cs;
  if (2 * pair + 1 < n) out[2 * pair + 1] = rad * sn;
}

// ---------------------------------------------------------------------------
// BLAS-1 helpers (Lanczos): deterministic two-stage reductions.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dot2_partial(const double *__restrict__ x,
                                                      const double *__restrict__ y,
                                                      const double *__restrict__ z, long n,
                                                      double *__restrict__ part)
{
  __shared__ double s0[256], s1[256];
  double a = 0.0, b = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double xv = x[i];
    a = __builtin_fma(xv, y[i], a);
    if (z) b = __builtin_fma(xv, z[i], b);
  }
  s0[threadIdx.x] = a; s1[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s0[threadIdx.x] += s0[threadIdx.x + s]; s1[threadIdx.x] += s1[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s0[0]; part[2 * blockIdx.x + 1] = s1[0]; }
}

__global__ __launch_bounds__(256) void k_dot2_final(const double *__restrict__ part, int nblk,
                                                    double *__restrict__ out2)
{
  __shared__ double s0[256], s1[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
  s0[threadIdx.x] = a; s1[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { s0[threadIdx.x] += s0[threadIdx.x + s]; s1[threadIdx.x] += s1[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = s0[0]; out2[1] = s1[0]; }
}

__global__ void k_axpby(long n, double a, const double *x, double b, const double *y, double *out)   // out may alias x or y
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.0);
}

// out = x - a y - b z + (w ? b w : 0): the right-hand side of the stochastic step in one launch (rbl_RHS_and_Midpoint_dev)
__global__ void k_rhs_combine(long n, const double *x, double a, const double *y, double b, const double *z, const double *w, double *out)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = x[i] - a * y[i];                              // (the order of the three separate updates it replaces)
  r = r - b * z[i];
  if (w) r = r + b * w[i];
  out[i] = r;
}

// ---- Lanczos recurrence without host round trips -------------------------------------------
// The three-term recurrence needs two inner products per iteration; here they stay on the device:
// a kernel leaves per-block partial sums, the NEXT kernel's every block re-adds them in the same
// fixed order (bitwise identical scalar in all blocks) and uses the result.  alpha / beta are
// stored for the host, which only looks at them when it tests convergence.
constexpr int LZ_BLOCKS = 512;

__device__ __forceinline__ double lz_block_sum(double v, double *sh)
{
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const double t = sh[0];
  __syncthreads();
  return t;
}

__device__ __forceinline__ double lz_sum_parts(const double *__restrict__ part, int np, double *sh)
{
  double a = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) a += part[i];
  return lz_block_sum(a, sh);
}

// u -= beta_prev * vprev (when vprev != null);  partA[block] = sum v_i u_i
// (blockIdx.y = vector of a lock-step pair: vectors vs doubles apart, their scalars ss, their partial sums ps)
__global__ __launch_bounds__(256) void k_lz_a(long n, double *__restrict__ u, const double *__restrict__ v,
                                              const double *__restrict__ vprev, const double *__restrict__ bprev,
                                              double *__restrict__ partA, long vs, long ss, long ps)
{
  __shared__ double sh[256];
  u += blockIdx.y * vs; v += blockIdx.y * vs; partA += blockIdx.y * ps;
  if (vprev) { vprev += blockIdx.y * vs; bprev += blockIdx.y * ss; }
  const double b = vprev ? *bprev : 0.0;
  double a = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double ui = u[i];
    if (vprev) { ui = __builtin_fma(-b, vprev[i], ui); u[i] = ui; }
    a = __builtin_fma(v[i], ui, a);
  }
  a = lz_block_sum(a, sh);
  if (threadIdx.x == 0) partA[blockIdx.x] = a;
}

// alpha = sum(partA);  u -= alpha v;  partB[block] = sum u_i^2
__global__ __launch_bounds__(256) void k_lz_b(long n, double *__restrict__ u, const double *__restrict__ v,
                                              const double *__restrict__ partA, int np,
                                              double *__restrict__ alpha_out, double *__restrict__ partB, long vs, long ss, long ps)
{
  __shared__ double sh[256];
  u += blockIdx.y * vs; v += blockIdx.y * vs; partA += blockIdx.y * ps; partB += blockIdx.y * ps; alpha_out += blockIdx.y * ss;
  const double al = lz_sum_parts(partA, np, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) *alpha_out = al;
  double a = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double ui = __builtin_fma(-al, v[i], u[i]);
    u[i] = ui;
    a = __builtin_fma(ui, ui, a);
  }
  a = lz_block_sum(a, sh);
  if (threadIdx.x == 0) partB[blockIdx.x] = a;
}

// beta = sqrt(sum(partB));  vnext = u / beta   (zeros on breakdown)
// (asrc != null: the re-orthogonalised Lanczos step also files its diagonal entry, beta_out[aoff] = asrc[pair * as])
__global__ __launch_bounds__(256) void k_lz_c(long n, const double *__restrict__ u,
                                              const double *__restrict__ partB, int np,
                                              double *__restrict__ beta_out, double *__restrict__ vnext, long vs, long ss, long ps,
                                              const double *__restrict__ asrc, long as, long aoff)
{
  __shared__ double sh[256];
  u += blockIdx.y * vs; vnext += blockIdx.y * vs; partB += blockIdx.y * ps; beta_out += blockIdx.y * ss;
  const double be = sqrt(lz_sum_parts(partB, np, sh));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *beta_out = be;
    if (asrc) beta_out[aoff] = asrc[blockIdx.y * as];
  }
  const double inv = (be > 1e-300) ? 1.0 / be : 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) vnext[i] = inv * u[i];
}

// out = sum_p coef[p] V[p]   (V: m vectors of length n, contiguous)
__global__ __launch_bounds__(256) void k_lz_combine(long n, const double *__restrict__ V,
                                                    const double *__restrict__ coef, int m,
                                                    double *__restrict__ out, long stride)
{
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a = 0.0;
  for (int p = 0; p < m; ++p) a = __builtin_fma(coef[p], V[(size_t)p * stride + i], a);
  out[i] = a;
}

// The estimate and its last correction in ONE pass over the basis: zx = V coef[0..m), zd = V coef[m..2m)  (the stopping
// test of the preconditioned root in the norm of the increment).  blockIdx.y = recurrence of a lock-step pair: basis
// vsep, coefficients csep, outputs osep doubles further.  zx is k_lz_combine's sum, term for term.
__global__ __launch_bounds__(256) void k_lz_combine_xd(long n, const double *__restrict__ V, long stride, long vsep,
                                                       const double *__restrict__ coef, long csep, int m,
                                                       double *__restrict__ zx, double *__restrict__ zd, long osep)
{
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  V += (size_t)blockIdx.y * (size_t)vsep; coef += (size_t)blockIdx.y * (size_t)csep;
  double a = 0.0, b = 0.0;
  for (int p = 0; p < m; ++p) {
    const double v = V[(size_t)p * stride + i];
    a = __builtin_fma(coef[p], v, a);
    b = __builtin_fma(coef[m + p], v, b);
  }
  zx[(size_t)blockIdx.y * (size_t)osep + i] = a;
  zd[(size_t)blockIdx.y * (size_t)osep + i] = b;
}

// o_v <- B o_v in place and the per-block partial sums of |B o_v|^2   (blockIdx.y = vector, `pitch` doubles apart;
// part[v][gridDim.x], added up by the host in block order)
__global__ __launch_bounds__(256) void k_damp_sqnorm(RblParams P, const double *__restrict__ r, long n_blobs,
                                                     double *__restrict__ o, long pitch, double *__restrict__ part)
{
  __shared__ double sh[256];
  o += (size_t)blockIdx.y * (size_t)pitch;
  double a = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 3 * n_blobs; i += (long)gridDim.x * 256) {
    const double x = damp_of(P, r[3 * (i / 3) + 2]) * o[i];
    o[i] = x;
    a = __builtin_fma(x, x, a);
  }
  a = lz_block_sum(a, sh);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = a;
}

// ---- Arnoldi orthogonalisation (GMRES) --------------------------------------------------------
// h = V[0..k)^T w as per-block partial sums (grid: blocks x vectors); the consumer re-adds them.
// (vstr: doubles between consecutive basis vectors; blockIdx.z = member of a lock-step pair of recurrences, whose basis /
// vector / partial sums lie pv / pw / pp doubles further -- the re-orthogonalised two-vector Lanczos)
__global__ __launch_bounds__(256) void k_mdot_partial(const double *__restrict__ V, long n, const double *__restrict__ w,
                                                      double *__restrict__ part, long vstr, long pv, long pw, long pp)
{
  __shared__ double sh[256];
  V += (size_t)blockIdx.z * (size_t)pv; w += (size_t)blockIdx.z * (size_t)pw; part += (size_t)blockIdx.z * (size_t)pp;
  const double *v = V + (size_t)blockIdx.y * (size_t)vstr;
  double a = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a = __builtin_fma(v[i], w[i], a);
  a = lz_block_sum(a, sh);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = a;
}

constexpr int GM_MAXK = 256;     // basis vectors an Arnoldi step can orthogonalise against (no restart)

// The two Gram-Schmidt passes and the normalisation of an Arnoldi step in four launches instead of six (a launch
// costs ~4.7 us of stream time whatever it does; at 8 100 blobs that was a third of an iteration):
//   k_mdot_partial            partials of h1 = V^T w
//   k_arnoldi_upd<false>      w -= V h1; the block's elements of w are final for this pass, so the SAME kernel leaves
//                             the partials of h2 = V^T w (second pass) over them
//   k_arnoldi_upd<true>       w -= V h2, Hcol += h2; partials of |w|^2
//   k_lz_c                    H[j+1][j] = |w|, V_{j+1} = w / |w|

// A thread owns up to AR_EPT elements (the launcher sizes the grid for that) and keeps them in registers; the basis is
// read eight vectors at a time with all loads issued before the first multiply-add (a loop of load -> fma over a
// run-time number of vectors is one memory round trip per vector: 16 us at 16 vectors for a 25 000-entry system).
constexpr int AR_EPT = 4;
template <bool LAST>
__global__ __launch_bounds__(256) void k_arnoldi_upd(const double *__restrict__ V, long n, int k, double *__restrict__ w,
                                                     const double *__restrict__ pin, int npin, double *__restrict__ Hcol,
                                                     double *__restrict__ pout, long vstr, long pv, long pw, long pp, long ph)
{
  __shared__ double h[GM_MAXK];
  __shared__ double sw[16][GM_MAXK];                 // per 16-lane row of the block
  __shared__ double sh[256];
  const int t = threadIdx.x;
  V += (size_t)blockIdx.y * (size_t)pv; w += (size_t)blockIdx.y * (size_t)pw;      // member of a lock-step pair (see k_mdot_partial)
  pin += (size_t)blockIdx.y * (size_t)pp; pout += (size_t)blockIdx.y * (size_t)pp; Hcol += (size_t)blockIdx.y * (size_t)ph;
  // h_v = sum of the previous kernel's partials: a 16-lane row per vector (one thread per vector walked npin dependent
  // loads: at 97 partials that prologue WAS the kernel)
  if (npin > 128) {
    // hundreds of partials per vector (the fused product leaves one per 64 entries): a WAVE per vector, four vectors of a wave
    // in flight, all loads issued before the first add (16 lanes walking 24 dependent loads each doubled this kernel's time)
    const int wv = t >> 6, lane = t & 63;
    for (int v0 = wv; v0 < k; v0 += 16) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int base = 0; base < npin; base += 512) {     // (512 partials per round; the second pass of a large system hands over 1024)
        double x[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int v = v0 + 4 * u, b = base + lane + 64 * e;
            x[u][e] = (v < k && b < npin) ? pin[(size_t)v * npin + b] : 0.0;
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc[u] += ((x[u][0] + x[u][1]) + (x[u][2] + x[u][3])) + ((x[u][4] + x[u][5]) + (x[u][6] + x[u][7]));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = v0 + 4 * u;
        double a = rbl_wave_sum64(acc[u]);
        if (lane == 0 && v < k) {
          h[v] = a;
          if (blockIdx.x == 0) Hcol[v] = LAST ? Hcol[v] + a : a;
        }
      }
    }
  } else
  for (int v = t >> 4; v < k; v += 16) {
    double a = 0.0;
    for (int b = t & 15; b < npin; b += 16) a += pin[(size_t)v * npin + b];
    a = rbl_row_sum16(a);
    if ((t & 15) == 0) {
      h[v] = a;
      if (blockIdx.x == 0) Hcol[v] = LAST ? Hcol[v] + a : a;
    }
  }
  __syncthreads();
  const long stride = (long)gridDim.x * 256, i0 = (long)blockIdx.x * 256 + t;
  const long span = stride * AR_EPT;                 // elements one pass of the grid covers
  const int nchunk = (int)((n + span - 1) / span);   // 1 unless the system has more than AR_BLOCKS x 256 x AR_EPT entries
  double x[AR_EPT];
  double nrm = 0.0;
  for (int c = 0; c < nchunk; ++c) {                 // w -= V h
    const long ic = i0 + (long)c * span;
#pragma unroll
    for (int e = 0; e < AR_EPT; ++e) { const long i = ic + e * stride; x[e] = i < n ? w[i] : 0.0; }
    for (int v0 = 0; v0 < k; v0 += 8) {
      double m[8][AR_EPT];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < AR_EPT; ++e) {
          const long i = ic + e * stride;
          m[u][e] = (v0 + u < k && i < n) ? V[(size_t)(v0 + u) * (size_t)vstr + i] : 0.0;
        }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double hv = v0 + u < k ? h[v0 + u] : 0.0;
#pragma unroll
        for (int e = 0; e < AR_EPT; ++e) x[e] = __builtin_fma(-hv, m[u][e], x[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < AR_EPT; ++e) {
      const long i = ic + e * stride;
      if (i < n) w[i] = x[e];
      nrm = __builtin_fma(x[e], x[e], nrm);          // (elements beyond n are zero)
    }
  }
  if (LAST) {
    nrm = lz_block_sum(nrm, sh);
    if (t == 0) pout[blockIdx.x] = nrm;
    return;
  }
  for (int v0 = 0; v0 < k; v0 += 8) {                // this block's share of V_v . w over its own elements
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < nchunk; ++c) {
      const long ic = i0 + (long)c * span;
      if (nchunk > 1) {                              // (one chunk: the updated entries are still in registers)
#pragma unroll
        for (int e = 0; e < AR_EPT; ++e) { const long i = ic + e * stride; x[e] = i < n ? w[i] : 0.0; }
      }
      double m[8][AR_EPT];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < AR_EPT; ++e) {
          const long i = ic + e * stride;
          m[u][e] = (v0 + u < k && i < n) ? V[(size_t)(v0 + u) * (size_t)vstr + i] : 0.0;
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < AR_EPT; ++e) a[u] = __builtin_fma(m[u][e], x[e], a[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      double r = a[u];
      r = rbl_row_sum16(r);                          // sums inside the 16-lane rows by DPP
      if ((t & 15) == 0 && v0 + u < k) sw[t >> 4][v0 + u] = r;
    }
  }
  __syncthreads();
  for (int v = t; v < k; v += 256) {
    double a = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += sw[r][v];      // fixed order
    pout[(size_t)v * gridDim.x + blockIdx.x] = a;
  }
}

__global__ void k_scale_by_damp(RblParams P, const double *__restrict__ r, long n_blobs,
                                const double *__restrict__ in, double *__restrict__ out)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * n_blobs) out[i] = damp_of(P, r[3 * (i / 3) + 2]) * in[i];
}

}  // namespace

// =============================================================================
// launchers
// =============================================================================
RblParams rbl_make_params(double a, double eta)
{
  RblParams P;
  P.a = a;
  P.inv_a = 1.0 / a;
  P.nf = 1.0 / (8.0 * M_PI * eta * a);
  P.four_a2 = 4.0 * a * a;
  P.tiny2 = (1e-12 * a) * (1e-12 * a);
  P.c_near_A = -0.375 / a;
  P.c_near_B = 0.125 / a;
  P.no_damp = 0;
  return P;
}

static int choose_jsplit(int64_t n_blobs, int64_t nrows, int n_cu, int override_)
{
  const int64_t itiles = (nrows + TB - 1) / TB;
  const int64_t jtiles = (n_blobs + TB - 1) / TB;
  int64_t js;
  if (override_ > 0) {
    js = override_;
  } else {
    // aim for ~8 rounds of (4 blocks per CU) so the tail round costs < ~6 %
    const int64_t target = (int64_t)(n_cu > 0 ? n_cu : 256) * 4 * 8;
    js = (target + itiles - 1) / itiles;
  }
  if (js > jtiles) js = jtiles;
  if (js < 1) js = 1;
  return (int)js;
}

size_t rbl_apply_M_part_bytes(int64_t n_blobs, int64_t nrows, int n_cu, int jsplit_override,
                              int *jsplit_out)
{
  const int js = choose_jsplit(n_blobs, nrows, n_cu, jsplit_override);
  if (jsplit_out) *jsplit_out = js;
  return js > 1 ? (size_t)js * (size_t)(3 * nrows) * sizeof(double) : 0;
}

void rbl_launch_apply_M(hipStream_t st, const RblParams &P, bool wall, const double *d_F,
                        const double *d_r, int64_t n_blobs, int64_t row_begin,
                        int64_t row_end, double *d_out, double *d_part, int jsplit,
                        int /*variant*/, unsigned *d_err)
{
  const int64_t nrows = row_end - row_begin;
  if (nrows <= 0 || n_blobs <= 0) return;
  const int64_t itiles = (nrows + TB - 1) / TB;
  const int64_t jtiles = (n_blobs + TB - 1) / TB;
  const int64_t jchunk = ((jtiles + jsplit - 1) / jsplit) * TB;
  // jsplit may shrink when the chunks are rounded up to whole tiles
  const int js_eff = (int)((n_blobs + jchunk - 1) / jchunk);
  dim3 grid((unsigned)itiles, (unsigned)js_eff), block(TB);
  if (wall)
    hipLaunchKernelGGL(k_apply_M<true>, grid, block, 0, st, d_r, d_F, d_out, d_part,
                       (long)n_blobs, (long)row_begin, (long)row_end, (long)jchunk, js_eff, P, d_err);
  else
    hipLaunchKernelGGL(k_apply_M<false>, grid, block, 0, st, d_r, d_F, d_out, d_part,
                       (long)n_blobs, (long)row_begin, (long)row_end, (long)jchunk, js_eff, P, d_err);
  if (js_eff > 1) {
    const int64_t n = 3 * nrows;
    dim3 g2((unsigned)((n + 255) / 256)), b2(256);
    if (wall)
      hipLaunchKernelGGL(k_reduce_parts<true>, g2, b2, 0, st, d_part, d_r, d_out,
                         (long)row_begin, (long)nrows, js_eff, P, d_err);
    else
      hipLaunchKernelGGL(k_reduce_parts<false>, g2, b2, 0, st, d_part, d_r, d_out,
                         (long)row_begin, (long)nrows, js_eff, P, d_err);
  }
}

// ---- symmetric variant ------------------------------------------------------
// tune (per context, RBL_OPT_SYM_*): chunk > 0 forces the chunk length C; ni2 > 0 the rows per lane of the two-vector kernel
static SymLayout sym_geometry(int64_t n_blobs, int n_cu, int i_first, int i_step, int nrhs, const RblSymTune &tune)
{
  SymLayout L;
  const int t = (int)((n_blobs + TS - 1) / TS);
  // 2 rows per lane once there is parallelism to spare: same speed on one GPU (the kernel is
  // VALU-issue bound either way) but half the column-sum slab to write and re-read
  int ni = (t >= 128 * i_step) ? 2 : 1;
  // one vector on one GPU: two rows per lane from 120 tiles on -- the wave-unit kernel's NI = 2 form shares the column reads and the
  // rotating column sums between two pairs (43.5 instead of 49 instructions per pair in free space): 8 100 blobs 71 against 74-77 us,
  // cfg 2's Brownian step -2 to -3 % (tools/bench_cfg2_step.py, interleaved); at 4 860 blobs it is the slower one (38 against 34 us)
  if (nrhs == 1 && i_step == 1 && t >= 120) ni = 2;
  // two vectors: one row per lane a little longer (8 346 wall blobs 175 against 186 us, 8 100 free 110 against 124; 12 960: 259 / 253)
  if (nrhs == 2 && t < 176 * i_step && !tune.relaxed) ni = 1;   // (the relaxed sweep carries the two rows of a lane: it keeps NI = 2)
  if (nrhs == 2 && tune.ni2 > 0) ni = tune.ni2;
  if (nrhs == 1 && tune.ni1 > 0) ni = tune.ni1;
  const int tsup = (t + ni - 1) / ni;                    // row super-tiles
  // workgroups of SW_LARGE waves (= units of SW_LARGE consecutive super-tiles, see sym_row_of) for large systems; a shard
  // keeps them as long as every rank still gets >= 8 units.  On one GPU they start at 160 super-tiles (20 480 blobs): below,
  // single-wave workgroups fill the chip a little better (tools/bench_midrange.py, wall, one box, forced shapes back to back:
  // 8 346 / 12 198 / 16 050 blobs 0.142 / 0.244 / 0.384 ms with four waves against 0.132 / 0.232 / 0.371 with one; 23 754: 0.778 /
  // 0.783; 32 742: 1.41 / 1.49; the two-vector product is indifferent below 24 000 blobs and 9 % better with four waves above)
  int sw = (ni == 2 && tsup >= SW_LARGE * i_step * (i_step > 1 ? 8 : 40)) ? SW_LARGE : 1;
  if (tune.sw == 1 || (tune.sw == SW_LARGE && ni == 2)) sw = tune.sw;   // (the only workgroup shapes the product build instantiates: 1, or 4 with two rows per lane)
  const int tunits = (tsup + sw - 1) / sw;
  const int rowsI = ((tunits + i_step - 1) / i_step) * sw;
  // a unit sweeps <= C column tiles.  Measured (tools/tune_sym_chunk.py): short chunks win -- many
  // wave-units balance the triangular work and hide tile-boundary latency; C = 4..16 is flat at
  // 128 400 blobs (29.8-29.9 ms vs 30.8 at C = 64), C = 2 best at 8 100.  Aim for ~8 rounds of
  // (4 waves/SIMD x 4 SIMD x CUs) wave-units, capped at 16 tiles.
  const double pairs = 0.5 * (double)rowsI * (double)t;
  const double target_units = (double)(n_cu > 0 ? n_cu : 256) * 16.0 * 8.0;
  int c = (int)(pairs / target_units);
  if (c < 1) c = 1;
  if (c > 16) c = 16;
  // Shards of large systems (four-wave workgroups drawing their units from the work queue: balance no longer needs thousands of
  // short units, and a shard's row sums are written once per chunk).  Measured per rank at cfg 3, tools/bench_shard_kernel.py with
  // CHUNK forced, two runs each on one box: 8 ranks (the rule above gives 3): C = 3 / 6 / 8 / 12 / 16 -> 2.75 / 2.74 / 2.64 / 2.70 /
  // 2.67 ms; 4 ranks (rule: 7): 7 / 8 / 16 -> 5.15 / 5.32 / 5.26; 2 ranks (rule: 15): 8 / 15 / 16 -> 10.25 / 10.53 / 10.68.  So: at
  // most 8 tiles per chunk, and 8 instead of anything below 7 while that still leaves >= 4 units per resident workgroup slot.
  if (sw > 1 && i_step > 1) {
    if (c > 8) c = 8;
    else if (c < 7 && ((long)(rowsI / sw) * (long)((t + 7) / 8)) / 2 >= 4L * 3 * (n_cu > 0 ? n_cu : 256)) c = 8;
  }
  if (ni == 2 && sw == 1 && c > 1 && (c & 1)) ++c;      // (the wave-unit kernel's closed-form unit index wants C = 1 or even there)
  if (tune.chunk > 0) c = tune.chunk;
  L.Npad = (long)t * TS; L.T = t; L.NI = ni; L.C = c; L.nch = (t + c - 1) / c; L.rowsI = rowsI;
  L.SW = sw;
  L.rowsG = (rowsI + L.SW - 1) / L.SW; L.tri = (i_step == 1) ? 1 : 0; L.nrhs = nrhs; L.i_first = i_first; L.i_step = i_step;
  return L;
}

static size_t sym_slabI_blobs(const SymLayout &L) { return (size_t)(sym_offI(L, L.nch - 1) + sym_HI(L, L.nch - 1)); }
static size_t sym_slabJ_blobs(const SymLayout &L)
{
  return (size_t)(sym_offJ(L, L.rowsG - 1) + (L.Npad - sym_RJ(L, L.rowsG - 1)));
}

size_t rbl_apply_M_sym_bytes(int64_t n_blobs, int n_cu, int i_step, int nrhs, const RblSymTune &tune, int *NI_out, int *C_out)
{
  const SymLayout L = sym_geometry(n_blobs, n_cu, 0, i_step, nrhs, tune);
  if (NI_out) *NI_out = L.NI;
  if (C_out) *C_out = L.C;
  // slabs + tile bounding boxes + far map (one byte per (row super-tile, tile))
  return ((sym_slabI_blobs(L) + sym_slabJ_blobs(L)) * 3 * nrhs + (size_t)L.T * 6) * sizeof(double) +
         (size_t)((L.T + L.NI - 1) / L.NI) * (size_t)L.T + 64 + 128;      // (+ the work-queue counter, 64-byte aligned, behind the far map)
}

template <bool WALL, int NI, int SW>
static void launch_sym(hipStream_t st, const RblParams &P, const double *d_F, const double *d_r, int64_t n_blobs,
                       double *d_out, double *slabI, double *slabJ, const SymLayout &L, unsigned *d_err, bool relaxed,
                       int n_cu, bool use_queue, double gap_ratio, const RblSaddleFuse &fuse)
{
  const int T = L.T, nrhs = L.nrhs;
  dim3 grid((unsigned)L.rowsG, (unsigned)L.nch), block(TS * SW);
  const int64_t n = 3 * n_blobs;
  dim3 g2((unsigned)((n + 63) / 64), (unsigned)nrhs), b2(64 * RG);
  unsigned char *farmap = nullptr;
  unsigned *queue = nullptr;
  if (NI == 2) {   // large systems only: two more tiny launches, then most tile pairs skip the overlap test
    double *bbox = slabJ + sym_slabJ_blobs(L) * 3 * nrhs;
    farmap = (unsigned char *)(bbox + (size_t)T * 6);
    const int nsup = (T + NI - 1) / NI;
    // work queue (multi-wave workgroups = large systems): its counter sits behind the far map and is zeroed by k_tile_far
    if (use_queue && SW > 1) {
      queue = (unsigned *)(((uintptr_t)(farmap + (size_t)nsup * (size_t)T) + 63) & ~(uintptr_t)63);
      // more workgroups than can be resident do no harm (late ones find the queue empty); fewer would leave CUs idle
      const unsigned want = (unsigned)(n_cu > 0 ? n_cu : 256) * 4u, have = (unsigned)L.rowsG * (unsigned)L.nch;
      grid = dim3(want < have ? want : have, 1);
    }
    hipLaunchKernelGGL(k_tile_bbox, dim3((unsigned)T), dim3(TS), 0, st, d_r, (long)n_blobs, P.inv_a, bbox);
    hipLaunchKernelGGL(k_tile_far, dim3((unsigned)((T + 255) / 256), (unsigned)nsup), dim3(256), 0, st,
                       (const double *)bbox, T, NI, farmap, queue, gap_ratio);
  }
  constexpr int PR = (NI == 2) ? 1 : 0;      // the relaxed form exists for two rows per lane
  if (nrhs == 2 && relaxed && NI == 2)
    hipLaunchKernelGGL((k_apply_M_sym2<WALL, NI, SW, PR>), grid, block, 0, st, d_r, d_F, slabI, slabJ, (long)n_blobs, L, P,
                       d_err, (const unsigned char *)farmap, queue);
  else if (nrhs == 2)
    hipLaunchKernelGGL((k_apply_M_sym2<WALL, NI, SW, 0>), grid, block, 0, st, d_r, d_F, slabI, slabJ, (long)n_blobs, L, P,
                       d_err, (const unsigned char *)farmap, queue);
  else if (relaxed && NI == 2)
    hipLaunchKernelGGL((k_apply_M_sym<WALL, NI, SW, PR>), grid, block, 0, st, d_r, d_F, slabI, slabJ, (long)n_blobs, L, P,
                       d_err, (const unsigned char *)farmap, queue);
  else
    hipLaunchKernelGGL((k_apply_M_sym<WALL, NI, SW, 0>), grid, block, 0, st, d_r, d_F, slabI, slabJ, (long)n_blobs, L, P,
                       d_err, (const unsigned char *)farmap, queue);
  hipLaunchKernelGGL(k_reduce_sym<WALL>, g2, b2, 0, st, slabI, slabJ, d_r, d_out, (long)n_blobs, L, P, d_err, nrhs == 1 ? fuse : RblSaddleFuse());
}

// ---- which instantiation a symmetric product launches: ONE table ---------------------------------------------------------------
// A row = one kernel family in one shape (rows per lane NI, waves per workgroup SW) with what it can do; sym_pick returns the FIRST
// row whose shape equals the layout's and whose conditions hold, or nullptr -- in which case nothing is launched and the caller
// reports RBL_ERR_ARG (round 4's cascade of `if`s ended in an `else` that launched <1, 1> for ANY layout it did not know: a forced
// option once ran the wrong shape and returned a wrong product, profiles/r04_midrange_final.txt).  The names are those of
// librbl.isa.json (bench.py prices a product with its own kernel's instruction count).
namespace {

struct SymArgs {
  hipStream_t st; const RblParams *P; bool wall; const double *d_F, *d_r; int64_t n_blobs; double *d_out, *slabI, *slabJ;
  const SymLayout *L; unsigned *d_err; bool relaxed; int n_cu; bool use_queue; double gap_ratio; const RblSaddleFuse *fuse;
};

template <int NI, int SW> void row_launch_sym(const SymArgs &a)
{
  if (a.wall) launch_sym<true, NI, SW>(a.st, *a.P, a.d_F, a.d_r, a.n_blobs, a.d_out, a.slabI, a.slabJ, *a.L, a.d_err, a.relaxed, a.n_cu, a.use_queue, a.gap_ratio, *a.fuse);
  else launch_sym<false, NI, SW>(a.st, *a.P, a.d_F, a.d_r, a.n_blobs, a.d_out, a.slabI, a.slabJ, *a.L, a.d_err, a.relaxed, a.n_cu, a.use_queue, a.gap_ratio, *a.fuse);
}

// wave-owned units (k_apply_M_symw / k_apply_M_symw2v): same slabs, same reduction
template <int NI, int NRHS> void row_launch_symw(const SymArgs &a)
{
  constexpr int IW = RBL_SYMW_IW;
  const SymLayout &L = *a.L;
  const long n_units = L.tri ? (NI == 2 || NRHS == 2 ? symw_prefix_ni(L.rowsI, L.C, L.nch, NI) : symw_prefix(L.rowsI, L.C, L.nch)) : (long)L.rowsI * L.nch;
  const dim3 grid((unsigned)((n_units + IW - 1) / IW)), block(TS * IW);
  if (NRHS == 2) {
    if (a.wall) hipLaunchKernelGGL((k_apply_M_symw2v<true, NI, IW>), grid, block, 0, a.st, a.d_r, a.d_F, a.slabI, a.slabJ, (long)a.n_blobs, L, *a.P, a.d_err, n_units);
    else hipLaunchKernelGGL((k_apply_M_symw2v<false, NI, IW>), grid, block, 0, a.st, a.d_r, a.d_F, a.slabI, a.slabJ, (long)a.n_blobs, L, *a.P, a.d_err, n_units);
  } else {
    if (a.wall) hipLaunchKernelGGL((k_apply_M_symw<true, NI, IW>), grid, block, 0, a.st, a.d_r, a.d_F, a.slabI, a.slabJ, (long)a.n_blobs, L, *a.P, a.d_err, n_units);
    else hipLaunchKernelGGL((k_apply_M_symw<false, NI, IW>), grid, block, 0, a.st, a.d_r, a.d_F, a.slabI, a.slabJ, (long)a.n_blobs, L, *a.P, a.d_err, n_units);
  }
  const int64_t n = 3 * a.n_blobs;
  dim3 g2((unsigned)((n + 63) / 64), (unsigned)NRHS), b2(64 * RG);
  const RblSaddleFuse fuse = NRHS == 1 ? *a.fuse : RblSaddleFuse();
  if (a.wall) hipLaunchKernelGGL(k_reduce_sym<true>, g2, b2, 0, a.st, a.slabI, a.slabJ, a.d_r, a.d_out, (long)a.n_blobs, L, *a.P, a.d_err, fuse);
  else hipLaunchKernelGGL(k_reduce_sym<false>, g2, b2, 0, a.st, a.slabI, a.slabJ, a.d_r, a.d_out, (long)a.n_blobs, L, *a.P, a.d_err, fuse);
}

struct SymRow {
  const char *name;        // printf format of the librbl.isa.json name: %s = wall
  int NI, SW;              // the shape the instantiation was compiled for: must EQUAL the layout's
  int nrhs;                // 1, 2, or 0 = both
  bool wave_units;         // a wave-unit kernel: needs RBL_OPT_SYM_WAVE_UNITS on, no relaxed sweep, and (NI == 2 or two vectors) C = 1 or even
  bool relaxed_form;       // has a packed-single-precision form (two rows per lane); others run fp64 when relaxation is asked for
  void (*launch)(const SymArgs &);
};

const SymRow kSymRows[] = {
    {"k_apply_M_symw2v<%s,2>", 2, 1, 2, true, false, row_launch_symw<2, 2>},
    {"k_apply_M_symw2v<%s,1>", 1, 1, 2, true, false, row_launch_symw<1, 2>},
    {"k_apply_M_symw<%s,2>", 2, 1, 1, true, false, row_launch_symw<2, 1>},
    {"k_apply_M_symw<%s>", 1, 1, 1, true, false, row_launch_symw<1, 1>},
    {"k_apply_M_sym%s<%s,2,4>", 2, SW_LARGE, 0, false, true, row_launch_sym<2, SW_LARGE>},
    {"k_apply_M_sym%s<%s,2,1>", 2, 1, 0, false, true, row_launch_sym<2, 1>},
    {"k_apply_M_sym%s<%s,1,1>", 1, 1, 0, false, false, row_launch_sym<1, 1>},
#ifdef RBL_WAVE_TRACE
    {"k_apply_M_sym%s<%s,1,2>", 1, 2, 0, false, false, row_launch_sym<1, 2>},     // experiment (tools/wave_trace.hip)
#endif
};

const SymRow *sym_pick(const SymLayout &L, int nrhs, bool relaxed, const RblSymTune &tune)
{
  for (const SymRow &r : kSymRows) {
    if (r.NI != L.NI || r.SW != L.SW) continue;
    if (r.nrhs && r.nrhs != nrhs) continue;
    if (r.wave_units) {
      if (tune.wave_units < 0 || relaxed) continue;
      if ((r.NI == 2 || r.nrhs == 2) && !(L.NI == 1 || L.C == 1 || (L.C & 1) == 0)) continue;   // closed-form unit index: C = 1 or even with two rows per lane
    }
    return &r;
  }
  return nullptr;
}

// a forced option the geometry could not honour is an error, not a silent fall-back
bool sym_forced_ok(const SymLayout &L, int nrhs, const RblSymTune &tune)
{
  if (tune.sw > 0 && L.SW != tune.sw) return false;                     // e.g. sym_waves = 4 with one row per lane
  if (nrhs == 1 && tune.ni1 > 0 && L.NI != tune.ni1) return false;
  if (nrhs == 2 && tune.ni2 > 0 && L.NI != tune.ni2) return false;
  if (tune.chunk > 0 && L.C != tune.chunk) return false;
  return true;
}

}  // namespace

// nrhs = 1 or 2 force vectors (d_F, d_out: [nrhs][3 n_blobs]); d_work from rbl_apply_M_sym_bytes(...).
// RBL_ERR_ARG (nothing launched): the options force a shape no instantiation has.
int rbl_launch_apply_M_sym(hipStream_t st, const RblParams &P, bool wall, const double *d_F,
                           const double *d_r, int64_t n_blobs, int i_first, int i_step,
                           double *d_out, double *d_work, int n_cu, unsigned *d_err, int nrhs, const RblSymTune &tune)
{
  if (n_blobs <= 0) return RBL_OK;
  const SymLayout L = sym_geometry(n_blobs, n_cu, i_first, i_step, nrhs, tune);
  const bool relaxed = tune.relaxed != 0;
  const SymRow *row = sym_forced_ok(L, nrhs, tune) ? sym_pick(L, nrhs, relaxed, tune) : nullptr;
  if (!row) return RBL_ERR_ARG;
  SymArgs a;
  a.st = st; a.P = &P; a.wall = wall; a.d_F = d_F; a.d_r = d_r; a.n_blobs = n_blobs; a.d_out = d_out;
  a.slabI = d_work; a.slabJ = d_work + sym_slabI_blobs(L) * 3 * nrhs;
  a.L = &L; a.d_err = d_err; a.relaxed = relaxed && row->relaxed_form; a.n_cu = n_cu; a.use_queue = tune.queue >= 0;
  a.gap_ratio = tune.gap_ratio > 0 ? (double)tune.gap_ratio : 15.0; a.fuse = &tune.fuse;
  row->launch(a);
  return RBL_OK;
}

// the instantiation rbl_launch_apply_M_sym would launch (reporting: bench.py names what it times): the same table, no launch;
// an empty name when the options force a shape no instantiation has
void rbl_apply_M_sym_kernel_name(int64_t n_blobs, int n_cu, int i_step, int nrhs, const RblSymTune &tune, bool wall, char *out, size_t len)
{
  const SymLayout L = sym_geometry(n_blobs, n_cu, 0, i_step, nrhs, tune);
  const char *w = wall ? "true" : "false";
  const SymRow *row = sym_forced_ok(L, nrhs, tune) ? sym_pick(L, nrhs, tune.relaxed != 0, tune) : nullptr;
  if (len) out[0] = 0;
  if (!row) return;
  if (row->wave_units) { std::snprintf(out, len, row->name, w); return; }
  // k_apply_M_sym / k_apply_M_sym2: librbl.isa.json names them <wall, NI> (one or four waves per workgroup: the same sweep)
  if (nrhs == 2) std::snprintf(out, len, "k_apply_M_sym2<%s,%d,%d>", w, L.NI, L.SW);
  else std::snprintf(out, len, "k_apply_M_sym<%s,%d>", w, L.NI);
}

// ---- multi-RHS (MFMA) variant ---------------------------------------------------
static void mrhs_geometry(int64_t n_blobs, int n_cu, int64_t *Npad, int *nsplit, int64_t *jchunk)
{
  const int64_t np = ((n_blobs + MTJ - 1) / MTJ) * MTJ;
  const int64_t iblocks = (n_blobs + 63) / 64;
  const int64_t jtiles = np / MTJ;
  int64_t js = ((int64_t)(n_cu > 0 ? n_cu : 256) * 4 * 4 + iblocks - 1) / iblocks;  // ~4 rounds of 4 blocks/CU
  if (js > jtiles) js = jtiles;
  if (js < 1) js = 1;
  const int64_t jc = ((jtiles + js - 1) / js) * MTJ;
  *Npad = np; *jchunk = jc; *nsplit = (int)((np + jc - 1) / jc);
}

size_t rbl_apply_M_mrhs_bytes(int64_t n_blobs, int n_cu)
{
  int64_t np, jc; int ns;
  mrhs_geometry(n_blobs, n_cu, &np, &ns, &jc);
  return (size_t)np * 3 * MR * sizeof(double) * (size_t)(1 + ns);
}

// d_F, d_out: column-major n3 x nrhs (nrhs <= 16).  d_work from rbl_apply_M_mrhs_bytes.
// ldF / ldO: doubles between consecutive right-hand sides / results (0: packed, 3 n_blobs)
void rbl_launch_apply_M_mrhs(hipStream_t st, const RblParams &P, bool wall, const double *d_F,
                             const double *d_r, int64_t n_blobs, int nrhs, double *d_out,
                             double *d_work, int n_cu, unsigned *d_err, int64_t ldF, int64_t ldO)
{
  if (n_blobs <= 0 || nrhs <= 0) return;
  if (ldF <= 0) ldF = 3 * n_blobs;
  if (ldO <= 0) ldO = 3 * n_blobs;
  int64_t np, jc; int ns;
  mrhs_geometry(n_blobs, n_cu, &np, &ns, &jc);
  double *Fp = d_work, *Up = d_work + (size_t)np * 3 * MR;
  const int64_t npk = np * 3 * MR, nun = 3 * n_blobs * nrhs;
  dim3 gp((unsigned)((npk + 255) / 256)), gu((unsigned)((nun + 255) / 256)), b(256);
  dim3 grid((unsigned)((n_blobs + 63) / 64), (unsigned)ns);
  if (wall) {
    hipLaunchKernelGGL(k_pack_rhs<true>, gp, b, 0, st, d_F, d_r, (long)n_blobs, (long)np, nrhs, P, Fp, (long)ldF);
    hipLaunchKernelGGL(k_apply_M_mrhs<true>, grid, b, 0, st, d_r, Fp, Up, (long)n_blobs, (long)np, (long)jc, P, d_err);
    hipLaunchKernelGGL(k_unpack_rhs<true>, gu, b, 0, st, Up, d_r, (long)n_blobs, (long)np, nrhs, ns, P, d_out, d_err, (long)ldO);
  } else {
    hipLaunchKernelGGL(k_pack_rhs<false>, gp, b, 0, st, d_F, d_r, (long)n_blobs, (long)np, nrhs, P, Fp, (long)ldF);
    hipLaunchKernelGGL(k_apply_M_mrhs<false>, grid, b, 0, st, d_r, Fp, Up, (long)n_blobs, (long)np, (long)jc, P, d_err);
    hipLaunchKernelGGL(k_unpack_rhs<false>, gu, b, 0, st, Up, d_r, (long)n_blobs, (long)np, nrhs, ns, P, d_out, d_err, (long)ldO);
  }
}

void rbl_launch_blob_positions(hipStream_t st, const double *d_X, const double *d_Q,
                               const double *d_cfg, int N_blb, int body_begin, int body_end,
                               double *d_out)
{
  const long n = (long)(body_end - body_begin) * N_blb;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_blob_positions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_X,
                     d_Q, d_cfg, N_blb, body_begin, n, d_out);
}

void rbl_launch_build_M(hipStream_t st, const RblParams &P, bool wall, bool scale_damp,
                        const double *d_r, int64_t n_blobs, double *d_M, unsigned *d_err)
{
  if (n_blobs <= 0) return;
  dim3 grid((unsigned)((n_blobs + 255) / 256), (unsigned)((n_blobs + JB - 1) / JB)), block(256);
  if (wall)
    hipLaunchKernelGGL(k_build_M<true>, grid, block, 0, st, d_r, d_M, (long)n_blobs,
                       scale_damp ? 1 : 0, P, d_err, 0L, 0L, 0);
  else
    hipLaunchKernelGGL(k_build_M<false>, grid, block, 0, st, d_r, d_M, (long)n_blobs,
                       scale_damp ? 1 : 0, P, d_err, 0L, 0L, 0);
}

// `batch` independent blob sets of n_blobs each (one rigid body each), matrices strideM apart
// lower_tiles: 0 = the whole matrices; t > 0 = only what a factorisation in tiles of t x t reads (the lower triangle and whole
// diagonal tiles: blocks entirely above them stay unwritten)
void rbl_launch_build_M_batched(hipStream_t st, const RblParams &P, bool wall, const double *d_r,
                                int64_t n_blobs, int batch, double *d_M, int64_t strideM, unsigned *d_err, int lower_tiles)
{
  if (n_blobs <= 0 || batch <= 0) return;
  dim3 grid((unsigned)((n_blobs + 255) / 256), (unsigned)((n_blobs + JB - 1) / JB), (unsigned)batch), block(256);
  if (wall)
    hipLaunchKernelGGL(k_build_M<true>, grid, block, 0, st, d_r, d_M, (long)n_blobs, 0, P, d_err,
                       (long)(3 * n_blobs), (long)strideM, lower_tiles);
  else
    hipLaunchKernelGGL(k_build_M<false>, grid, block, 0, st, d_r, d_M, (long)n_blobs, 0, P, d_err,
                       (long)(3 * n_blobs), (long)strideM, lower_tiles);
}

void rbl_launch_pair_blocks(hipStream_t st, const RblParams &P, bool wall, int mode,
                            const double *d_ri, const double *d_rj, const int32_t *d_ii,
                            const int32_t *d_jj, int64_t n, double *d_out9, unsigned *d_err)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_pair_blocks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_ri, d_rj,
                     d_ii, d_jj, (long)n, wall ? 1 : 0, mode, P, d_out9, d_err);
}

void rbl_launch_normal(hipStream_t st, uint64_t seed, uint64_t offset, int64_t n, double *d_out)
{
  if (n <= 0) return;
  const long pairs = (n + 1) / 2;
  hipLaunchKernelGGL(k_normal, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, seed,
                     offset, (long)n, d_out);
}

// d_out2 must have room for 2 + 2*DOT_BLOCKS doubles: [0..1] result, rest scratch
static constexpr int DOT_BLOCKS = 512;
void rbl_launch_dot2(hipStream_t st, const double *x, const double *y, const double *z,
                     int64_t n, double *d_out2)
{
  int nblk = (int)std::min<int64_t>(DOT_BLOCKS, (n + 255) / 256);
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(k_dot2_partial, dim3(nblk), dim3(256), 0, st, x, y, z, (long)n, d_out2 + 2);
  hipLaunchKernelGGL(k_dot2_final, dim3(1), dim3(256), 0, st, d_out2 + 2, nblk, d_out2);
}

void rbl_launch_axpby(hipStream_t st, int64_t n, double a, const double *x, double b,
                      const double *y, double *out)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_axpby, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long)n, a, x,
                     b, y, out);
}

void rbl_launch_rhs_combine(hipStream_t st, int64_t n, const double *x, double a, const double *y, double b, const double *z,
                            const double *w, double *out)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rhs_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long)n, x, a, y, b, z, w, out);
}

static int lz_grid(int64_t n)
{
  int g = (int)std::min<int64_t>(LZ_BLOCKS, (n + 255) / 256);
  return g < 1 ? 1 : g;
}

size_t rbl_lanczos_part_doubles(void) { return 4 * (size_t)LZ_BLOCKS; }      // two partial-sum arrays for each vector of a pair

// V0 = W / |W|, *wnorm_out = |W|
// (nvec vectors in the same two launches: vector k of d_W / V0 is k n doubles further, its norm k * scal_stride)
void rbl_launch_lanczos_init(hipStream_t st, int64_t n, const double *d_W, double *wnorm_out, double *V0,
                             double *part, int nvec, int64_t scal_stride)
{
  const int g = lz_grid(n);
  const long vs = nvec > 1 ? (long)n : 0L, ss = nvec > 1 ? (long)scal_stride : 0L, ps = nvec > 1 ? 2L * LZ_BLOCKS : 0L;
  hipLaunchKernelGGL(k_lz_a, dim3(g, nvec), dim3(256), 0, st, (long)n, const_cast<double *>(d_W), d_W,
                     (const double *)nullptr, (const double *)nullptr, part, vs, ss, ps);
  hipLaunchKernelGGL(k_lz_c, dim3(g, nvec), dim3(256), 0, st, (long)n, d_W, (const double *)part, g, wnorm_out, V0, vs, ss, ps,
                     (const double *)nullptr, 0L, 0L);
}

// one Lanczos step after u = A v:  u -= beta_prev vprev; alpha = v.u; u -= alpha v; beta = |u|; vnext = u/beta
// nvec (1 or 2) recurrences in lock step, one launch each: vector k of u / v / vprev / vnext is k * vec_stride doubles
// further, its scalars (beta_prev, alpha_out, beta_out) k * scal_stride
void rbl_launch_lanczos_step(hipStream_t st, int64_t n, double *u, const double *v, const double *vprev,
                             const double *beta_prev, double *alpha_out, double *beta_out, double *vnext,
                             double *part, int nvec, int64_t vec_stride, int64_t scal_stride)
{
  const int g = lz_grid(n);
  double *pA = part, *pB = part + LZ_BLOCKS;
  const long vs = (long)vec_stride, ss = (long)scal_stride, ps = 2L * LZ_BLOCKS;
  hipLaunchKernelGGL(k_lz_a, dim3(g, nvec), dim3(256), 0, st, (long)n, u, v, vprev, beta_prev, pA, vs, ss, ps);
  hipLaunchKernelGGL(k_lz_b, dim3(g, nvec), dim3(256), 0, st, (long)n, u, v, (const double *)pA, g, alpha_out, pB, vs, ss, ps);
  hipLaunchKernelGGL(k_lz_c, dim3(g, nvec), dim3(256), 0, st, (long)n, (const double *)u, (const double *)pB, g,
                     beta_out, vnext, vs, ss, ps, (const double *)nullptr, 0L, 0L);
}

// out = sum_p coef[p] V_p, consecutive basis vectors `stride` doubles apart (0: contiguous, = n)
void rbl_launch_lanczos_combine(hipStream_t st, int64_t n, const double *V, const double *coef, int m,
                                double *out, int64_t stride)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_lz_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long)n, V, coef, m, out,
                     (long)(stride > 0 ? stride : n));
}

void rbl_launch_lanczos_combine_xd(hipStream_t st, int64_t n, const double *V, int64_t stride, int64_t vsep, const double *coef,
                                   int64_t csep, int m, double *zx, double *zd, int64_t osep, int nvec)
{
  if (n <= 0 || nvec <= 0) return;
  hipLaunchKernelGGL(k_lz_combine_xd, dim3((unsigned)((n + 255) / 256), nvec), dim3(256), 0, st, (long)n, V, (long)stride, (long)vsep, coef,
                     (long)csep, m, zx, zd, (long)osep);
}

// returns the number of partial sums per vector (part: nv x that many doubles, at most RBL_SQNORM_BLOCKS each)
int rbl_launch_damp_sqnorm(hipStream_t st, const RblParams &P, const double *d_r, int64_t n_blobs, double *o, int64_t pitch, int nv,
                           double *part)
{
  if (n_blobs <= 0 || nv <= 0) return 0;
  int g = (int)std::min<int64_t>(RBL_SQNORM_BLOCKS, (3 * n_blobs + 255) / 256);
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_damp_sqnorm, dim3(g, nv), dim3(256), 0, st, P, d_r, (long)n_blobs, o, (long)pitch, part);
  return g;
}

int rbl_gmres_max_vectors(void) { return GM_MAXK; }
constexpr int AR_BLOCKS = 1024;
constexpr int AR_P1 = 512;          // partial sums per vector of the first pass: 128 from k_mdot_partial, up to this many from a fused product (k_arnoldi_upd: 8 x 64)
int rbl_gmres_p1_capacity(void) { return AR_P1; }
size_t rbl_gmres_part_doubles(void) { return (size_t)GM_MAXK * AR_P1 + (size_t)GM_MAXK * AR_BLOCKS + AR_BLOCKS; }

// one Arnoldi step after w = A P^-1 v_j: classical Gram-Schmidt twice against V[0..k), then Hcol[k] = |w| and
// vnext = w / |w|  (four launches, see k_arnoldi_upd)
void rbl_launch_arnoldi_step(hipStream_t st, const double *V, int64_t n, int k, double *w, double *Hcol, double *vnext,
                             double *part, int fused_np, bool skip_norm, const double **norm_part, int *norm_np)
{
  if (k <= 0 || n <= 0) return;
  int nb = (int)std::min<int64_t>(128, (n + 1023) / 1024);
  if (nb < 1) nb = 1;
  const int64_t gneed = (n + 256 * AR_EPT - 1) / (256 * AR_EPT);          // AR_EPT entries per thread ...
  int g = (int)std::max<int64_t>(gneed, std::min<int64_t>(256, (n + 255) / 256));    // ... fewer in small systems: more blocks
  if (g > AR_BLOCKS) g = AR_BLOCKS;                                        // beyond 1 048 576 entries the kernels take several passes
  double *p1 = part, *p2 = part + (size_t)GM_MAXK * AR_P1, *pn = p2 + (size_t)GM_MAXK * AR_BLOCKS;
  if (fused_np > 0) nb = fused_np;                     // (the product's slab reduction left part[k][fused_np]: RblSaddleFuse)
  else hipLaunchKernelGGL(k_mdot_partial, dim3(nb, k), dim3(256), 0, st, V, (long)n, (const double *)w, p1, (long)n, 0L, 0L, 0L);
  hipLaunchKernelGGL(k_arnoldi_upd<false>, dim3(g), dim3(256), 0, st, V, (long)n, k, w, (const double *)p1, nb, Hcol, p2, (long)n, 0L,
                     0L, 0L, 0L);
  hipLaunchKernelGGL(k_arnoldi_upd<true>, dim3(g), dim3(256), 0, st, V, (long)n, k, w, (const double *)p2, g, Hcol, pn, (long)n, 0L, 0L,
                     0L, 0L);
  if (norm_part) *norm_part = pn;
  if (norm_np) *norm_np = g;
  if (skip_norm) return;                               // (the consumer normalises: RblNormFold)
  hipLaunchKernelGGL(k_lz_c, dim3(lz_grid(n)), dim3(256), 0, st, (long)n, (const double *)w, (const double *)pn, g, Hcol + k,
                     vnext, 0L, 0L, 0L, (const double *)nullptr, 0L, 0L);
}

// One step of the Lanczos recurrence WITH full re-orthogonalisation, for nvec (1 or 2) recurrences in lock step: u_p is made
// orthogonal (classical Gram-Schmidt twice) to ALL k earlier vectors of its own basis -- for a symmetric operator the
// Arnoldi process, whose Hessenberg column is the tridiagonal one up to rounding: alpha = h[k-1], beta = |u|.  Basis vector
// j of recurrence p at V + (j nvec + p) n (the interleaved layout of the two-vector product), u_p at u + p n;
// alpha_out / beta_out of recurrence p are scal_stride doubles apart; hcol: nvec x hcol_stride (>= k + 1) scratch;
// part: nvec x rbl_gmres_part_doubles().  The three-term recurrence alone loses orthogonality once a Ritz value has
// converged, its estimate of M^{1/2} W then stagnates near 1e-6 (round 2's red run); the basis is kept for the final
// combination anyway, and O(k n) vector work is nothing beside an O(N^2) product.
void rbl_launch_lanczos_step_reorth(hipStream_t st, int64_t n, int k, double *u, const double *V, double *vnext, double *alpha_out,
                                    double *beta_out, int64_t scal_stride, double *hcol, int64_t hcol_stride, double *part, int nvec)
{
  if (k <= 0 || n <= 0 || nvec < 1) return;
  int nb = (int)std::min<int64_t>(128, (n + 1023) / 1024);
  if (nb < 1) nb = 1;
  const int64_t gneed = (n + 256 * AR_EPT - 1) / (256 * AR_EPT);
  int g = (int)std::max<int64_t>(gneed, std::min<int64_t>(256, (n + 255) / 256));
  if (g > AR_BLOCKS) g = AR_BLOCKS;
  const long pp = (long)rbl_gmres_part_doubles(), ph = (long)hcol_stride, vstr = (long)nvec * (long)n, pn_ = (long)n;
  double *p1 = part, *p2 = part + (size_t)GM_MAXK * AR_P1, *pn = p2 + (size_t)GM_MAXK * AR_BLOCKS;
  // more than GM_MAXK basis vectors: one group of GM_MAXK after the other (classical Gram-Schmidt twice inside a group,
  // the groups in sequence); the norm partials of the last group's second pass belong to the final vector
  for (int c0 = 0; c0 < k; c0 += GM_MAXK) {
    const int kc = k - c0 < GM_MAXK ? k - c0 : GM_MAXK;
    const double *Vc = V + (size_t)c0 * (size_t)vstr;
    double *hc = hcol + c0;
    hipLaunchKernelGGL(k_mdot_partial, dim3(nb, kc, nvec), dim3(256), 0, st, Vc, (long)n, (const double *)u, p1, vstr, pn_, pn_, pp);
    hipLaunchKernelGGL(k_arnoldi_upd<false>, dim3(g, nvec), dim3(256), 0, st, Vc, (long)n, kc, u, (const double *)p1, nb, hc, p2, vstr,
                       pn_, pn_, pp, ph);
    hipLaunchKernelGGL(k_arnoldi_upd<true>, dim3(g, nvec), dim3(256), 0, st, Vc, (long)n, kc, u, (const double *)p2, g, hc, pn, vstr,
                       pn_, pn_, pp, ph);
  }
  hipLaunchKernelGGL(k_lz_c, dim3(lz_grid(n), nvec), dim3(256), 0, st, (long)n, (const double *)u, (const double *)pn, g, beta_out,
                     vnext, pn_, (long)scal_stride, pp, (const double *)(hcol + (k - 1)), ph, (long)(alpha_out - beta_out));
}

void rbl_launch_scale_by_damp(hipStream_t st, const RblParams &P, const double *d_r,
                              int64_t n_blobs, const double *in, double *out)
{
  const int64_t n = 3 * n_blobs;
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scale_by_damp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, d_r,
                     (long)n_blobs, in, out);
}
