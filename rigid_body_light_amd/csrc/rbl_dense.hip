// rbl_dense.hip -- dense fp64 kernels behind M_half_W (reference c_rigid_obj.cpp:661-675):
// in-place blocked lower Cholesky (what Eigen::LLT computes, :670-671) whose
// trailing updates run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), and the
// triangular product L W (:672).
//
// Storage: column-major n x n, ld = n, 64-bit indexing (cfg 5 has 2.36e10 entries).
//
// Blocking: outer panels of NB columns; inside a panel, IB-wide steps of
//   potf2 (one workgroup, LDS) -> trsm (row per lane, L_kk in LDS) -> rank-IB update
//   of the rest of the panel; then ONE rank-NB MFMA update of the trailing matrix.
// Only the lower triangle is referenced/updated (tiles strictly above the diagonal
// are skipped; diagonal tiles are updated whole).
#include "rbl_internal.hpp"

namespace {

constexpr int IB = 32;    // inner step width
constexpr int NB = 256;   // outer panel width (K of the trailing MFMA update)

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- potf2: factor the IB x IB diagonal block at (k,k) -------------------------
__global__ __launch_bounds__(256) void k_potf2(double *__restrict__ A, long n, long k, int nb,
                                               unsigned *err)
{
  __shared__ double s[IB][IB + 1];
  const int t = threadIdx.x;
  for (int e = t; e < IB * IB; e += 256) {
    const int i = e % IB, j = e / IB;
    s[i][j] = (i < nb && j < nb) ? A[(size_t)(k + j) * n + (k + i)] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int c = 0; c < nb; ++c) {
    const double d = s[c][c];
    if (!(d > 0.0)) {
      if (t == 0) atomicOr(err, (unsigned)RBL_FLAG_NOT_SPD);
      return;  // uniform: every thread reads the same s[c][c]
    }
    const double sd = sqrt(d);
    __syncthreads();
    if (t == 0) s[c][c] = sd;
    if (t > c && t < nb) s[t][c] = s[t][c] / sd;
    __syncthreads();
    for (int e = t; e < IB * IB; e += 256) {
      const int i = e % IB, j = e / IB;
      if (j > c && i >= j && i < nb) s[i][j] -= s[i][c] * s[j][c];
    }
    __syncthreads();
  }
  for (int e = t; e < IB * IB; e += 256) {
    const int i = e % IB, j = e / IB;
    if (i < nb && j < nb && i >= j) A[(size_t)(k + j) * n + (k + i)] = s[i][j];
  }
}

// ---- trsm: rows below the diagonal block:  X <- X L_kk^{-T} ----------------------
__global__ __launch_bounds__(256) void k_trsm(double *__restrict__ A, long n, long k, int nb)
{
  __shared__ double L[IB][IB + 1];
  const int t = threadIdx.x;
  for (int e = t; e < IB * IB; e += 256) {
    const int i = e % IB, j = e / IB;
    L[i][j] = (i < nb && j < nb && i >= j) ? A[(size_t)(k + j) * n + (k + i)] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  const long row = k + nb + (long)blockIdx.x * 256 + t;
  if (row >= n) return;
  double x[IB];
#pragma unroll
  for (int c = 0; c < IB; ++c) x[c] = (c < nb) ? A[(size_t)(k + c) * n + row] : 0.0;
#pragma unroll
  for (int c = 0; c < IB; ++c) {
    double v = x[c];
#pragma unroll
    for (int m = 0; m < c; ++m) v = __builtin_fma(-x[m], L[c][m], v);
    x[c] = v / L[c][c];
  }
#pragma unroll
  for (int c = 0; c < IB; ++c)
    if (c < nb) A[(size_t)(k + c) * n + row] = x[c];
}

// ---- rank-K update on the matrix cores -----------------------------------------
//   C[i][j] -= sum_{k in [k0,k0+K)} A[i][k] A[j][k]     i in [r0,n), j in [r0,c1), i-tile >= j-tile
// Workgroup = 4 waves (2x2), 128x128 tile; wave = 64x64 = 4x4 MFMA tiles of 16x16.
// The MFMA computes the TRANSPOSED tile (A-operand from the j rows, B-operand from the
// i rows) so that, in the C/D layout row=(lane>>4)+4v, col=lane&15, a lane's 16-lane
// group touches 16 consecutive ROWS of one column of C: 128-B runs in column-major C.
__global__ __launch_bounds__(256) void k_syrk_mfma(double *__restrict__ A, long n, long r0,
                                                   long c1, long k0, int K)
{
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bi < bj) return;  // strictly-upper tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long i0 = r0 + (long)bi * 128 + (wave & 1) * 64;
  const long j0 = r0 + (long)bj * 128 + (wave >> 1) * 64;
  if (i0 >= n || j0 >= c1) return;
  if (i0 + 63 < j0) return;  // wave tile strictly above the diagonal
  const int l15 = lane & 15, l4 = lane >> 4;

  double4_t acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  long irow[4], jrow[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    long ir = i0 + 16 * q + l15; irow[q] = ir < n ? ir : n - 1;
    long jr = j0 + 16 * q + l15; jrow[q] = jr < n ? jr : n - 1;
  }
  const double *base = A + (size_t)(k0 + l4) * (size_t)n;
  for (int kk = 0; kk < K; kk += 4) {
    const double *colp = base + (size_t)kk * (size_t)n;
    double av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { av[q] = colp[jrow[q]]; bv[q] = colp[irow[q]]; }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
        acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tj], bv[ti], acc[tj][ti], 0, 0, 0);
  }
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const long row = i0 + 16 * ti + l15;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const long col = j0 + 16 * tj + l4 + 4 * v;
        if (row < n && col < c1) {
          double *p = A + (size_t)col * (size_t)n + row;
          *p = *p - acc[tj][ti][v];
        }
      }
    }
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

size_t rbl_cholesky_work_bytes(int64_t) { return 0; }

int rbl_launch_cholesky(hipStream_t st, double *d_M, int64_t n, bool zero_upper, unsigned *d_err,
                        double *, size_t)
{
  for (int64_t k = 0; k < n; k += NB) {
    const int64_t pw = (n - k < NB) ? (n - k) : NB;  // panel width
    const int64_t pend = k + pw;
    for (int64_t kk = k; kk < pend; kk += IB) {
      const int nb = (int)((pend - kk < IB) ? (pend - kk) : IB);
      hipLaunchKernelGGL(k_potf2, dim3(1), dim3(256), 0, st, d_M, (long)n, (long)kk, nb, d_err);
      const int64_t rows = n - (kk + nb);
      if (rows > 0) {
        hipLaunchKernelGGL(k_trsm, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, d_M,
                           (long)n, (long)kk, nb);
        // rank-nb update of the rest of THIS panel: rows >= kk+nb, cols [kk+nb, pend)
        const int64_t r0 = kk + nb;
        if (r0 < pend) {
          if (nb % 4 != 0) return RBL_ERR_SIZE;  // cannot happen: n3 multiple of 3, handled below
          dim3 grid((unsigned)((n - r0 + 127) / 128), (unsigned)((pend - r0 + 127) / 128));
          hipLaunchKernelGGL(k_syrk_mfma, grid, dim3(256), 0, st, d_M, (long)n, (long)r0,
                             (long)pend, (long)kk, nb);
        }
      }
    }
    if (pend < n) {  // trailing update with the whole panel, K = pw
      dim3 grid((unsigned)((n - pend + 127) / 128), (unsigned)((n - pend + 127) / 128));
      hipLaunchKernelGGL(k_syrk_mfma, grid, dim3(256), 0, st, d_M, (long)n, (long)pend, (long)n,
                         (long)k, (int)pw);
    }
  }
  if (zero_upper)
    hipLaunchKernelGGL(k_zero_upper, dim3((unsigned)n, (unsigned)((n + 255) / 256)), dim3(256), 0,
                       st, d_M, (long)n);
  return RBL_OK;
}

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
