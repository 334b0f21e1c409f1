// Wave timeline of the symmetric product kernel at a mid-size system (cfg 2: 8 100 blobs, free space): when every
// workgroup's wave started and ended (100 MHz constant clock) and on which XCD / SE / CU / SIMD it ran.  Compiles the
// product's own kernel source with -DRBL_WAVE_TRACE (the hooks are empty in the library build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DRBL_WAVE_TRACE -Irigid_body_light_amd/csrc tools/wave_trace.hip -o tools/wave_trace
//   tools/wave_trace [n_blobs [wall [rows_per_lane [chunk [waves_per_workgroup [queue: -1 off]]]]]] > gpurun_out/wave_trace.csv       (summary on stderr)
#include <hip/hip_runtime.h>
__device__ unsigned long long *g_wave_trace = nullptr;
#include "rbl_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

int main(int argc, char **argv)
{
  const long N = argc > 1 ? atol(argv[1]) : 8100;
  const bool wall = argc > 2 && atoi(argv[2]) != 0;
  std::mt19937_64 gen(1);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const double a = 0.12, box = 14.0;                       // cfg 2's blob radius, ~its extent
  std::vector<double> r(3 * N), F(3 * N);
  for (long i = 0; i < N; ++i) { r[3 * i] = box * U(gen); r[3 * i + 1] = box * U(gen); r[3 * i + 2] = 1.0 + box * U(gen); }
  for (auto &f : F) f = U(gen) - 0.5;
  hipStream_t st; hipStreamCreate(&st);
  int ncu = 256;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); ncu = prop.multiProcessorCount;
  RblSymTune tune{};
  if (argc > 3) tune.ni1 = atoi(argv[3]);
  if (argc > 4) tune.chunk = atoi(argv[4]);
  if (argc > 5) tune.sw = atoi(argv[5]);
  if (argc > 6) tune.queue = atoi(argv[6]);
  const size_t wb = rbl_apply_M_sym_bytes(N, ncu, 1, 1, tune);
  double *dr, *dF, *dU, *dW; unsigned *derr;
  hipMalloc((void **)&dr, 24 * N); hipMalloc((void **)&dF, 24 * N); hipMalloc((void **)&dU, 24 * N); hipMalloc((void **)&dW, wb);
  hipMalloc((void **)&derr, 4); hipMemset(derr, 0, 4);
  hipMemcpy(dr, r.data(), 24 * N, hipMemcpyHostToDevice); hipMemcpy(dF, F.data(), 24 * N, hipMemcpyHostToDevice);
  const RblParams P = rbl_make_params(a, 1.0);
  const long T = (N + 63) / 64;
  const size_t nwg = (size_t)T * T;                          // upper bound of the grid (row groups x chunks)
  unsigned long long *dT; hipMalloc((void **)&dT, 32 * nwg); hipMemset(dT, 0, 32 * nwg);
  for (int rep = 0; rep < 5; ++rep) rbl_launch_apply_M_sym(st, P, wall, dF, dr, N, 0, 1, dU, dW, ncu, derr, 1, tune);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int rep = 0; rep < 20; ++rep) rbl_launch_apply_M_sym(st, P, wall, dF, dr, N, 0, 1, dU, dW, ncu, derr, 1, tune);
  hipEventRecord(e1, st); hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  fprintf(stderr, "N = %ld: %.1f us per product (pair kernel + slab sums), untraced\n", N, ms / 20 * 1e3);
  hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace), &dT, sizeof(dT));
  rbl_launch_apply_M_sym(st, P, wall, dF, dr, N, 0, 1, dU, dW, ncu, derr, 1, tune);
  hipStreamSynchronize(st);
  std::vector<unsigned long long> h(4 * nwg);
  hipMemcpy(h.data(), dT, 32 * nwg, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  size_t live = 0;
  for (size_t w = 0; w < nwg; ++w) if (h[4 * w]) { t0 = std::min(t0, h[4 * w]); t1 = std::max(t1, h[4 * w + 1]); ++live; }
  printf("workgroup,start_ns,end_ns,xcc,se,cu,simd,wave_slot\n");
  std::map<unsigned, std::vector<std::pair<double, double>>> per_simd;
  double life = 0.0;
  for (size_t w = 0; w < nwg; ++w) {
    if (!h[4 * w]) continue;
    const unsigned hw = (unsigned)h[4 * w + 2], xcc = (unsigned)h[4 * w + 3] & 15u;
    const unsigned wave = hw & 15u, simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, se = (hw >> 13) & 7u;
    const double s = (h[4 * w] - t0) * 10.0, e = (h[4 * w + 1] - t0) * 10.0;
    printf("%zu,%.0f,%.0f,%u,%u,%u,%u,%u\n", w, s, e, xcc, se, cu, simd, wave);
    per_simd[(xcc << 16) | (se << 8) | (cu << 2) | simd].push_back({s, e});
    life += e - s;
  }
  const double span = (t1 - t0) * 10.0;
  fprintf(stderr, "traced launch: %zu workgroups with rows, first start -> last end %.1f us, mean wave life %.1f us, SIMDs used %zu\n", live,
          span / 1e3, life / live / 1e3, per_simd.size());
  // start-time histogram (5 us bins), units per SIMD, mean concurrency
  std::vector<int> hist((size_t)(span / 5000.0) + 1, 0), endh(hist.size(), 0);
  for (size_t w = 0; w < nwg; ++w) if (h[4 * w]) { ++hist[(size_t)((h[4 * w] - t0) * 10.0 / 5000.0)]; ++endh[(size_t)((h[4 * w + 1] - t0) * 10.0 / 5000.0)]; }
  fprintf(stderr, "starts per 5 us:");
  for (int v : hist) fprintf(stderr, " %d", v);
  fprintf(stderr, "\nends   per 5 us:");
  for (int v : endh) fprintf(stderr, " %d", v);
  std::map<size_t, int> cnt;
  double last_min = 1e30, last_max = 0;
  for (auto &kv : per_simd) {
    ++cnt[kv.second.size()];
    double le = 0; for (auto &p : kv.second) le = std::max(le, p.second);
    last_min = std::min(last_min, le); last_max = std::max(last_max, le);
  }
  fprintf(stderr, "\nunits per SIMD (count of SIMDs):");
  for (auto &kv : cnt) fprintf(stderr, " %zu:%d", kv.first, kv.second);
  fprintf(stderr, "\nlast end per SIMD: earliest %.1f us, latest %.1f us\n", last_min / 1e3, last_max / 1e3);
  return 0;
}
