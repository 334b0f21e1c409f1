// Host-side cost of the runtime calls a launch-bound solver loop is made of (gfx950, ROCm 7.2): what one
// hipLaunchKernelGGL / hipMemcpyAsync / hipMemsetAsync costs the CALLING thread, and how long a small device-to-host
// round trip takes by a copy + stream drain versus a kernel writing into pinned host memory + stream drain.
//   hipcc --offload-arch=gfx950 -O2 -w tools/launch_costs.hip -o tools/launch_costs && tools/launch_costs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_copy(double *dst, const double *src, long n)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ void k_fetch_flag(unsigned *h, unsigned *d) { *h = *d; *d = 0; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  hipStream_t st; hipStreamCreate(&st);
  const long n = 24300;
  double *a, *b, *hp; unsigned *dflag, *hflag;
  hipMalloc((void **)&a, 8 * n); hipMalloc((void **)&b, 8 * n); hipMalloc((void **)&dflag, 4);
  hipHostMalloc((void **)&hp, 8 * 4096, hipHostMallocDefault); hipHostMalloc((void **)&hflag, 4, hipHostMallocDefault);
  std::vector<double> pageable(4096);
  hipMemset(a, 0, 8 * n); hipMemset(dflag, 0, 4);
  const int R = 2000;
  auto timeit = [&](const char *name, auto fn) {
    for (int i = 0; i < 50; ++i) fn();
    hipStreamSynchronize(st);
    const double t0 = now();
    for (int i = 0; i < R; ++i) fn();
    const double t1 = now();
    hipStreamSynchronize(st);
    const double t2 = now();
    printf("%-62s host %.2f us per call, drained after %.2f us per call\n", name, (t1 - t0) / R * 1e6, (t2 - t0) / R * 1e6);
  };
  timeit("kernel launch (copy kernel, 24 300 doubles)", [&] { hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, st, b, a, n); });
  timeit("hipMemcpyAsync device -> device, 24 300 doubles", [&] { hipMemcpyAsync(b, a, 8 * n, hipMemcpyDeviceToDevice, st); });
  timeit("hipMemsetAsync, 24 300 doubles", [&] { hipMemsetAsync(b, 0, 8 * n, st); });
  timeit("hipMemsetAsync, 4 bytes", [&] { hipMemsetAsync(dflag, 0, 4, st); });
  timeit("hipMemcpyAsync pinned host -> device, 20 doubles", [&] { hipMemcpyAsync(b, hp, 160, hipMemcpyHostToDevice, st); });
  timeit("kernel reading pinned host memory, 20 doubles", [&] { hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, st, b, (const double *)hp, 20L); });
  auto rt = [&](const char *name, auto fn) {
    for (int i = 0; i < 20; ++i) { fn(); hipStreamSynchronize(st); }
    const double t0 = now();
    for (int i = 0; i < R; ++i) { fn(); hipStreamSynchronize(st); }
    printf("%-62s round trip %.2f us\n", name, (now() - t0) / R * 1e6);
  };
  rt("D2H 4 bytes: hipMemcpyAsync to pinned + hipMemsetAsync + drain", [&] { hipMemcpyAsync(hflag, dflag, 4, hipMemcpyDeviceToHost, st); hipMemsetAsync(dflag, 0, 4, st); });
  rt("D2H 4 bytes: one kernel writing pinned memory + drain", [&] { hipLaunchKernelGGL(k_fetch_flag, dim3(1), dim3(1), 0, st, hflag, dflag); });
  rt("D2H 340 doubles: hipMemcpyAsync to PAGEABLE + drain", [&] { hipMemcpyAsync(pageable.data(), a, 8 * 340, hipMemcpyDeviceToHost, st); });
  rt("D2H 340 doubles: hipMemcpyAsync to pinned + drain", [&] { hipMemcpyAsync(hp, a, 8 * 340, hipMemcpyDeviceToHost, st); });
  rt("D2H 340 doubles: kernel writing pinned memory + drain", [&] { hipLaunchKernelGGL(k_copy, dim3(2), dim3(256), 0, st, hp, (const double *)a, 340L); });
  rt("empty: drain of an idle stream", [&] {});
  return 0;
}
