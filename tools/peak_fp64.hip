// peak_fp64.hip -- measures, on the box, the fp64 ceilings the roofline fractions are
// priced against (the CDNA4 guide has no fp64 rows): v_fma_f64 rate, v_mfma_f64_16x16x4
// rate, v_rsq_f64 rate, and the accuracy of the v_rsq_f64 seed + our Newton step.
//   hipcc -O3 --offload-arch=gfx950 tools/peak_fp64.hip -o tools/peak_fp64 && tools/peak_fp64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../rigid_body_light_amd/csrc/rbl_pair.hpp"

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_dfma(double *out, int iters)
{
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 0.999999, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
    a4 = __builtin_fma(a4, b, c); a5 = __builtin_fma(a5, b, c); a6 = __builtin_fma(a6, b, c); a7 = __builtin_fma(a7, b, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(256) void k_mfma(double *out, int iters)
{
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = threadIdx.x * 1e-3, b = 1.0 - a;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ __launch_bounds__(256) void k_rsq(double *out, int iters)
{
  double a0 = 1.0 + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_rsq(a0) + 1.5; a1 = __builtin_amdgcn_rsq(a1) + 1.5;
    a2 = __builtin_amdgcn_rsq(a2) + 1.5; a3 = __builtin_amdgcn_rsq(a3) + 1.5;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

__global__ void k_rsq_acc(const double *x, double *seed, double *refined, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { seed[i] = __builtin_amdgcn_rsq(x[i]); refined[i] = rbl_rsqrt(x[i]); }
}

template <class F> double time_ms(F f)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  const int blocks = p.multiProcessorCount * 8, iters = 20000;
  double *out; hipMalloc(&out, sizeof(double) * blocks * 256);
  double ms = time_ms([&] { hipLaunchKernelGGL(k_dfma, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("v_fma_f64      : %8.2f TFLOP/s\n", 2.0 * 8 * iters * (double)blocks * 256 / (ms * 1e-3) / 1e12);
  ms = time_ms([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters / 4); });
  printf("mfma_f64_16x16x4: %7.2f TFLOP/s\n", 2.0 * 16 * 16 * 4 * 4 * (iters / 4) * (double)blocks * 4 / (ms * 1e-3) / 1e12);
  ms = time_ms([&] { hipLaunchKernelGGL(k_rsq, dim3(blocks), dim3(256), 0, 0, out, iters / 4); });
  printf("v_rsq_f64 (+add): %7.2f Gop/s  (fma-equivalent issue slots per rsq+add: %.2f)\n",
         4.0 * (iters / 4) * (double)blocks * 256 / (ms * 1e-3) / 1e9, 0.0);
  const int n = 1 << 20;
  std::vector<double> hx(n), hs(n), hr(n);
  for (int i = 0; i < n; ++i) hx[i] = std::exp(-30.0 + 60.0 * (i + 0.5) / n) * (1.0 + 0.37 * ((i * 2654435761u) % 1000) / 1000.0);
  double *dx, *ds, *dr; hipMalloc(&dx, n * 8); hipMalloc(&ds, n * 8); hipMalloc(&dr, n * 8);
  hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_rsq_acc, dim3(n / 256), dim3(256), 0, 0, dx, ds, dr, n);
  hipMemcpy(hs.data(), ds, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), dr, n * 8, hipMemcpyDeviceToHost);
  double es = 0, er = 0;
  for (int i = 0; i < n; ++i) {
    long double ref = 1.0L / sqrtl((long double)hx[i]);
    es = fmax(es, (double)fabsl((hs[i] - ref) / ref)); er = fmax(er, (double)fabsl((hr[i] - ref) / ref));
  }
  printf("v_rsq_f64 seed max rel err %.3e ; rbl_rsqrt max rel err %.3e (eps = 1.1e-16)\n", es, er);
  return 0;
}
