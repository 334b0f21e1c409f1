// peak_fp32pk.hip -- issue rates of the instructions the RELAXED (packed single precision) pair sweep is made of,
// measured on the box: v_pk_fma_f32, v_pk_mul_f32 with a splat (op_sel) operand, v_rsq_f32, ds_add_f32 / ds_add_f64.
//   hipcc -O3 --offload-arch=gfx950 tools/peak_fp32pk.hip -o tools/peak_fp32pk && tools/peak_fp32pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_pkfma(float *out, int iters)
{
  f2 a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const f2 b = {0.999999f, 0.99999f}, c = {1e-9f, 2e-9f};
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_elementwise_fma(a0, b, c); a1 = __builtin_elementwise_fma(a1, b, c); a2 = __builtin_elementwise_fma(a2, b, c);
    a3 = __builtin_elementwise_fma(a3, b, c); a4 = __builtin_elementwise_fma(a4, b, c); a5 = __builtin_elementwise_fma(a5, b, c);
    a6 = __builtin_elementwise_fma(a6, b, c); a7 = __builtin_elementwise_fma(a7, b, c);
  }
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void k_pkfma_splat(float *out, const float *in, int iters)
{
  f2 a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float bs = in[threadIdx.x & 7];
  const f2 c = {1e-9f, 2e-9f};
  for (int i = 0; i < iters; ++i) {
    const f2 b = {bs, bs};
    a0 = __builtin_elementwise_fma(a0, b, c); a1 = __builtin_elementwise_fma(a1, b, c); a2 = __builtin_elementwise_fma(a2, b, c);
    a3 = __builtin_elementwise_fma(a3, b, c); a4 = __builtin_elementwise_fma(a4, b, c); a5 = __builtin_elementwise_fma(a5, b, c);
    a6 = __builtin_elementwise_fma(a6, b, c); a7 = __builtin_elementwise_fma(a7, b, c);
  }
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void k_fma32(float *out, int iters)
{
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 0.999999f, c = 1e-9f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
    a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(256) void k_rsq32(float *out, int iters)
{
  float a0 = 1.0f + threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_rsqf(a0) + 1.5f; a1 = __builtin_amdgcn_rsqf(a1) + 1.5f;
    a2 = __builtin_amdgcn_rsqf(a2) + 1.5f; a3 = __builtin_amdgcn_rsqf(a3) + 1.5f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <class T>
__global__ __launch_bounds__(64) void k_dsadd(T *out, int iters)
{
  __shared__ T s[3][64];
  const int lane = threadIdx.x;
  s[0][lane] = 0; s[1][lane] = 0; s[2][lane] = 0;
  __syncthreads();
  T v = (T)(lane * 1e-3);
  for (int i = 0; i < iters; ++i) {
    const int jj = (lane + i) & 63;
    __hip_atomic_fetch_add(&s[0][jj], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&s[1][jj], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&s[2][jj], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  out[blockIdx.x * 64 + lane] = s[0][lane] + s[1][lane] + s[2][lane];
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
  printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
  const int blocks = p.multiProcessorCount * 8, iters = 20000;
  float *out, *in; hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&in, 64); hipMemset(in, 0, 64);
  const double thr = (double)blocks * 256;
  double ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("v_pk_fma_f32        : %8.2f TFLOP/s  (%.2f G wave-instr/s)\n", 4.0 * 8 * iters * thr / (ms * 1e-3) / 1e12, 8.0 * iters * thr / 64 / (ms * 1e-3) / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma_splat, dim3(blocks), dim3(256), 0, 0, out, (const float *)in, iters); });
  printf("v_pk_fma_f32 (splat): %8.2f TFLOP/s\n", 4.0 * 8 * iters * thr / (ms * 1e-3) / 1e12);
  ms = time_ms([&] { hipLaunchKernelGGL(k_fma32, dim3(blocks), dim3(256), 0, 0, out, iters); });
  printf("v_fma_f32           : %8.2f TFLOP/s  (%.2f G wave-instr/s)\n", 2.0 * 8 * iters * thr / (ms * 1e-3) / 1e12, 8.0 * iters * thr / 64 / (ms * 1e-3) / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(k_rsq32, dim3(blocks), dim3(256), 0, 0, out, iters / 4); });
  printf("v_rsq_f32 (+add)    : %8.2f G wave-instr pairs/s\n", 4.0 * (iters / 4) * thr / 64 / (ms * 1e-3) / 1e9);
  const int b64 = p.multiProcessorCount * 12;
  ms = time_ms([&] { hipLaunchKernelGGL(k_dsadd<float>, dim3(b64), dim3(64), 0, 0, out, iters); });
  printf("ds_add_f32          : %8.2f G wave-instr/s\n", 3.0 * iters * (double)b64 / (ms * 1e-3) / 1e9);
  ms = time_ms([&] { hipLaunchKernelGGL(k_dsadd<double>, dim3(b64), dim3(64), 0, 0, (double *)out, iters); });
  printf("ds_add_f64          : %8.2f G wave-instr/s\n", 3.0 * iters * (double)b64 / (ms * 1e-3) / 1e9);
  return 0;
}
