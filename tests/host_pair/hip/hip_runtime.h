// Host stand-in for <hip/hip_runtime.h>, ONLY for tools/host_pair: compiles rigid_body_light_amd/csrc/rbl_pair.hpp
// (the device pair arithmetic) with g++ so that algebra changes can be checked against the reference fixtures on
// the CPU-only build container before a GPU run.  Not part of the product, never shipped in librbl.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __global__
static inline bool __any(bool x) { return x; }
// v_rsq_f64 stand-in: 1/sqrt(x) with a deterministic relative error of up to ~5e-8 (the measured seed accuracy on
// gfx950), so the Newton correction in rbl_rsqrt is exercised the way it is on the device
static inline double rbl_host_rsq_seed(double x)
{
  const double y = 1.0 / std::sqrt(x);
  uint64_t b; std::memcpy(&b, &x, 8);
  b = (b * 0x9E3779B97F4A7C15ull) >> 11;
  const double e = ((double)b / 9007199254740992.0 - 0.5) * 1.0e-7;
  return y * (1.0 + e);
}
#define __builtin_amdgcn_rsq(x) rbl_host_rsq_seed(x)
