// Shared host/device helpers of libnsr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nsr.h"

#define NSR_WAVE 64

#define NSR_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t nsr_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- torch.linspace(0, 1, n) element i, fp32 (ATen CPU kernel: symmetric form,
// start + step*i below the midpoint, end - step*(n-1-i) above it).
__device__ __forceinline__ float nsr_linspace01(int i, int n) {
  if (n == 1) return 0.0f;
  const float step = 1.0f / (float)(n - 1);
  return (i < n / 2) ? __fmul_rn(step, (float)i) : __fsub_rn(1.0f, __fmul_rn(step, (float)(n - 1 - i)));
}

// z of the deterministic stratified sampler (models/utils.py:31-35), fp32, no contraction.
__device__ __forceinline__ float nsr_coarse_z(float near_, float far_, float t, int lindisp) {
  if (lindisp) {
    const float a = __fmul_rn(__fdiv_rn(1.0f, near_), __fsub_rn(1.0f, t));
    const float b = __fmul_rn(__fdiv_rn(1.0f, far_), t);
    return __fdiv_rn(1.0f, __fadd_rn(a, b));
  }
  return __fadd_rn(__fmul_rn(near_, __fsub_rn(1.0f, t)), __fmul_rn(far_, t));
}

// ---- wave64 reductions / scans (DPP-free, shuffle based) -------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// inclusive prefix product over the 64 lanes
__device__ __forceinline__ double wave_scan_mul_d(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double t = __shfl_up(v, o, 64);
    if (lane >= o) v *= t;
  }
  return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wave_scan_add_d(double v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
