// Epilogue shared by the training-step GEMM kernels (nsr_gemm.hip: fp32 MFMA; nsr_gemm_f16.hip: split-fp16 MFMA).
// A wave holds a 64 x 64 quadrant as 2 x 2 accumulator blocks of v_mfma_f32_32x32x*: lane (column li, half h)
// of block (bi, bj) holds rows 8 (r >> 2) + 4 h + (r & 3), r = 0..15.
#pragma once
#include "nsr_gemm.h"

namespace nsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// bias -> activation -> ReLU mask of the consumer -> store (C row-major and / or C^T) -> per-tile column sums.
// `lds`: >= 2 * TN floats, free to use (the caller's K loop ended with a barrier); z = split-K slice.
template <int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], const bool (&col_on)[2], int64_t m0,
                                              int n0, int wm, int wn, int li, int h, int tid, int z, int64_t row_tile,
                                              float* lds) {
  float* C = g.C ? g.C + (g.splits > 1 ? (int64_t)z * g.split_stride : 0) : nullptr;
  float csum[2] = {0.0f, 0.0f};
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) {
    const int n = n0 + 64 * wn + 32 * bj + li;
    if (!col_on[bj] || n >= g.n_valid) continue;
    const float bias = g.bias ? g.bias[n] : 0.0f;
    const bool scaled = g.acc_scale != 0.0f && g.acc_scale != 1.0f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int64_t mb = m0 + 64 * wm + 32 * bi + 4 * h;
      float mk[16];
      if (g.mask) {   // all 16 loads in flight before the first use
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int64_t m = mb + 8 * (r >> 2) + (r & 3);
          m = m < g.M ? m : g.M - 1;
          mk[r] = g.mask[m * g.ldm + n];
        }
      }
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t m = mb + 8 * rq + e;
          float x = scaled ? fmaf(acc[bi][bj][4 * rq + e], g.acc_scale, bias) : acc[bi][bj][4 * rq + e] + bias;
          if (g.act == kActRelu) x = nsr_relu_nan(x);
          else if (g.act == kActSigmoid) x = 1.0f / (1.0f + expf(-x));
          else if (g.act == kActTanh) x = tanhf(x);
          if (g.mask) x = mk[4 * rq + e] > 0.0f ? x : 0.0f;
          v[e] = x;
          if (m < g.M) {
            if (C) C[m * g.ldc + n] = x;
            csum[bj] += x;
          }
        }
        if (g.Ct) {
          const int64_t m = mb + 8 * rq;
          float* dst = g.Ct + (int64_t)n * g.ldct + m;
          if (m + 3 < g.M) *reinterpret_cast<f32x4*>(dst) = v;
          else
            for (int e = 0; e < 4; ++e)
              if (m + e < g.M) dst[e] = v[e];
        }
      }
    }
  }
  if (g.col_sums) {   // column sums of this 128-row tile: halves by shuffle, the two row-waves through LDS
    float* red = lds;   // the K loop is over (its last statement is a barrier)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const float s = csum[bj] + __shfl_xor(csum[bj], 32, 64);
      if (h == 0) red[wm * TN + 64 * wn + 32 * bj + li] = s;
    }
    __syncthreads();
    if (tid < TN && n0 + tid < g.n_valid) g.col_sums[row_tile * g.N + n0 + tid] = red[tid] + red[TN + tid];
  }
}

}  // namespace nsr
