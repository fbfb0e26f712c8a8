// Input side of the fused MLP kernels: the 32 + 16 encoded values lane (m, h) contributes for its sample point,
// in the register order of nsr_mlp_layout.h (pecol / dircol).
#pragma once
#include "nsr_common.h"
#include "nsr_mlp_layout.h"

namespace nsr {

// MODE 0: x is (P, 90) embedded rows (VanillaMLP.forward).  MODE 1: x is rays (R, stride), zv (R, N): cast_rays
// (models/utils.py:14, separate multiply and add as ATen does) + both positional encodings
// (models/embedding.py:44-62) computed here.
template <int MODE>
__device__ __forceinline__ void encode_point(const float* __restrict__ x, const float* __restrict__ zv, int64_t pc,
                                             int samples_per_ray, int stride, int h, float (&pe)[32], float (&de)[16]) {
  if (MODE == 0) {
    const float* row = x + pc * kInCh;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const int col = pecol(t, h);
      pe[t] = (col == kPad) ? 0.0f : row[col];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int col = dircol(t, h);
      de[t] = (col == kPad) ? 0.0f : row[kPosCh + col];
    }
  } else {
    const int64_t ray = pc / samples_per_ray;
    const NsrRay rq = nsr_load_ray(x, ray, stride);
    const float zk = zv[pc];
    const float d[3] = {rq.v[0], rq.v[1], rq.v[2]};           // the direction that is ENCODED
    const float v[3] = {__fadd_rn(rq.o[0], __fmul_rn(zk, rq.d[0])), __fadd_rn(rq.o[1], __fmul_rn(zk, rq.d[1])),
                        __fadd_rn(rq.o[2], __fmul_rn(zk, rq.d[2]))};
    pe[0] = h ? v[2] : v[0];
    pe[1] = h ? 0.0f : v[1];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float sn, cs;
        nsr_sincos(ldexpf(v[c], 5 * h + f), sn, cs);
        pe[2 + 6 * f + c] = sn;
        pe[2 + 6 * f + 3 + c] = cs;
      }
    de[0] = h ? d[2] : d[0];
    de[1] = h ? 0.0f : d[1];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float sn, cs;
        nsr_sincos(ldexpf(d[c], 2 * h + f), sn, cs);
        de[2 + 6 * f + c] = sn;
        de[2 + 6 * f + 3 + c] = cs;
      }
    de[14] = 0.0f;
    de[15] = 0.0f;
  }
}

}  // namespace nsr
