// N2: depth warp into the reference view (include/nsr_warp.h; reference: warp.py:100-176).  One thread per pixel,
// HBM-bound (4 B read, 24 + 12 B written per pixel); float64 arithmetic in the reference's operation order
// (this TU is built with -ffp-contract=off; the products below must not fuse).
#include "nsr_common.h"
#include "../../include/nsr_warp.h"

namespace {

struct WarpArgs {
  float c2w[12];
  double ref[12];
  double focal, half_w, half_h;
  float half_w32, half_h32, focal32;
  int H, W, ndc;   // ndc: nsr_depth_kind
};

__device__ __forceinline__ double affine_row(const double* m, double x0, double x1, double x2) {
  return __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m[0], x0), __dmul_rn(m[1], x1)), __dmul_rn(m[2], x2)), m[3]);
}

__global__ void __launch_bounds__(256) depth_warp_kernel(WarpArgs a, const float* __restrict__ depth,
                                                         const float* __restrict__ ref_rgb, double* __restrict__ locs,
                                                         float* __restrict__ warped) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)a.H * a.W;
  if (p >= n) return;
  const int y = (int)(p / a.W), x = (int)(p % a.W);
  float D = depth[p];
  const float gx = __fsub_rn((float)x + 0.5f, a.half_w32), gy = -__fsub_rn((float)y + 0.5f, a.half_h32);
  if (a.ndc == NSR_DEPTH_NDC) {
    D = __fdiv_rn(1.0f, __fadd_rn(__fsub_rn(1.0f, D), 1e-6f));
  } else if (a.ndc == NSR_DEPTH_RAY) {
    // distance along the UNIT-norm ray through the pixel centre (get_rays normalises, models/utils.py:150) -> depth
    // along the camera axis: divide by |(gx/f, gy/f, -1)|, float32
    const float cx = __fdiv_rn(gx, a.focal32), cy = __fdiv_rn(gy, a.focal32);
    D = __fdiv_rn(D, sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(cx, cx), __fmul_rn(cy, cy)), 1.0f)));
  }
  const double Dd = (double)D;
  const double x0 = __dmul_rn(__ddiv_rn((double)gx, a.focal), Dd);
  const double x1 = __dmul_rn(__ddiv_rn((double)gy, a.focal), Dd);
  const double x2 = (double)(-D);
  double c[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) c[i] = (double)a.c2w[i];
  const double w0 = affine_row(c, x0, x1, x2), w1 = affine_row(c + 4, x0, x1, x2), w2 = affine_row(c + 8, x0, x1, x2);
  const double q0 = affine_row(a.ref, w0, w1, w2), q1 = affine_row(a.ref + 4, w0, w1, w2), q2 = affine_row(a.ref + 8, w0, w1, w2);
  const double den = -q2;
  const double u = trunc(__dadd_rn(__dmul_rn(__ddiv_rn(q0, den), a.focal), a.half_w));
  const double v = trunc(__dadd_rn(__dmul_rn(__ddiv_rn(q1, den), -a.focal), a.half_h));
  locs[p * 3 + 0] = u;
  locs[p * 3 + 1] = v;
  locs[p * 3 + 2] = __ddiv_rn(q2, den);
  if (warped) {
    const bool inside = u >= 0.0 && u < (double)a.W && v >= 0.0 && v < (double)a.H;
    const int64_t src = inside ? (int64_t)v * a.W + (int64_t)u : 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) warped[ch * n + p] = inside ? ref_rgb[ch * n + src] : 0.0f;
  }
}

}  // namespace

extern "C" int nsr_depth_warp(const float* depth, int H, int W, double focal, const float* c2w, const double* ref_w2c,
                              int ndc, const float* ref_rgb, double* locs, float* warped, void* stream) {
  if (H < 0 || W < 0 || !c2w || !ref_w2c || !(focal > 0.0)) return NSR_ERR_INVALID_ARG;
  if (ndc != NSR_DEPTH_METRIC && ndc != NSR_DEPTH_NDC && ndc != NSR_DEPTH_RAY) return NSR_ERR_UNSUPPORTED;
  if ((int64_t)H * W == 0) return NSR_OK;
  if (!depth || !locs || (warped && !ref_rgb)) return NSR_ERR_INVALID_ARG;
  WarpArgs a;
  for (int i = 0; i < 12; ++i) {
    a.c2w[i] = c2w[i];
    a.ref[i] = ref_w2c[i];
  }
  a.focal = focal;
  a.half_w = W / 2.0;
  a.half_h = H / 2.0;
  a.half_w32 = (float)(W / 2.0);
  a.half_h32 = (float)(H / 2.0);
  a.focal32 = (float)focal;
  a.H = H; a.W = W; a.ndc = ndc;
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(depth_warp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nsr_stream(stream), a, depth,
                     ref_rgb, locs, warped);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
