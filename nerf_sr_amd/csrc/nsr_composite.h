// V1 for ONE ray by ONE 64-lane wavefront (reference: models/rendering.py:75-111) -- shared by the stand-alone compositor
// (nsr_render.hip) and by the MLP kernels that composite the rays they have just evaluated (the (R, N, 4) network output
// then never goes to HBM).  Numerics follow the reference's CPU path: torch.cumprod on CPU accumulates fp32 inputs in
// double and rounds every output element to fp32 (ATen cpu_cum_base_kernel, acc_type<float, false> = double), so the
// transmittance scan runs in double and rounds per element as well.  Both users compile with -ffp-contract=off.
#pragma once
#include "nsr_common.h"

// `white`: the entry points' `white_bkgd` word (include/nsr.h): bit 0 = white background, bit 1 = sigma_activation 'softplus'.
// K samples per lane, lane l owns samples l*K .. l*K+K-1 of the ray whose sample k lives at rgb[k * rgb_stride + c],
// sigma[k * sigma_stride], z[k] (pointers already offset to the ray); r = the ray's index in the output arrays.
template <int K>
__device__ __forceinline__ void composite_ray(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride,
                                              const float* z, int N, int white, int lane, int64_t r,
                                              float* __restrict__ comp_rgb, float* __restrict__ depth,
                                              float* __restrict__ opacity, float* __restrict__ weights) {
  float zk[K], sg[K], cr[K], cg[K], cb[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = lane * K + i;
    const bool ok = k < N;
    const int64_t p = ok ? k : N - 1;
    zk[i] = z[p];
    sg[i] = sigma[p * sigma_stride];
    cr[i] = rgb[p * rgb_stride + 0];
    cg[i] = rgb[p * rgb_stride + 1];
    cb[i] = rgb[p * rgb_stride + 2];
  }
  const float z_next_lane = __shfl_down(zk[0], 1, 64);
  float alpha[K];
  double pl[K];   // lane-local inclusive products of (1 - alpha + 1e-10)
  double run = 1.0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = lane * K + i;
    const float zn = (i + 1 < K) ? zk[(i + 1 < K) ? i + 1 : i] : z_next_lane;
    const float delta = (k >= N - 1) ? 1e10f : __fsub_rn(zn, zk[i]);
    const float s = (white & NSR_SIGMA_SOFTPLUS) ? nsr_softplus_density(sg[i]) : fmaxf(sg[i], 0.0f);
    float a = __fsub_rn(1.0f, expf(__fmul_rn(-delta, s)));
    if (k >= N) a = 0.0f;
    alpha[i] = a;
    const float f = (k < N) ? __fadd_rn(__fsub_rn(1.0f, a), 1e-10f) : 1.0f;
    run *= (double)f;
    pl[i] = run;
  }
  // exclusive prefix over lanes of the lane totals
  const double incl = wave_scan_mul_d(run, lane);
  double excl = __shfl_up(incl, 1, 64);
  if (lane == 0) excl = 1.0;
  float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_o = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = lane * K + i;
    // T_k = fp32( prod_{j<k} f_j ), T_0 = 1 exactly
    const double t_d = (i == 0) ? excl : excl * pl[(i > 0) ? i - 1 : 0];
    const float T = (k == 0) ? 1.0f : (float)t_d;
    const float w = __fmul_rn(alpha[i], T);
    if (k < N) {
      if (weights) weights[r * N + k] = w;
      acc_r = __fadd_rn(acc_r, __fmul_rn(w, cr[i]));
      acc_g = __fadd_rn(acc_g, __fmul_rn(w, cg[i]));
      acc_b = __fadd_rn(acc_b, __fmul_rn(w, cb[i]));
      acc_d = __fadd_rn(acc_d, __fmul_rn(w, zk[i]));
      acc_o = __fadd_rn(acc_o, w);
    }
  }
  acc_r = wave_sum(acc_r); acc_g = wave_sum(acc_g); acc_b = wave_sum(acc_b);
  acc_d = wave_sum(acc_d); acc_o = wave_sum(acc_o);
  if (lane == 0) {
    if (white & NSR_WHITE_BKGD) {
      const float bg = __fsub_rn(1.0f, acc_o);
      acc_r = __fadd_rn(acc_r, bg); acc_g = __fadd_rn(acc_g, bg); acc_b = __fadd_rn(acc_b, bg);
    }
    if (comp_rgb) { comp_rgb[r * 3 + 0] = acc_r; comp_rgb[r * 3 + 1] = acc_g; comp_rgb[r * 3 + 2] = acc_b; }
    if (depth) depth[r] = acc_d;
    if (opacity) opacity[r] = acc_o;
  }
}

// Where a fused MLP launch puts the composited rays (any pointer may be null); used when COMP is set
struct NsrCompOut {
  float* comp_rgb;
  float* depth;
  float* opacity;
  float* weights;
  int white;
};

// Epilogue of a 128-point MLP tile whose points are whole rays (NS = 64: two rays, NS = 128: one ray): stage the tile's
// (r, g, b, sigma) and z in `lds` (>= 640 floats, free for the workgroup to use; the caller has drained its DMAs), then
// one wavefront per ray composites.  `mine`: this lane holds the result of point wave * 32 + m.  `tile`: index of the
// 128-point tile in the launch (blockIdx.x for one-tile workgroups).  DEDICATED: `lds` is used for nothing else and at
// least one workgroup barrier lies between two calls (persistent kernels), so the barrier that protects the previous
// content is not needed.
template <int NS, bool DEDICATED = false>
__device__ __forceinline__ void composite_tile(float* lds, bool mine, int wave, int m, int lane, float4 value, float zk,
                                               int64_t n_rays, const NsrCompOut& co, int64_t tile) {
  static_assert(NS == 64 || NS == 128, "tiles of 128 points must hold whole rays");
  if (!DEDICATED) __syncthreads();                 // every wave is done with whatever the LDS region held before
  float4* st = reinterpret_cast<float4*>(lds);
  float* zst = lds + 4 * 128;
  if (mine) {
    st[wave * 32 + m] = value;
    zst[wave * 32 + m] = zk;
  }
  __syncthreads();
  constexpr int kRays = 128 / NS;
  if (wave < kRays) {
    const int64_t ray = tile * kRays + wave;
    if (ray < n_rays)
      composite_ray<NS / 64>(lds + 4 * NS * wave, 4, lds + 4 * NS * wave + 3, 4, zst + NS * wave, NS, co.white, lane, ray,
                             co.comp_rgb, co.depth, co.opacity, co.weights);
  }
}
