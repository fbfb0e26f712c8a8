// Ray-side stages of the path: sub-pixel ray generation (R1-R4), positional
// encoding as a stand-alone op (E1), stratified sampling (S1), the s^2 mean (A1)
// and the HR un-flatten (A2).  All HBM-bound elementwise kernels: one thread per
// output element / ray, coalesced stores, fp32 arithmetic in the reference's
// operation order (compiled with -ffp-contract=off).
#include "nsr_common.h"

// ---------------------------------------------------------------------------
// R1-R4  (reference: models/utils.py:98-196, data/llff_downX_dataset.py:473-490)
// ---------------------------------------------------------------------------
struct GenRaysArgs {
  float c2w[12];
  int H, W, s, ndc;
  float focal, half_w, half_h;   // W/2, H/2 as fp32
  float ndc_ax, ndc_ay;          // -1/(W/(2f)), -1/(H/(2f)) evaluated in double on the host
  float near_, far_;
};

// rays of the LR pixels [lr0, lr0 + n_rays / s^2): output row r holds ray  lr0 * s^2 + r  of the frame
__global__ void __launch_bounds__(256) gen_rays_kernel(GenRaysArgs a, float* __restrict__ rays, int64_t lr0,
                                                       int64_t n_rays) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const int s = a.s, s2 = s * s, w_lr = a.W / s;
  const int64_t lr = lr0 + r / s2;
  const int sub = (int)(r % s2);
  const int py = (int)(lr / w_lr) * s + sub / s;   // HR row  (h s1)
  const int px = (int)(lr % w_lr) * s + sub % s;   // HR col  (w s2)
  // camera-space direction through the pixel centre
  const float cx = __fdiv_rn(__fsub_rn((float)px + 0.5f, a.half_w), a.focal);
  const float cy = -__fdiv_rn(__fsub_rn((float)py + 0.5f, a.half_h), a.focal);
  const float cz = -1.0f;
  // rotate into the world frame, normalise
  float d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    d[k] = fmaf(cz, a.c2w[4 * k + 2], fmaf(cy, a.c2w[4 * k + 1], __fmul_rn(cx, a.c2w[4 * k + 0])));
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
  d[0] = __fdiv_rn(d[0], nrm); d[1] = __fdiv_rn(d[1], nrm); d[2] = __fdiv_rn(d[2], nrm);
  float o[3] = {a.c2w[3], a.c2w[7], a.c2w[11]};
  float nr = a.near_, fr = a.far_;
  if (a.ndc) {
    // shift the origin to the near plane (near = 1.0), then project
    const float t = __fdiv_rn(-__fadd_rn(1.0f, o[2]), d[2]);
    o[0] = __fadd_rn(o[0], __fmul_rn(t, d[0]));
    o[1] = __fadd_rn(o[1], __fmul_rn(t, d[1]));
    o[2] = __fadd_rn(o[2], __fmul_rn(t, d[2]));
    const float ox_oz = __fdiv_rn(o[0], o[2]);
    const float oy_oz = __fdiv_rn(o[1], o[2]);
    const float o0 = __fmul_rn(a.ndc_ax, ox_oz);
    const float o1 = __fmul_rn(a.ndc_ay, oy_oz);
    const float o2 = __fadd_rn(1.0f, __fdiv_rn(2.0f, o[2]));
    const float d0 = __fmul_rn(a.ndc_ax, __fsub_rn(__fdiv_rn(d[0], d[2]), ox_oz));
    const float d1 = __fmul_rn(a.ndc_ay, __fsub_rn(__fdiv_rn(d[1], d[2]), oy_oz));
    const float d2 = __fsub_rn(1.0f, o2);
    o[0] = o0; o[1] = o1; o[2] = o2;
    d[0] = d0; d[1] = d1; d[2] = d2;
    nr = 0.0f; fr = 1.0f;
  }
  float4* out = reinterpret_cast<float4*>(rays + r * 8);
  out[0] = make_float4(o[0], o[1], o[2], d[0]);
  out[1] = make_float4(d[1], d[2], nr, fr);
}

extern "C" int nsr_gen_rays(const float* c2w, int H, int W, double focal, int s, int ndc, float near_, float far_,
                            float* rays_dev, void* stream) {
  if (H <= 0 || W <= 0 || s <= 0) return NSR_ERR_INVALID_ARG;
  return nsr_gen_rays_range(c2w, H, W, focal, s, ndc, near_, far_, 0, (int64_t)(H / s) * (W / s), rays_dev, stream);
}

extern "C" int nsr_gen_rays_range(const float* c2w, int H, int W, double focal, int s, int ndc, float near_, float far_,
                                  int64_t lr_lo, int64_t lr_hi, float* rays_dev, void* stream) {
  if (!c2w || H <= 0 || W <= 0 || s <= 0 || !(focal > 0.0)) return NSR_ERR_INVALID_ARG;
  if (H % s != 0 || W % s != 0) return NSR_ERR_INVALID_ARG;
  if (lr_lo < 0 || lr_hi < lr_lo || lr_hi > (int64_t)(H / s) * (W / s)) return NSR_ERR_INVALID_ARG;
  if (lr_hi == lr_lo) return NSR_OK;   // empty shard: nothing to write (the pointer may be null)
  if (!rays_dev || (reinterpret_cast<uintptr_t>(rays_dev) & 15) != 0) return NSR_ERR_INVALID_ARG;
  GenRaysArgs a;
  for (int i = 0; i < 12; ++i) a.c2w[i] = c2w[i];
  a.H = H; a.W = W; a.s = s; a.ndc = ndc;
  a.focal = (float)focal;
  a.half_w = (float)(W / 2.0);
  a.half_h = (float)(H / 2.0);
  a.ndc_ax = (float)(-1.0 / (W / (2.0 * focal)));
  a.ndc_ay = (float)(-1.0 / (H / (2.0 * focal)));
  a.near_ = near_; a.far_ = far_;
  const int64_t n = (lr_hi - lr_lo) * s * s;
  const int threads = 256;
  const int64_t blocks = (n + threads - 1) / threads;
  hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)blocks), dim3(threads), 0, nsr_stream(stream), a, rays_dev, lr_lo, n);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// ---------------------------------------------------------------------------
// E1  (reference: models/embedding.py:44-62)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) posenc_kernel(const float* __restrict__ x, int64_t n, int deg,
                                                     float* __restrict__ out) {
  const int C = 3 + 6 * deg;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * C) return;
  const int64_t row = idx / C;
  const int col = (int)(idx - row * C);
  float v;
  if (col < 3) {
    v = x[row * 3 + col];
  } else {
    const int f = (col - 3) / 6, c = (col - 3) % 6;
    const float arg = ldexpf(x[row * 3 + (c % 3)], f);   // 2^f * x, exact
    v = (c < 3) ? sinf(arg) : cosf(arg);
  }
  out[idx] = v;
}

extern "C" int nsr_posenc(const float* x, int64_t n, int deg, float* out, void* stream) {
  if (n < 0 || deg < 0 || deg > 16) return NSR_ERR_INVALID_ARG;
  if (n == 0) return NSR_OK;
  if (!x || !out) return NSR_ERR_INVALID_ARG;
  const int64_t total = n * (3 + 6 * deg);
  const int threads = 256;
  hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0,
                     nsr_stream(stream), x, n, deg, out);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// ---------------------------------------------------------------------------
// S1  (reference: models/utils.py:5-44)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sample_kernel(const float* __restrict__ rays, int stride, int64_t R, int N, int lindisp,
                                                     const float* __restrict__ u, float* __restrict__ z,
                                                     float* __restrict__ pts) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * N) return;
  const int64_t r = idx / N;
  const int k = (int)(idx - r * N);
  const NsrRay q = nsr_load_ray(rays, r, stride);
  const float near_ = q.near_, far_ = q.far_;
  float zk = nsr_coarse_z(near_, far_, nsr_linspace01(k, N), lindisp);
  if (u != nullptr) {
    // per-bin jitter: lower + u * (upper - lower)   (utils.py:37-41)
    const float zl = (k > 0) ? nsr_coarse_z(near_, far_, nsr_linspace01(k - 1, N), lindisp) : zk;
    const float zr = (k < N - 1) ? nsr_coarse_z(near_, far_, nsr_linspace01(k + 1, N), lindisp) : zk;
    const float lower = (k > 0) ? __fmul_rn(0.5f, __fadd_rn(zl, zk)) : zk;
    const float upper = (k < N - 1) ? __fmul_rn(0.5f, __fadd_rn(zk, zr)) : zk;
    zk = __fadd_rn(lower, __fmul_rn(u[idx], __fsub_rn(upper, lower)));
  }
  z[idx] = zk;
  if (pts != nullptr) {
    pts[idx * 3 + 0] = __fadd_rn(q.o[0], __fmul_rn(zk, q.d[0]));
    pts[idx * 3 + 1] = __fadd_rn(q.o[1], __fmul_rn(zk, q.d[1]));
    pts[idx * 3 + 2] = __fadd_rn(q.o[2], __fmul_rn(zk, q.d[2]));
  }
}

extern "C" int nsr_sample_along_rays(const float* rays, int ray_stride, int64_t R, int n_samples, int lindisp,
                                     const float* u, float* z, float* pts, void* stream) {
  if (R < 0 || n_samples <= 0 || !nsr_ray_stride_ok(ray_stride)) return NSR_ERR_INVALID_ARG;
  if (R == 0) return NSR_OK;
  if (!rays || !z || (ray_stride == 8 && (reinterpret_cast<uintptr_t>(rays) & 15) != 0)) return NSR_ERR_INVALID_ARG;
  const int64_t total = R * n_samples;
  const int threads = 256;
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0,
                     nsr_stream(stream), rays, ray_stride, R, n_samples, lindisp, u, z, pts);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// ---------------------------------------------------------------------------
// A1  (reference: models/nerf_downX_model.py:337-348)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sr_mean_kernel(const float* __restrict__ hr, int64_t n_lr, int s2, int c,
                                                      float* __restrict__ lr) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_lr * c) return;
  const int64_t p = idx / c;
  const int ch = (int)(idx - p * c);
  float acc = 0.0f;
  for (int k = 0; k < s2; ++k) acc = __fadd_rn(acc, hr[(p * s2 + k) * c + ch]);
  lr[idx] = __fdiv_rn(acc, (float)s2);
}

extern "C" int nsr_sr_mean(const float* hr, int64_t n_lr, int s2, int c, float* lr, void* stream) {
  if (n_lr < 0 || s2 <= 0 || c <= 0) return NSR_ERR_INVALID_ARG;
  if (n_lr == 0) return NSR_OK;
  if (!hr || !lr) return NSR_ERR_INVALID_ARG;
  const int64_t total = n_lr * c;
  const int threads = 256;
  hipLaunchKernelGGL(sr_mean_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0,
                     nsr_stream(stream), hr, n_lr, s2, c, lr);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// ---------------------------------------------------------------------------
// A2  (reference: models/nerf_downX_model.py:410-416)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) unflatten_kernel(const float* __restrict__ x, int H, int W, int s, int c,
                                                        float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over the OUTPUT (H, W, c)
  const int64_t total = (int64_t)H * W * c;
  if (idx >= total) return;
  const int ch = (int)(idx % c);
  const int64_t pix = idx / c;
  const int px = (int)(pix % W), py = (int)(pix / W);
  const int w_lr = W / s;
  const int64_t src = ((int64_t)(py / s) * w_lr + px / s) * (s * s) + (py % s) * s + (px % s);
  out[idx] = x[src * c + ch];
}

extern "C" int nsr_unflatten(const float* x, int H, int W, int s, int c, float* out, void* stream) {
  if (!x || !out || H <= 0 || W <= 0 || s <= 0 || c <= 0 || H % s != 0 || W % s != 0) return NSR_ERR_INVALID_ARG;
  const int64_t total = (int64_t)H * W * c;
  const int threads = 256;
  hipLaunchKernelGGL(unflatten_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0,
                     nsr_stream(stream), x, H, W, s, c, out);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
