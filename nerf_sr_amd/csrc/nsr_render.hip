// Per-ray stages of the path: alpha compositing (V1) and hierarchical
// inverse-CDF resampling (S2).  One 64-lane wavefront owns one ray; the
// sigma->weight chain is a wavefront scan (shuffle based), the (rgb, depth,
// opacity) sums are wavefront reductions.  HBM-bound: each sample's 16-20 bytes
// are read once, the weights written once.
//
// Numerics follow the reference's CPU path: torch.cumprod / torch.cumsum on CPU
// accumulate fp32 inputs in double and round every output element to fp32
// (ATen cpu_cum_base_kernel, acc_type<float,false> = double), so the scans here
// run in double and round per element as well.
#include "nsr_common.h"
#include "nsr_composite.h"

#define NSR_MAX_SAMPLES 512   // composite: N <= 512 (8 samples per lane)
#define NSR_RS_MAX 256        // resample: Nc <= 256, Ni <= 256

// ---------------------------------------------------------------------------
// V1  (reference: models/rendering.py:75-111): one wavefront per ray, nsr_composite.h
// ---------------------------------------------------------------------------
template <int K>   // samples per lane, lane l owns samples l*K .. l*K+K-1
__global__ void __launch_bounds__(256) composite_kernel(const float* __restrict__ rgb, int rgb_stride,
                                                        const float* __restrict__ sigma, int sigma_stride,
                                                        const float* __restrict__ z, int64_t R, int N, int white,
                                                        float* __restrict__ comp_rgb, float* __restrict__ depth,
                                                        float* __restrict__ opacity, float* __restrict__ weights) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;   // whole wave exits together
  const int64_t base = r * N;
  composite_ray<K>(rgb + base * rgb_stride, rgb_stride, sigma + base * sigma_stride, sigma_stride, z + base, N, white, lane, r,
                   comp_rgb, depth, opacity, weights);
}

extern "C" int nsr_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z,
                             int64_t R, int n_samples, int white_bkgd, float* comp_rgb, float* depth,
                             float* opacity, float* weights, void* stream) {
  if (R < 0 || n_samples <= 0 || rgb_stride < 3 || sigma_stride < 1 || (white_bkgd & ~(NSR_WHITE_BKGD | NSR_SIGMA_SOFTPLUS)) != 0)
    return NSR_ERR_INVALID_ARG;
  if (n_samples > NSR_MAX_SAMPLES) return NSR_ERR_UNSUPPORTED;
  if (R == 0) return NSR_OK;
  if (!rgb || !sigma || !z) return NSR_ERR_INVALID_ARG;
  const int K = (n_samples + 63) / 64;
  const dim3 block(256), grid((unsigned)((R + 3) / 4));
  hipStream_t st = nsr_stream(stream);
#define NSR_LAUNCH_COMPOSITE(KK)                                                                              \
  hipLaunchKernelGGL(composite_kernel<KK>, grid, block, 0, st, rgb, rgb_stride, sigma, sigma_stride, z, R, \
                     n_samples, white_bkgd, comp_rgb, depth, opacity, weights)
  switch (K) {
    case 1: NSR_LAUNCH_COMPOSITE(1); break;
    case 2: NSR_LAUNCH_COMPOSITE(2); break;
    case 3: NSR_LAUNCH_COMPOSITE(3); break;
    case 4: NSR_LAUNCH_COMPOSITE(4); break;
    default: NSR_LAUNCH_COMPOSITE(8); break;
  }
#undef NSR_LAUNCH_COMPOSITE
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// ---------------------------------------------------------------------------
// S2  (reference: models/utils.py:47-95)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) resample_kernel(const float* __restrict__ rays, int stride,
                                                       const float* __restrict__ z,
                                                       const float* __restrict__ weights, int64_t R, int Nc, int Ni,
                                                       const float* __restrict__ u, float* __restrict__ z_out,
                                                       float* __restrict__ pts) {
  __shared__ float s_z[4][NSR_RS_MAX];
  __shared__ float s_cdf[4][NSR_RS_MAX];
  __shared__ float s_bins[4][NSR_RS_MAX];
  __shared__ float s_val[4][2 * NSR_RS_MAX];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wv;
  if (r >= R) return;   // wave-uniform; no block-level barrier is used below
  float* zs = s_z[wv];
  float* cdf = s_cdf[wv];
  float* bins = s_bins[wv];
  float* val = s_val[wv];
  const float eps = 1e-5f;
  const int nb = Nc - 2;            // entries of the pdf (weights[:, 1:-1])
  // stage z, and the coarse half of the merge buffer
  for (int k = lane; k < Nc; k += 64) {
    const float v = z[r * Nc + k];
    zs[k] = v;
    val[k] = v;
  }
  __builtin_amdgcn_wave_barrier();
  // bins = interval mid points (Nc-1 of them)
  for (int k = lane; k < Nc - 1; k += 64) bins[k] = __fmul_rn(0.5f, __fadd_rn(zs[k], zs[k + 1]));
  // pdf normaliser: sum of the fp32 terms, accumulated in double and rounded once (torch's
  // fp32 sum order is build dependent; the correctly rounded sum is within 1 ulp of any of them)
  double part = 0.0;
  for (int j = lane; j < nb; j += 64) part += (double)__fadd_rn(weights[r * Nc + j + 1], eps);
  const float wsum = (float)wave_sum_d(part);
  // cdf = [0, cumsum(pdf)]  (double accumulate, fp32 per element)
  double carry = 0.0;
  for (int seg = 0; seg < nb; seg += 64) {
    const int j = seg + lane;
    const float pdf = (j < nb) ? __fdiv_rn(__fadd_rn(weights[r * Nc + j + 1], eps), wsum) : 0.0f;
    const double incl = wave_scan_add_d((double)pdf, lane) + carry;
    if (j < nb) cdf[j + 1] = (float)incl;
    carry = __shfl(incl, 63, 64);
  }
  if (lane == 0) cdf[0] = 0.0f;
  __builtin_amdgcn_wave_barrier();
  // invert the cdf at u
  for (int j = lane; j < Ni; j += 64) {
    const float uj = u ? u[r * Ni + j] : nsr_linspace01(j, Ni);
    int lo = 0, hi = Nc - 1;        // searchsorted(cdf[0..Nc-2], u, right=True)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
    }
    const int below = max(lo - 1, 0), above = min(lo, nb);
    const float cb = cdf[below], ca = cdf[above];
    const float bb = bins[below], ba = bins[above];
    float denom = __fsub_rn(ca, cb);
    if (denom < eps) denom = 1.0f;
    const float zn = __fadd_rn(bb, __fmul_rn(__fdiv_rn(__fsub_rn(uj, cb), denom), __fsub_rn(ba, bb)));
    val[Nc + j] = zn;
  }
  __builtin_amdgcn_wave_barrier();
  // sort(cat([z, z_new])) by stable rank counting: every value's output slot is the
  // number of values that precede it (ties broken by position).
  const int M = Nc + Ni;
  for (int i = lane; i < M; i += 64) {
    const float v = val[i];
    int rank = 0;
    for (int j = 0; j < M; ++j) {
      const float o = val[j];
      rank += (o < v || (o == v && j < i)) ? 1 : 0;
    }
    z_out[r * M + rank] = v;
    if (pts) {
      const NsrRay q = nsr_load_ray(rays, r, stride);
      float* p = pts + (r * M + rank) * 3;
      p[0] = __fadd_rn(q.o[0], __fmul_rn(v, q.d[0]));
      p[1] = __fadd_rn(q.o[1], __fmul_rn(v, q.d[1]));
      p[2] = __fadd_rn(q.o[2], __fmul_rn(v, q.d[2]));
    }
  }
}

extern "C" int nsr_resample_along_rays(const float* rays, int ray_stride, const float* z, const float* weights,
                                       int64_t R, int n_coarse, int n_importance, const float* u, float* z_out,
                                       float* pts, void* stream) {
  if (R < 0 || n_coarse < 3 || n_importance <= 0 || !nsr_ray_stride_ok(ray_stride)) return NSR_ERR_INVALID_ARG;
  if (n_coarse > NSR_RS_MAX || n_importance > NSR_RS_MAX) return NSR_ERR_UNSUPPORTED;
  if (R == 0) return NSR_OK;
  if (!z || !weights || !z_out) return NSR_ERR_INVALID_ARG;
  if (pts && (!rays || (ray_stride == 8 && (reinterpret_cast<uintptr_t>(rays) & 15) != 0))) return NSR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, nsr_stream(stream), rays,
                     ray_stride, z, weights, R, n_coarse, n_importance, u, z_out, pts);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
