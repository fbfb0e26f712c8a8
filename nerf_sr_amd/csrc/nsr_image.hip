// N4: LR-target construction (include/nsr_image.h; reference: data/llff_downX_dataset.py:312-329 on top of Pillow's
// 8-bit LANCZOS resampler).  HBM-bound byte work: one thread per output byte, the taps of one output sample are a
// short contiguous (horizontal pass) or strided (vertical pass) run of the source; the fixed-point weights of a
// sample are shared by all channels / lines and come from L2.  Integer arithmetic only: bit-identical to Pillow.
#include <math.h>
#include "nsr_common.h"
#include "../../include/nsr_image.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

double lanczos(double x) {
  if (-3.0 <= x && x < 3.0) {
    auto sinc = [](double t) {
      if (t == 0.0) return 1.0;
      t = t * M_PI;
      return sin(t) / t;
    };
    return sinc(x) * sinc(x / 3.0);
  }
  return 0.0;
}

// dst element (o, line, c): o = output sample along the resampled axis; src strides in bytes
__global__ void __launch_bounds__(256) resample_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            const int32_t* __restrict__ bounds,
                                                            const int32_t* __restrict__ kk, int ksize, int64_t total,
                                                            int out_size, int inner, int64_t src_step,
                                                            int64_t src_line, int64_t dst_step, int64_t dst_line) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  // idx enumerates dst in memory order for both axes: (line_outer, o, inner) with inner = bytes that share a sample
  const int in = (int)(idx % inner);
  const int64_t t = idx / inner;
  const int o = (int)(t % out_size);
  const int64_t line = t / out_size;
  const int x0 = bounds[2 * o], n = bounds[2 * o + 1];
  const int32_t* k = kk + (int64_t)o * ksize;
  const uint8_t* p = src + line * src_line + (int64_t)x0 * src_step + in;
  int acc = 1 << (kPrecisionBits - 1);
  for (int x = 0; x < n; ++x) acc += (int)p[(int64_t)x * src_step] * k[x];
  acc >>= kPrecisionBits;
  dst[line * dst_line + (int64_t)o * dst_step + in] = (uint8_t)min(max(acc, 0), 255);
}

__global__ void __launch_bounds__(256) image_to_targets_kernel(const uint8_t* __restrict__ img, int H, int W, int s,
                                                               float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over the OUTPUT (N_lr, s*s, 3)
  const int64_t total = (int64_t)H * W * 3;
  if (idx >= total) return;
  const int c = (int)(idx % 3);
  const int64_t r = idx / 3;
  const int s2 = s * s, w_lr = W / s;
  const int64_t lr = r / s2;
  const int sub = (int)(r % s2);
  const int py = (int)(lr / w_lr) * s + sub / s, px = (int)(lr % w_lr) * s + sub % s;
  out[idx] = __fdiv_rn((float)img[((int64_t)py * W + px) * 3 + c], 255.0f);
}

// Pillow's RGBA <-> premultiplied "RGBa" conversions (src/libImaging/Convert.c rgbA2rgba / rgba2rgbA), one thread per pixel
__global__ void __launch_bounds__(256) premultiply_kernel(const uint8_t* src, uint8_t* dst,   // src == dst allowed
                                                          int64_t n_px, int inverse) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_px) return;
  const uchar4 p = reinterpret_cast<const uchar4*>(src)[i];
  const unsigned a = p.w;
  unsigned c[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (!inverse) {
      const unsigned t = c[k] * a + 128u;                 // MULDIV255
      c[k] = ((t >> 8) + t) >> 8;
    } else if (a != 0u && a != 255u) {
      const unsigned q = (255u * c[k]) / a;
      c[k] = q > 255u ? 255u : q;
    }
  }
  reinterpret_cast<uchar4*>(dst)[i] = make_uchar4((uint8_t)c[0], (uint8_t)c[1], (uint8_t)c[2], p.w);
}

// ToTensor + blend onto white + regroup of an RGBA image: rgb / 255 * (a / 255) + (1 - a / 255), separate fp32 operations
// like the dataset's tensor expression (this unit is compiled with -ffp-contract=off)
__global__ void __launch_bounds__(256) image_to_targets_rgba_kernel(const uint8_t* __restrict__ img, int H, int W, int s,
                                                                    float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over the OUTPUT (N_lr, s*s, 3)
  const int64_t total = (int64_t)H * W * 3;
  if (idx >= total) return;
  const int c = (int)(idx % 3);
  const int64_t r = idx / 3;
  const int s2 = s * s, w_lr = W / s;
  const int64_t lr = r / s2;
  const int sub = (int)(r % s2);
  const int py = (int)(lr / w_lr) * s + sub / s, px = (int)(lr % w_lr) * s + sub % s;
  const uint8_t* p = img + ((int64_t)py * W + px) * 4;
  const float a = __fdiv_rn((float)p[3], 255.0f);
  out[idx] = __fadd_rn(__fmul_rn(__fdiv_rn((float)p[c], 255.0f), a), __fsub_rn(1.0f, a));
}

}  // namespace

extern "C" int nsr_rgba_premultiply_u8(const uint8_t* src, int64_t n_px, int inverse, uint8_t* dst, void* stream) {
  if (n_px < 0) return NSR_ERR_INVALID_ARG;
  if (n_px == 0) return NSR_OK;
  if (!src || !dst || (reinterpret_cast<uintptr_t>(src) & 3) || (reinterpret_cast<uintptr_t>(dst) & 3)) return NSR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(premultiply_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, nsr_stream(stream), src, dst, n_px,
                     inverse ? 1 : 0);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_image_to_targets_rgba(const uint8_t* img, int H, int W, int s, float* out, void* stream) {
  if (H < 0 || W < 0 || s <= 0 || (H > 0 && W > 0 && (H % s != 0 || W % s != 0))) return NSR_ERR_INVALID_ARG;
  const int64_t total = (int64_t)H * W * 3;
  if (total == 0) return NSR_OK;
  if (!img || !out) return NSR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(image_to_targets_rgba_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nsr_stream(stream), img,
                     H, W, s, out);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_lanczos_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(3.0 * filterscale) * 2 + 1;
}

extern "C" int nsr_lanczos_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk) {
  if (in_size <= 0 || out_size <= 0 || !bounds || !kk) return NSR_ERR_INVALID_ARG;
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 3.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  double w[1024];
  if (ksize > 1024) return NSR_ERR_UNSUPPORTED;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      w[x] = lanczos((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    int32_t* k = kk + (int64_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
      double v = 0.0;
      if (x < xmax) v = (ww != 0.0) ? w[x] / ww : w[x];
      k[x] = v < 0 ? (int32_t)(-0.5 + v * (1 << kPrecisionBits)) : (int32_t)(0.5 + v * (1 << kPrecisionBits));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return NSR_OK;
}

extern "C" int nsr_resample_pass_u8(const uint8_t* src, int H, int W, int C, int axis, int out_size,
                                    const int32_t* bounds_dev, const int32_t* kk_dev, int ksize, uint8_t* dst, void* stream) {
  if (H < 0 || W < 0 || C <= 0 || out_size < 0 || ksize <= 0 || (axis != 0 && axis != 1)) return NSR_ERR_INVALID_ARG;
  const int64_t total = axis == 1 ? (int64_t)H * out_size * C : (int64_t)out_size * W * C;
  if (total == 0) return NSR_OK;
  if (!src || !dst || !bounds_dev || !kk_dev || H == 0 || W == 0) return NSR_ERR_INVALID_ARG;
  const int threads = 256;
  const dim3 grid((unsigned)((total + threads - 1) / threads));
  if (axis == 1)   // rows: lines = H, a sample = C bytes, taps C bytes apart
    hipLaunchKernelGGL(resample_pass_kernel, grid, dim3(threads), 0, nsr_stream(stream), src, dst, bounds_dev, kk_dev, ksize,
                       total, out_size, C, (int64_t)C, (int64_t)W * C, (int64_t)C, (int64_t)out_size * C);
  else             // columns: one "line", a sample = a whole row of W * C bytes, taps one row apart
    hipLaunchKernelGGL(resample_pass_kernel, grid, dim3(threads), 0, nsr_stream(stream), src, dst, bounds_dev, kk_dev, ksize,
                       total, out_size, W * C, (int64_t)W * C, (int64_t)0, (int64_t)W * C, (int64_t)0);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_image_to_targets(const uint8_t* img, int H, int W, int s, float* out, void* stream) {
  if (H < 0 || W < 0 || s <= 0 || (H > 0 && W > 0 && (H % s != 0 || W % s != 0))) return NSR_ERR_INVALID_ARG;
  const int64_t total = (int64_t)H * W * 3;
  if (total == 0) return NSR_OK;
  if (!img || !out) return NSR_ERR_INVALID_ARG;
  hipLaunchKernelGGL(image_to_targets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nsr_stream(stream), img,
                     H, W, s, out);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
