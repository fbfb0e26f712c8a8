// N3: forward pass of the refinement network (include/nsr_refine.h; reference: models/networks.py:735-990,
// MaxPoolingModel = Model_VNPCAT_Encoder + max over the reference patches + Model_VNPCAT_Decoder, eval mode).
//
// Every 3x3 convolution is im2col (NHWC, taps-major K = (ky, kx, cin)) + the fp32-MFMA GEMM of nsr_gemm.hip:
// M = images x output pixels, N = Cout, K = 9 Cin.  NHWC makes a row of the im2col matrix nine contiguous channel
// runs (coalesced float4 copies), makes the GEMM output (pixels x Cout) the next layer's activation as it is, and
// lets a GEMM write straight into a slice of a concatenated buffer (row stride = total channels), so torch.cat costs
// nothing; nn.Upsample(scale_factor=2) (nearest) is an index map (y >> 1, x >> 1) inside the next layer's im2col.
// BatchNorm2d in eval mode is an affine map per channel: folded into the packed weights and bias.
// Two arithmetic modes (include/nsr_refine.h): NSR_FP32 = explicit im2col + the fp32-MFMA GEMM; NSR_F16X3 = the
// split-fp16 GEMM with IMPLICIT im2col (the 3 x 3 gather happens while the A tile is staged, no col matrix exists;
// only the 3-channel first layer still goes through a 32-column col matrix).  In that mode every activation between
// two convolutions lives PRE-SPLIT in HBM: an fp16 "hi" plane and an fp16 "lo" plane of the same NHWC shape (the
// same 4 bytes per element as fp32), written by the producing GEMM's epilogue and staged by the consuming GEMM as
// they are -- a 3 x 3 convolution reads every activation nine times, splitting it once instead of nine times is what
// the kernel's issue slots were spent on (nsr_gemm.h: GemmF16Args::Ah / Ch).
#include <stdlib.h>
#include "nsr_common.h"
#include "nsr_gemm.h"
#include "../../include/nsr_refine.h"

using namespace nsr;

namespace {

struct Layer {
  int cin, cout, stride, bn, up, act;
};
// forward order = state_dict order (E.conv1..7, D.conv1, 2, 2_up, 3, 4, 4_up, 5, 6, 6_up, 7, 8, 9).
// Variant 0: Model_VNPCAT_Decoder (decoder inputs [.. | F_synth_i | F_max_i]); variant 1: --not_use_ref,
// Model_VNPCAT_Decoder_NoPooling (networks.py:866-945: the same layers without the F_max_i channels, i.e. D.conv1,
// D.conv3, D.conv5 and D.conv7 take 512 / 1024 / 512 / 256 input channels).
constexpr Layer kLayersV[2][NSR_REFINE_N_LAYERS] = {{
    {3, 128, 1, 0, 0, kActRelu},    {128, 128, 1, 1, 0, kActRelu}, {128, 256, 2, 1, 0, kActRelu},
    {256, 256, 1, 1, 0, kActRelu},  {256, 512, 2, 1, 0, kActRelu}, {512, 512, 1, 1, 0, kActRelu},
    {512, 512, 2, 1, 0, kActRelu},
    {1024, 512, 1, 1, 0, kActRelu}, {512, 512, 1, 1, 0, kActRelu}, {512, 512, 1, 1, 1, kActRelu},
    {1536, 512, 1, 1, 0, kActRelu}, {512, 512, 1, 1, 0, kActRelu}, {512, 256, 1, 1, 1, kActRelu},
    {768, 256, 1, 1, 0, kActRelu},  {256, 256, 1, 1, 0, kActRelu}, {256, 128, 1, 1, 1, kActRelu},
    {384, 128, 1, 1, 0, kActRelu},  {128, 128, 1, 1, 0, kActRelu}, {128, 3, 1, 0, 0, kActTanh},
}, {
    {3, 128, 1, 0, 0, kActRelu},    {128, 128, 1, 1, 0, kActRelu}, {128, 256, 2, 1, 0, kActRelu},
    {256, 256, 1, 1, 0, kActRelu},  {256, 512, 2, 1, 0, kActRelu}, {512, 512, 1, 1, 0, kActRelu},
    {512, 512, 2, 1, 0, kActRelu},
    {512, 512, 1, 1, 0, kActRelu},  {512, 512, 1, 1, 0, kActRelu}, {512, 512, 1, 1, 1, kActRelu},
    {1024, 512, 1, 1, 0, kActRelu}, {512, 512, 1, 1, 0, kActRelu}, {512, 256, 1, 1, 1, kActRelu},
    {512, 256, 1, 1, 0, kActRelu},  {256, 256, 1, 1, 0, kActRelu}, {256, 128, 1, 1, 1, kActRelu},
    {256, 128, 1, 1, 0, kActRelu},  {128, 128, 1, 1, 0, kActRelu}, {128, 3, 1, 0, 0, kActTanh},
}};
constexpr int pad32(int n) { return (n + 31) & ~31; }
constexpr int kpad(int v, int l) { return pad32(9 * kLayersV[v][l].cin); }
// the last layer (128 -> 3, tanh) is padded to 64 columns, the narrowest conv_halo_kernel shape (round 6: until then it ran on
// the staged 128-column tile, 97 % of whose MFMAs multiplied column padding, fetching every activation nine times)
constexpr bool last_tanh(int v, int l) { return kLayersV[v][l].act == kActTanh && kLayersV[v][l].cout <= 64 && (kLayersV[v][l].cin % 16) == 0; }
constexpr int npad(int v, int l) { return last_tanh(v, l) ? 64 : pad32(kLayersV[v][l].cout); }
// layers conv_halo_kernel can take (nsr_gemm_f16.hip; the shape decides at run time) carry a second, stream-ordered copy of
// their split weights (GemmF16Args::Bs)
constexpr bool streamed(int v, int l) {
  return ((kLayersV[v][l].cin % 16) == 0 && (kLayersV[v][l].cout % 128) == 0 && kLayersV[v][l].act == kActRelu && kpad(v, l) == 9 * kLayersV[v][l].cin) ||
         (last_tanh(v, l) && kpad(v, l) == 9 * kLayersV[v][l].cin);
}
// Layer 0 (3 -> 128 channels; round 6): in NSR_F16X3 mode the input is staged as pre-split NHWC planes of kL0Cin = 16 channels
// (3 real) and the layer runs on conv_halo_kernel as ONE channel chunk of nine k-steps -- no im2col matrix (it was 128 B per
// pixel written and read back, a launch of its own) and the paired half shape's overlap for a layer that is all epilogue.  Its
// stream-ordered weights (np x 9 x 16 halves x 2 planes) sit behind the layer's bias in the blob.
constexpr int kL0Cin = 16;
constexpr int64_t l0_stream_floats(int v) { return (int64_t)npad(v, 0) * 9 * kL0Cin; }      // bytes / 4
// W' (npad x kpad) | b' (npad) | stream-ordered W' (same size; streamed layers) | layer 0: its 16-channel stream
constexpr int64_t layer_floats(int v, int l) {
  return (int64_t)npad(v, l) * kpad(v, l) * (streamed(v, l) ? 2 : 1) + npad(v, l) + (l == 0 ? l0_stream_floats(v) : 0);
}
constexpr int64_t layer_offset(int v, int l) { return l == 0 ? 0 : layer_offset(v, l - 1) + layer_floats(v, l - 1); }
constexpr int64_t pack_floats(int v) { return layer_offset(v, NSR_REFINE_N_LAYERS - 1) + layer_floats(v, NSR_REFINE_N_LAYERS - 1); }
constexpr float kBnEps = 1e-5f;   // nn.BatchNorm2d default

inline int64_t align64(int64_t n) { return (n + 63) & ~(int64_t)63; }

// W'[n][(ky * 3 + kx) * cin + c] = s_n * W[n][c][ky][kx],  b'[n] = (b[n] - mean[n]) * s_n + beta[n],
// s_n = gamma[n] / sqrt(var[n] + eps)  (1 and the plain bias without a BatchNorm); padding rows / columns are zero
__global__ void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var,
                                 int cin, int cout, int kp, int np, int f16x3, float* __restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nw = (int64_t)np * kp;
  if (idx >= nw + np) return;
  if (idx < nw) {
    const int n = (int)(idx / kp), k = (int)(idx % kp);
    float v = 0.0f;
    if (n < cout && k < 9 * cin) {
      const int tap = k / cin, c = k % cin;
      const float s = gamma ? __fdiv_rn(gamma[n], sqrtf(__fadd_rn(var[n], kBnEps))) : 1.0f;
      v = __fmul_rn(s, w[((int64_t)n * cin + c) * 9 + tap]);
    }
    if (f16x3) {   // [hi halves (np x kp)] [lo halves (np x kp)] [bias]: the same number of bytes as the fp32 layout
      v *= kSplitScale;   // see nsr_gemm.h: keeps the lo half clear of fp16's subnormal floor; undone by acc_scale
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
      d16[idx] = __builtin_bit_cast(unsigned short, hi);
      d16[nw + idx] = __builtin_bit_cast(unsigned short, lo);
    } else {
      dst[idx] = v;
    }
  } else {
    const int n = (int)(idx - nw);
    float v = 0.0f;
    if (n < cout) {
      if (gamma) {
        const float s = __fdiv_rn(gamma[n], sqrtf(__fadd_rn(var[n], kBnEps)));
        v = __fadd_rn(__fmul_rn(__fsub_rn(b[n], mean[n]), s), beta[n]);
      } else {
        v = b[n];
      }
    }
    dst[idx] = v;
  }
}

// GemmF16Args::Bs of layer 0 over kL0Cin padded channels, straight from the reference's (cout, 3, 3, 3) weight (no BatchNorm on
// that layer): [column block][tap][plane][lane][8 halves], lane (i, h) = output channel 32 nb + i, channels 8 h .. 8 h + 7
__global__ void l0_stream_kernel(const float* __restrict__ w, int cout, int np, unsigned short* __restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // ((nb * 9 + tap) * 2 + plane) * 64 + lane
  if (idx >= (int64_t)(np / 32) * 9 * 128) return;
  const int lane = (int)(idx & 63), plane = (int)((idx >> 6) & 1);
  const int64_t u = idx >> 7;
  const int nb = (int)(u / 9), tap = (int)(u % 9), n = 32 * nb + (lane & 31);
  __attribute__((aligned(16))) unsigned short out[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * (lane >> 5) + e;
    const float x = (c < 3 && n < cout) ? kSplitScale * w[((int64_t)n * 3 + c) * 9 + tap] : 0.0f;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    out[e] = __builtin_bit_cast(unsigned short, plane ? lo : hi);
  }
  *reinterpret_cast<uint4*>(dst + idx * 8) = *reinterpret_cast<const uint4*>(out);
}
// the reference's NCHW fp32 input (n_img, 3, H, W) as pre-split NHWC planes of kL0Cin channels (hi | lo, 13 zero channels)
__global__ void __launch_bounds__(256) nchw3_to_planes16_kernel(const float* __restrict__ src, int64_t px_per_img, int64_t n_px,
                                                                 unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_px) return;
  const int64_t img = p / px_per_img, q = p - img * px_per_img;
  __attribute__((aligned(16))) unsigned short h[16], l[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) h[c] = l[c] = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = src[(img * 3 + c) * px_per_img + q];
    const _Float16 xh = (_Float16)x;
    const _Float16 xl = (_Float16)(x - (float)xh);
    h[c] = __builtin_bit_cast(unsigned short, xh);
    l[c] = __builtin_bit_cast(unsigned short, xl);
  }
  uint4* dh = reinterpret_cast<uint4*>(hi + p * 16);
  uint4* dl = reinterpret_cast<uint4*>(lo + p * 16);
  dh[0] = reinterpret_cast<const uint4*>(h)[0];
  dh[1] = reinterpret_cast<const uint4*>(h)[1];
  dl[0] = reinterpret_cast<const uint4*>(l)[0];
  dl[1] = reinterpret_cast<const uint4*>(l)[1];
}

// GemmF16Args::Bs from the packed (hi | lo) planes: one thread per 16 B
__global__ void stream_order_kernel(const unsigned short* __restrict__ hl, int np, int kp, int cin, unsigned short* __restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // ((nb * nks + ks) * 2 + plane) * 64 + lane
  const int nks = kp / 16;
  if (idx >= (int64_t)(np / 32) * nks * 128) return;
  const int lane = (int)(idx & 63), plane = (int)((idx >> 6) & 1);
  const int64_t u = idx >> 7;
  const int nb = (int)(u / nks), ks = (int)(u % nks), cc = ks / 9, tap = ks % 9;
  const int n = 32 * nb + (lane & 31), k = tap * cin + 16 * cc + 8 * (lane >> 5);
  const uint4 v = *reinterpret_cast<const uint4*>(hl + (int64_t)plane * np * kp + (int64_t)n * kp + k);
  *reinterpret_cast<uint4*>(dst + idx * 8) = v;
}

// im2col of a 3x3 / pad 1 convolution.  Source: NHWC with row stride `ld` (channels [0, cin) of a possibly wider
// buffer), or NCHW (first layer: the reference's input tensor); `up`: the source is read through a nearest x2 upsample.
// col (M, kp), M = n_img * Ho * Wo; one thread per (row, tap, channel quad)
template <bool NCHW>
__global__ void __launch_bounds__(256) im2col_kernel(const float* __restrict__ src, int64_t ld, int cin, int n_img, int Hs,
                                                     int Ws, int stride, int up, int Ho, int Wo, int kp,
                                                     float* __restrict__ col) {
  const int q_per_row = kp / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n_img * Ho * Wo * q_per_row;
  if (idx >= total) return;
  const int64_t m = idx / q_per_row;
  const int k0 = (int)(idx % q_per_row) * 4;
  const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), img = (int)(m / ((int64_t)Wo * Ho));
  const int Hin = up ? 2 * Hs : Hs, Win = up ? 2 * Ws : Ws;     // extent the convolution sees
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if (!NCHW && (cin & 3) == 0) {
    if (k0 < 9 * cin) {
      const int tap = k0 / cin, c = k0 % cin;
      const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
      if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
        const int sy = up ? iy >> 1 : iy, sx = up ? ix >> 1 : ix;
        const float4 t = *reinterpret_cast<const float4*>(src + (((int64_t)img * Hs + sy) * Ws + sx) * ld + c);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      if (k >= 9 * cin) continue;
      const int tap = k / cin, c = k % cin;
      const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
      if (iy < 0 || iy >= Hin || ix < 0 || ix >= Win) continue;
      const int sy = up ? iy >> 1 : iy, sx = up ? ix >> 1 : ix;
      v[e] = NCHW ? src[(((int64_t)img * cin + c) * Hs + sy) * Ws + sx]
                  : src[(((int64_t)img * Hs + sy) * Ws + sx) * ld + c];
    }
  }
  *reinterpret_cast<float4*>(col + m * kp + k0) = make_float4(v[0], v[1], v[2], v[3]);
}

// dst[(b, px)][c] = max_r src[(b * R + r, px)][c]   (torch.max over the reference patches, networks.py:980-983)
__global__ void __launch_bounds__(256) max_refs_kernel(const float* __restrict__ src, int C, int R, int64_t px_per_img,
                                                       int64_t n, float* __restrict__ dst, int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * px * C / 4
  if (idx >= n) return;
  const int c4 = (int)(idx % (C / 4));
  const int64_t bp = idx / (C / 4), b = bp / px_per_img, px = bp % px_per_img;
  float4 m = *reinterpret_cast<const float4*>(src + ((b * R) * px_per_img + px) * C + 4 * c4);
  for (int r = 1; r < R; ++r) {
    const float4 t = *reinterpret_cast<const float4*>(src + ((b * R + r) * px_per_img + px) * C + 4 * c4);
    m.x = nsr_max_nan(m.x, t.x); m.y = nsr_max_nan(m.y, t.y); m.z = nsr_max_nan(m.z, t.z); m.w = nsr_max_nan(m.w, t.w);
  }
  *reinterpret_cast<float4*>(dst + bp * ld + 4 * c4) = m;
}

// the same on pre-split activations: value order = (hi, lo) lexicographic order (|lo| <= ulp(hi) / 2)
__global__ void __launch_bounds__(256) max_refs_planes_kernel(const unsigned short* __restrict__ src, int64_t src_plane, int C,
                                                              int R, int64_t px_per_img, int64_t n,
                                                              unsigned short* __restrict__ dst, int64_t dst_plane, int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * px * C / 2
  if (idx >= n) return;
  const int c2 = (int)(idx % (C / 2));
  const int64_t bp = idx / (C / 2), b = bp / px_per_img, px = bp % px_per_img;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const int64_t o0 = ((b * R) * px_per_img + px) * C + 2 * c2;
  h2 mh = *reinterpret_cast<const h2*>(src + o0), ml = *reinterpret_cast<const h2*>(src + src_plane + o0);
  for (int r = 1; r < R; ++r) {
    const int64_t o = ((b * R + r) * px_per_img + px) * C + 2 * c2;
    const h2 th = *reinterpret_cast<const h2*>(src + o), tl = *reinterpret_cast<const h2*>(src + src_plane + o);
#pragma unroll
    for (int e = 0; e < 2; ++e)
      // NaN-propagating like torch.max (and like the fused epilogue): the first NaN met stays
      if (!(mh[e] != mh[e]) && (th[e] != th[e] || th[e] > mh[e] || (th[e] == mh[e] && tl[e] > ml[e]))) {
        mh[e] = th[e];
        ml[e] = tl[e];
      }
  }
  *reinterpret_cast<h2*>(dst + bp * ld + 2 * c2) = mh;
  *reinterpret_cast<h2*>(dst + dst_plane + bp * ld + 2 * c2) = ml;
}

// (B * H * W, 3) NHWC -> (B, 3, H, W)
__global__ void nhwc3_to_nchw_kernel(const float* __restrict__ src, int64_t px_per_img, int64_t n, float* __restrict__ dst) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * 3 * px
  if (idx >= n) return;
  const int64_t px = idx % px_per_img, c = (idx / px_per_img) % 3, b = idx / (3 * px_per_img);
  dst[idx] = src[(b * px_per_img + px) * 3 + c];
}

struct Work {
  float *col, *a, *b;                 // im2col matrix, two ping-pong activation buffers
  float *cat1, *cat3, *cat5, *cat7;   // decoder inputs [F_synth_3 | F_max_3], [x2_up | F_synth_2 | F_max_2], ...
  float *fc0, *fc1, *fc2, *fc3;       // encoder features of the reference patches, before the max
  float* rgb;                         // NHWC output of conv9
};

// `implicit`: NSR_F16X3 -- only the 3-channel first layer still materialises an im2col matrix
int64_t work_floats(int B, int R, int H, int W, Work* w, float* base, bool implicit) {
  const int64_t px0 = (int64_t)H * W, px1 = px0 / 4, px2 = px0 / 16, px3 = px0 / 64, nimg = (int64_t)B * (R > 1 ? R : 1);
  int64_t off = 0;
  auto take = [&](int64_t n) {
    float* p = base ? base + off : nullptr;
    off += align64(n);
    return p;
  };
  Work tmp;
  Work& k = w ? *w : tmp;
  // largest im2col matrices: encoder conv2 over the reference patches, decoder conv7
  int64_t col = nimg * px0 * kpad(0, implicit ? 0 : 1);
  const int64_t dec[] = {B * px3 * kpad(0, 7), B * px2 * kpad(0, 9), B * px2 * kpad(0, 10), B * px1 * kpad(0, 12), B * px1 * kpad(0, 13),
                         B * px0 * kpad(0, 15), B * px0 * kpad(0, 16), nimg * px1 * kpad(0, 3), nimg * px2 * kpad(0, 5), nimg * px1 * kpad(0, 2)};
  if (!implicit)
    for (int64_t d : dec) col = d > col ? d : col;
  k.col = take(col);
  const int64_t act = nimg * px0 * 128;   // largest plain activation: conv1 / conv2 outputs of the reference patches
  k.a = take(act);
  k.b = take(act);
  k.cat1 = take(B * px3 * 1024);
  k.cat3 = take(B * px2 * 1536);
  k.cat5 = take(B * px1 * 768);
  k.cat7 = take(B * px0 * 384);
  k.fc0 = take(nimg * px0 * 128);
  k.fc1 = take(nimg * px1 * 256);
  k.fc2 = take(nimg * px2 * 512);
  k.fc3 = take(nimg * px3 * 512);
  k.rgb = take(B * px0 * 3);
  return off;
}

#define NSR_TRY(expr)            \
  do {                           \
    const int rc_ = (expr);      \
    if (rc_ != NSR_OK) return rc_; \
  } while (0)

// An activation tensor between two layers: `p` = start of its buffer, (rows, ld) = the buffer's shape, ch0 = first
// channel of the slice this layer reads / writes (concatenated buffers).  NSR_FP32: fp32 NHWC.  NSR_F16X3: the buffer
// holds two fp16 planes of (rows, ld) halves each -- hi at (u16*)p, lo `rows * ld` halves behind it.
struct Act {
  float* p;
  int64_t rows, ld;
  int ch0;
};
inline const float* f32_of(const Act& t) { return t.p + t.ch0; }
inline unsigned short* hi_of(const Act& t) { return reinterpret_cast<unsigned short*>(t.p) + t.ch0; }
inline int64_t plane_of(const Act& t) { return t.rows * t.ld; }

// one convolution layer: src (NHWC activation; NCHW fp32 input tensor for layer 0) -> dst (NHWC slice; the last layer
// writes fp32 in both modes)
// `mx` (NSR_F16X3, implicit gather, n_img a multiple of 8): also write the maximum over each group of 8 consecutive images
// into this slice -- the GEMM then orders its rows member-fastest and reduces in its epilogue (nsr_gemm.h, GemmF16Args::group)
int conv(hipStream_t st, const float* packed, int precision, int v, int l, const Act& src, const float* src_nchw, int n_img,
         int Hs, int Ws, float* col, const Act& dst, const Act* mx = nullptr) {
  const Layer& L = kLayersV[v][l];
  const bool nchw = src_nchw != nullptr, f16 = precision == NSR_F16X3, last = l == NSR_REFINE_N_LAYERS - 1;
  const int Hin = L.up ? 2 * Hs : Hs, Win = L.up ? 2 * Ws : Ws;
  const int Ho = (Hin - 1) / L.stride + 1, Wo = (Win - 1) / L.stride + 1;   // k = 3, pad = 1
  const int kp = kpad(v, l);
  const int64_t M = (int64_t)n_img * Ho * Wo, total = M * (kp / 4);
  // layer 0 in split-fp16 mode, whole 16 x 16 blocks, planes inside 32-bit byte offsets: the conv_halo route over 16 padded
  // channels (above).  A matter of the patch SHAPE only (never of the batch: the entry points cut it, sets_per_pass).
  if (l == 0 && f16 && nchw && L.cin == 3 && L.stride == 1 && !L.up && !L.bn && (Hs % 16) == 0 && (Ws % 16) == 0 &&
      M * kL0Cin * 4 < ((int64_t)1 << 32)) {
    unsigned short* ph = reinterpret_cast<unsigned short*>(col);
    unsigned short* pl = ph + M * kL0Cin;
    hipLaunchKernelGGL(nchw3_to_planes16_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, src_nchw, (int64_t)Hs * Ws, M, ph, pl);
    NSR_CHECK_LAUNCH();
    const float* wp0 = packed + layer_offset(v, l);
    GemmF16Args a{};
    a.g.M = M; a.g.N = npad(v, l); a.g.K = 9 * kL0Cin; a.g.n_valid = L.cout; a.g.act = L.act; a.g.splits = 1;
    a.g.bias = wp0 + (int64_t)npad(v, l) * kp;
    a.g.acc_scale = kSplitInvScale;
    a.g.lda = kL0Cin; a.g.ldc = dst.ld;
    a.Bh = reinterpret_cast<const unsigned short*>(wp0);       // the staged route's planes: present, unused by conv_halo_kernel
    a.Bl = a.Bh + (int64_t)npad(v, l) * kp;
    a.ldbh = kp;
    a.Bs = reinterpret_cast<const unsigned short*>(wp0 + (int64_t)npad(v, l) * kp + npad(v, l));
    a.Ah = ph;
    a.a_plane = M * kL0Cin;
    a.conv = ConvGather{kL0Cin, Hs, Ws, Ho, Wo, 1, 0};
    a.Ch = hi_of(dst);
    a.c_plane = plane_of(dst);
    return gemm_f16x3(a, st);
  }
  const bool implicit = f16 && !nchw && (L.cin % 32) == 0;
  if (!implicit) {
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (nchw) hipLaunchKernelGGL(im2col_kernel<true>, grid, block, 0, st, src_nchw, (int64_t)0, L.cin, n_img, Hs, Ws, L.stride, L.up, Ho, Wo, kp, col);
    else hipLaunchKernelGGL(im2col_kernel<false>, grid, block, 0, st, f32_of(src), src.ld, L.cin, n_img, Hs, Ws, L.stride, L.up, Ho, Wo, kp, col);
    NSR_CHECK_LAUNCH();
  }
  const float* wp = packed + layer_offset(v, l);
  GemmArgs g{};
  g.A = col; g.lda = kp; g.B = wp; g.ldb = kp; g.C = dst.p + dst.ch0; g.ldc = dst.ld; g.bias = wp + (int64_t)npad(v, l) * kp;
  g.M = M; g.N = npad(v, l); g.K = kp; g.n_valid = L.cout; g.act = L.act; g.splits = 1;
  if (!f16) return gemm(g, st);
  GemmF16Args a{};
  a.g = g;
  a.g.acc_scale = kSplitInvScale;
  a.g.B = nullptr;
  a.Bh = reinterpret_cast<const unsigned short*>(wp);
  a.Bl = a.Bh + (int64_t)npad(v, l) * kp;
  a.ldbh = kp;
  if (streamed(v, l)) a.Bs = a.Bh + 2 * ((int64_t)npad(v, l) * kp + npad(v, l));       // behind the bias
  if (implicit) {   // gather from the pre-split NHWC activation
    a.g.A = nullptr;
    a.Ah = hi_of(src);
    a.a_plane = plane_of(src);
    a.g.lda = src.ld;
    a.conv = ConvGather{L.cin, Hs, Ws, Ho, Wo, L.stride, L.up};
  }
  if (!last) {      // ... and leave the result pre-split for the next layer
    a.g.C = nullptr;
    a.Ch = hi_of(dst);
    a.c_plane = plane_of(dst);
  }
  if (mx) {
    if (!implicit || last || (n_img % 8) != 0) return NSR_ERR_INVALID_ARG;
    a.group = 8;
    a.Mh = hi_of(*mx);
    a.m_plane = plane_of(*mx);
    a.ldm = mx->ld;
  }
  return gemm_f16x3(a, st);
}

// Model_VNPCAT_Encoder.forward (networks.py:760-774): features x2, x4, x6, x7 into the four destination slots
// mx[4] (optional): where the maxima over groups of 8 images of the four features go (the reference patches' F_max_i)
int encoder(hipStream_t st, const float* packed, int prec, int v, const Work& k, const float* x_nchw, int n_img, int H, int W,
            const Act& d0, const Act& d1, const Act& d2, const Act& d3, const Act* mx = nullptr) {
  const int64_t px0 = (int64_t)H * W;
  const Act a128{k.a, n_img * px0, 128, 0}, a256{k.a, n_img * px0 / 4, 256, 0}, a512{k.a, n_img * px0 / 16, 512, 0};
  NSR_TRY(conv(st, packed, prec, v, 0, Act{}, x_nchw, n_img, H, W, k.col, a128));
  NSR_TRY(conv(st, packed, prec, v, 1, a128, nullptr, n_img, H, W, k.col, d0, mx ? mx + 0 : nullptr));
  NSR_TRY(conv(st, packed, prec, v, 2, d0, nullptr, n_img, H, W, k.col, a256));
  NSR_TRY(conv(st, packed, prec, v, 3, a256, nullptr, n_img, H / 2, W / 2, k.col, d1, mx ? mx + 1 : nullptr));
  NSR_TRY(conv(st, packed, prec, v, 4, d1, nullptr, n_img, H / 2, W / 2, k.col, a512));
  NSR_TRY(conv(st, packed, prec, v, 5, a512, nullptr, n_img, H / 4, W / 4, k.col, d2, mx ? mx + 2 : nullptr));
  NSR_TRY(conv(st, packed, prec, v, 6, d2, nullptr, n_img, H / 4, W / 4, k.col, d3, mx ? mx + 3 : nullptr));
  return NSR_OK;
}

int max_refs(hipStream_t st, int prec, const Act& src, int C, int B, int R, int64_t px, const Act& dst) {
  if (prec == NSR_F16X3) {
    const int64_t n = (int64_t)B * px * (C / 2);
    hipLaunchKernelGGL(max_refs_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hi_of(src), plane_of(src), C,
                       R, px, n, hi_of(dst), plane_of(dst), dst.ld);
  } else {
    const int64_t n = (int64_t)B * px * (C / 4);
    hipLaunchKernelGGL(max_refs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f32_of(src), C, R, px, n,
                       dst.p + dst.ch0, dst.ld);
  }
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

}  // namespace

namespace {

int pack_weights(const float* const* t, void* packed, int precision, int v, void* stream) {
  if (!t || !packed || (reinterpret_cast<uintptr_t>(packed) & 15) != 0) return NSR_ERR_INVALID_ARG;
  if (precision != NSR_FP32 && precision != NSR_F16X3) return NSR_ERR_UNSUPPORTED;
  for (int i = 0; i < NSR_REFINE_N_TENSORS; ++i)
    if (!t[i]) return NSR_ERR_INVALID_ARG;
  float* dst = static_cast<float*>(packed);
  int ti = 0;
  for (int l = 0; l < NSR_REFINE_N_LAYERS; ++l) {
    const Layer& L = kLayersV[v][l];
    const float *w = t[ti], *b = t[ti + 1];
    const float *gamma = nullptr, *beta = nullptr, *mean = nullptr, *var = nullptr;
    ti += 2;
    if (L.bn) {
      gamma = t[ti]; beta = t[ti + 1]; mean = t[ti + 2]; var = t[ti + 3];
      ti += 4;
    }
    const int64_t n = (int64_t)npad(v, l) * kpad(v, l) + npad(v, l);
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nsr_stream(stream), w, b, gamma, beta,
                       mean, var, L.cin, L.cout, kpad(v, l), npad(v, l), precision == NSR_F16X3, dst + layer_offset(v, l));
    NSR_CHECK_LAUNCH();
    if (precision == NSR_F16X3 && l == 0) {
      if (L.bn || L.cin != 3) return NSR_ERR_UNSUPPORTED;      // the stream is made from the raw weight (cannot happen: kLayersV)
      const int64_t n16 = (int64_t)(npad(v, 0) / 32) * 9 * 128;
      unsigned short* s0 = reinterpret_cast<unsigned short*>(dst + layer_offset(v, 0) + (int64_t)npad(v, 0) * kpad(v, 0) + npad(v, 0));
      hipLaunchKernelGGL(l0_stream_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, nsr_stream(stream), w, L.cout, npad(v, 0), s0);
      NSR_CHECK_LAUNCH();
    }
    if (precision == NSR_F16X3 && streamed(v, l)) {
      const int64_t np = npad(v, l), kp = kpad(v, l), n16 = np * kp / 4;          // 16-byte pieces of both planes
      unsigned short* hl = reinterpret_cast<unsigned short*>(dst + layer_offset(v, l));
      hipLaunchKernelGGL(stream_order_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, nsr_stream(stream), hl, (int)np, (int)kp,
                         L.cin, hl + 2 * (np * kp + np));
      NSR_CHECK_LAUNCH();
    }
  }
  return NSR_OK;
}

// v = 0: x_candi (B, R, ...) given, decoder inputs [.. | F_synth_i | F_max_i]; v = 1 (--not_use_ref): no references,
// decoder inputs [.. | F_synth_i] (MaxPoolingModel.forward, networks.py:963-969)
int forward(const void* packed_v, int prec, int v, const float* x_synth, const float* x_candi, int B, int R, int H, int W, float* out,
            void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || R <= 0 || H <= 0 || W <= 0) return NSR_ERR_INVALID_ARG;
  if (prec != NSR_FP32 && prec != NSR_F16X3) return NSR_ERR_UNSUPPORTED;
  if ((H % 8) || (W % 8)) return NSR_ERR_UNSUPPORTED;
  if (B == 0) return NSR_OK;
  if (!packed_v || !x_synth || (v == 0 && !x_candi) || !out || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) != 0)
    return NSR_ERR_INVALID_ARG;
  if (workspace_bytes < nsr_refine_workspace_bytes_for(prec, B, R, H, W)) return NSR_ERR_WORKSPACE;
  const float* packed = static_cast<const float*>(packed_v);
  hipStream_t st = nsr_stream(stream);
  Work k;
  work_floats(B, R, H, W, &k, static_cast<float*>(workspace), prec == NSR_F16X3);
  const int64_t px0 = (int64_t)H * W, px1 = px0 / 4, px2 = px0 / 16, px3 = px0 / 64;
  const int64_t nref = (int64_t)B * R;
  // concatenated decoder inputs: [x_up | F_synth | F_max] (v = 0) or [x_up | F_synth] (v = 1); level 3 has no x_up
  const int w1 = v ? 512 : 1024, w3 = v ? 1024 : 1536, w5 = v ? 512 : 768, w7 = v ? 256 : 384;
  const Act cat1{k.cat1, B * px3, w1, 0}, cat3{k.cat3, B * px2, w3, 0}, cat5{k.cat5, B * px1, w5, 0}, cat7{k.cat7, B * px0, w7, 0};
  auto slice = [](Act t, int ch0) { t.ch0 = ch0; return t; };
  // encoder on the synthesised patches: features land in their decoder concat slots (F_synth_i)
  NSR_TRY(encoder(st, packed, prec, v, k, x_synth, B, H, W, slice(cat7, 128), slice(cat5, 256), slice(cat3, 512), cat1));
  if (v == 0) {
    // encoder on the B * R reference patches, then the max over the R references (F_max_i)
    const Act fc0{k.fc0, nref * px0, 128, 0}, fc1{k.fc1, nref * px1, 256, 0}, fc2{k.fc2, nref * px2, 512, 0}, fc3{k.fc3, nref * px3, 512, 0};
    const bool fused_max = prec == NSR_F16X3 && R == 8 && !nsr_dev_env("NSR_REFINE_SEPARATE_MAX");   // env: A/B runs
    if (fused_max) {
      // the reference's 8 patches per tile (llff_refine_dataset.py: num_ref_patches): the producing GEMMs reduce over them
      // in their epilogues -- the four max kernels (1.1 ms per 800 x 800 frame, 5.1 GB read at 4.5 TB/s) are gone
      const Act mx[4] = {slice(cat7, 256), slice(cat5, 512), slice(cat3, 1024), slice(cat1, 512)};
      NSR_TRY(encoder(st, packed, prec, v, k, x_candi, B * R, H, W, fc0, fc1, fc2, fc3, mx));
    } else {
      NSR_TRY(encoder(st, packed, prec, v, k, x_candi, B * R, H, W, fc0, fc1, fc2, fc3));
      NSR_TRY(max_refs(st, prec, fc0, 128, B, R, px0, slice(cat7, 256)));
      NSR_TRY(max_refs(st, prec, fc1, 256, B, R, px1, slice(cat5, 512)));
      NSR_TRY(max_refs(st, prec, fc2, 512, B, R, px2, slice(cat3, 1024)));
      NSR_TRY(max_refs(st, prec, fc3, 512, B, R, px3, slice(cat1, 512)));
    }
  }
  // Model_VNPCAT_Decoder(_NoPooling).forward (networks.py:827-857, 906-935); a / b ping-pong, the *_up layers write into
  // the next concatenated buffer
  const int h3 = H / 8, w3p = W / 8;
  auto buf = [](float* p, int64_t rows, int64_t ld) { return Act{p, rows, ld, 0}; };
  NSR_TRY(conv(st, packed, prec, v, 7, cat1, nullptr, B, h3, w3p, k.col, buf(k.a, B * px3, 512)));
  NSR_TRY(conv(st, packed, prec, v, 8, buf(k.a, B * px3, 512), nullptr, B, h3, w3p, k.col, buf(k.b, B * px3, 512)));
  NSR_TRY(conv(st, packed, prec, v, 9, buf(k.b, B * px3, 512), nullptr, B, h3, w3p, k.col, cat3));                  // upsample + conv2_up
  NSR_TRY(conv(st, packed, prec, v, 10, cat3, nullptr, B, 2 * h3, 2 * w3p, k.col, buf(k.a, B * px2, 512)));
  NSR_TRY(conv(st, packed, prec, v, 11, buf(k.a, B * px2, 512), nullptr, B, 2 * h3, 2 * w3p, k.col, buf(k.b, B * px2, 512)));
  NSR_TRY(conv(st, packed, prec, v, 12, buf(k.b, B * px2, 512), nullptr, B, 2 * h3, 2 * w3p, k.col, cat5));          // upsample + conv4_up
  NSR_TRY(conv(st, packed, prec, v, 13, cat5, nullptr, B, 4 * h3, 4 * w3p, k.col, buf(k.a, B * px1, 256)));
  NSR_TRY(conv(st, packed, prec, v, 14, buf(k.a, B * px1, 256), nullptr, B, 4 * h3, 4 * w3p, k.col, buf(k.b, B * px1, 256)));
  NSR_TRY(conv(st, packed, prec, v, 15, buf(k.b, B * px1, 256), nullptr, B, 4 * h3, 4 * w3p, k.col, cat7));          // upsample + conv6_up
  NSR_TRY(conv(st, packed, prec, v, 16, cat7, nullptr, B, H, W, k.col, buf(k.a, B * px0, 128)));
  NSR_TRY(conv(st, packed, prec, v, 17, buf(k.a, B * px0, 128), nullptr, B, H, W, k.col, buf(k.b, B * px0, 128)));
  NSR_TRY(conv(st, packed, prec, v, 18, buf(k.b, B * px0, 128), nullptr, B, H, W, k.col, buf(k.rgb, B * px0, 3)));  // conv9 + tanh
  const int64_t n = (int64_t)B * 3 * px0;
  hipLaunchKernelGGL(nhwc3_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, k.rgb, px0, n, out);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

}  // namespace

extern "C" size_t nsr_refine_packed_bytes(int precision) {
  return (precision == NSR_FP32 || precision == NSR_F16X3) ? (size_t)pack_floats(0) * sizeof(float) : 0;
}
extern "C" size_t nsr_refine_packed_bytes_noref(int precision) {
  return (precision == NSR_FP32 || precision == NSR_F16X3) ? (size_t)pack_floats(1) * sizeof(float) : 0;
}
extern "C" int nsr_refine_pack_weights(const float* const* t, void* packed, int precision, void* stream) {
  return pack_weights(t, packed, precision, 0, stream);
}
extern "C" int nsr_refine_pack_weights_noref(const float* const* t, void* packed, int precision, void* stream) {
  return pack_weights(t, packed, precision, 1, stream);
}

extern "C" size_t nsr_refine_workspace_bytes(int B, int R, int H, int W) {
  return nsr_refine_workspace_bytes_for(NSR_FP32, B, R, H, W);   // the larger of the two modes
}

extern "C" size_t nsr_refine_workspace_bytes_for(int precision, int B, int R, int H, int W) {
  if (B <= 0 || R <= 0 || H <= 0 || W <= 0 || (H % 8) || (W % 8)) return 0;
  if (precision != NSR_FP32 && precision != NSR_F16X3) return 0;
  return (size_t)work_floats(B, R, H, W, nullptr, nullptr, precision == NSR_F16X3) * sizeof(float);
}

namespace {
// Patch sets per internal pass.  conv_halo_kernel addresses its input planes with 32-bit byte offsets (nsr_gemm_f16.hip), and
// whether a layer runs there or on the staged tiles must be a matter of its SHAPE only (the two sum K in different orders: a
// patch set has to give the same bits whatever batch it arrives in).  The largest input any layer sees is the 128-channel
// full-resolution activation of the reference encoder, (hi, lo) planes of B R H W x 128 halves each: a batch is cut so that
// (2 x that) x 2 bytes stays below 2^32 -- 255 patch sets of the reference's 64 x 64 / 8-reference shape (the
// default tile batch of refine.py, 256, sat exactly ON the limit and dropped that layer to the staged kernel) -- and so that
// the widest concatenated decoder input (384 channels at full resolution) does too.
int sets_per_pass(int R, int H, int W) {
  const int64_t px0 = (int64_t)H * W, lim = ((int64_t)1 << 32) - 1;
  const int64_t by_refs = lim / (2 * 2 * 128 * px0 * (R > 1 ? R : 1)), by_cat = lim / (2 * 2 * 384 * px0);
  const int64_t cap = by_refs < by_cat ? by_refs : by_cat;
  return cap < 1 ? 1 : (cap > (1 << 20) ? (1 << 20) : (int)cap);
}
int forward_batched(const void* packed_v, int prec, int v, const float* x_synth, const float* x_candi, int B, int R, int H, int W,
                    float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || R <= 0 || H <= 0 || W <= 0) return NSR_ERR_INVALID_ARG;
  if (prec != NSR_FP32 && prec != NSR_F16X3) return NSR_ERR_UNSUPPORTED;
  if (B > 0 && workspace_bytes < nsr_refine_workspace_bytes_for(prec, B, R, H, W)) return NSR_ERR_WORKSPACE;   // the documented size, whatever the cut
  const int cap = prec == NSR_F16X3 ? sets_per_pass(R, H, W) : B;
  const int64_t img = (int64_t)3 * H * W;
  for (int b0 = 0; b0 < B || b0 == 0; b0 += cap) {
    const int nb = B - b0 < cap ? B - b0 : cap;
    const int rc = forward(packed_v, prec, v, x_synth ? x_synth + b0 * img : nullptr, x_candi ? x_candi + (int64_t)b0 * R * img : nullptr,
                           nb, R, H, W, out ? out + b0 * img : nullptr, workspace, workspace_bytes, stream);
    if (rc != NSR_OK || B == 0) return rc;
  }
  return NSR_OK;
}
}  // namespace

extern "C" int nsr_refine_forward(const void* packed_v, int prec, const float* x_synth, const float* x_candi, int B, int R, int H,
                                  int W, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return forward_batched(packed_v, prec, 0, x_synth, x_candi, B, R, H, W, out, workspace, workspace_bytes, stream);
}
extern "C" int nsr_refine_forward_noref(const void* packed_v, int prec, const float* x_synth, int B, int H, int W, float* out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  return forward_batched(packed_v, prec, 1, x_synth, nullptr, B, 1, H, W, out, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// patch tiler / gather / stitcher (data/llff_refine_dataset.py:303-340, models/refine_model.py:205-216)
// ---------------------------------------------------------------------------------------------------------------
namespace {

// one workgroup per tile: ordered compaction of the in-image warp targets, x outer / y inner, first n_ref win
__global__ void __launch_bounds__(256) tile_refs_kernel(const double* __restrict__ locs, int H, int W, int patch, int n_ref,
                                                        int tiles_y, int* __restrict__ starts, int* __restrict__ ref_starts) {
  __shared__ int wave_cnt[4];
  __shared__ int running;
  const int t = blockIdx.x, ti = t / tiles_y, tj = t % tiles_y;
  const int x0 = min(W - patch, ti * patch), y0 = min(H - patch, tj * patch);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    starts[2 * t] = x0;
    starts[2 * t + 1] = y0;
    running = 0;
  }
  for (int i = tid; i < 2 * n_ref; i += 256) ref_starts[(int64_t)t * 2 * n_ref + i] = -1;
  __syncthreads();
  const int total = patch * patch;
  for (int base = 0; base < total; base += 256) {
    const int s = base + tid;
    bool ok = false;
    int rx = 0, ry = 0;
    if (s < total) {
      const int m = x0 + s / patch, n = y0 + s % patch;
      const double lx = locs[((int64_t)n * W + m) * 3], ly = locs[((int64_t)n * W + m) * 3 + 1];
      ok = lx >= 0.0 && lx < (double)W && ly >= 0.0 && ly < (double)H;
      rx = min(W - patch, (int)lx);
      ry = min(H - patch, (int)ly);
    }
    const unsigned long long mask = __ballot(ok);
    if (lane == 0) wave_cnt[wave] = __popcll(mask);
    __syncthreads();
    int before = running;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    before += __popcll(mask & ((1ull << lane) - 1ull));
    if (ok && before < n_ref) {
      ref_starts[((int64_t)t * n_ref + before) * 2] = rx;
      ref_starts[((int64_t)t * n_ref + before) * 2 + 1] = ry;
    }
    __syncthreads();
    if (tid == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
    if (running >= n_ref) break;   // uniform: every thread reads the same shared value
  }
}

// idx over n * (1 + n_ref) * 3 * patch * patch: slot 0 = the SR tile, slots 1.. = its reference patches
__global__ void __launch_bounds__(256) gather_patches_kernel(const float* __restrict__ sr_img, const float* __restrict__ ref_img,
                                                             int H, int W, int patch, int n_ref, const int* __restrict__ starts,
                                                             const int* __restrict__ ref_starts, int64_t total,
                                                             float* __restrict__ sr_patch, float* __restrict__ ref_patches) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int pp = patch * patch;
  const int px = (int)(idx % pp), c = (int)((idx / pp) % 3), slot = (int)((idx / (3 * pp)) % (1 + n_ref));
  const int t = (int)(idx / ((int64_t)3 * pp * (1 + n_ref)));
  const int dy = px / patch, dx = px % patch;
  int x = starts[2 * t], y = starts[2 * t + 1];
  const float* src = sr_img;
  if (slot > 0) {
    const int rx = ref_starts[((int64_t)t * n_ref + slot - 1) * 2], ry = ref_starts[((int64_t)t * n_ref + slot - 1) * 2 + 1];
    if (rx >= 0) { x = rx; y = ry; src = ref_img; }
  }
  const float v = src[((int64_t)c * H + y + dy) * W + x + dx];
  if (slot == 0) sr_patch[((int64_t)t * 3 + c) * pp + px] = v;
  else ref_patches[(((int64_t)t * n_ref + slot - 1) * 3 + c) * pp + px] = v;
}

// image[c][y][x] = the LAST tile (in order) that covers (x, y), 0 if none
__global__ void __launch_bounds__(256) stitch_kernel(const float* __restrict__ patches, const int* __restrict__ starts, int n,
                                                     int patch, int H, int W, float* __restrict__ image) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)3 * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), c = (int)(idx / ((int64_t)W * H));
  float v = 0.0f;
  for (int t = n - 1; t >= 0; --t) {
    const int x0 = starts[2 * t], y0 = starts[2 * t + 1];
    if (x >= x0 && x < x0 + patch && y >= y0 && y < y0 + patch) {
      v = patches[(((int64_t)t * 3 + c) * patch + (y - y0)) * patch + (x - x0)];
      break;
    }
  }
  image[idx] = v;
}

}  // namespace

extern "C" int nsr_refine_tile(const double* locs, int H, int W, int patch, int n_ref, int* starts, int* ref_starts,
                               void* stream) {
  if (H <= 0 || W <= 0 || patch <= 0 || patch > H || patch > W || n_ref <= 0 || !locs || !starts || !ref_starts)
    return NSR_ERR_INVALID_ARG;
  const int tiles_x = (W + patch - 1) / patch, tiles_y = (H + patch - 1) / patch;
  hipLaunchKernelGGL(tile_refs_kernel, dim3(tiles_x * tiles_y), dim3(256), 0, nsr_stream(stream), locs, H, W, patch, n_ref,
                     tiles_y, starts, ref_starts);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_refine_gather(const float* sr_img, const float* ref_img, int H, int W, int patch, int n_ref,
                                 const int* starts, const int* ref_starts, int n, float* sr_patch, float* ref_patches,
                                 void* stream) {
  if (H <= 0 || W <= 0 || patch <= 0 || n_ref <= 0 || n < 0) return NSR_ERR_INVALID_ARG;
  if (n == 0) return NSR_OK;
  if (!sr_img || !ref_img || !starts || !ref_starts || !sr_patch || !ref_patches) return NSR_ERR_INVALID_ARG;
  const int64_t total = (int64_t)n * (1 + n_ref) * 3 * patch * patch;
  hipLaunchKernelGGL(gather_patches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nsr_stream(stream), sr_img,
                     ref_img, H, W, patch, n_ref, starts, ref_starts, total, sr_patch, ref_patches);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_refine_stitch(const float* patches, const int* starts, int n, int patch, int H, int W, float* image,
                                 void* stream) {
  if (H <= 0 || W <= 0 || patch <= 0 || n < 0 || !image || (n > 0 && (!patches || !starts))) return NSR_ERR_INVALID_ARG;
  const int64_t total = (int64_t)3 * H * W;
  hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nsr_stream(stream), patches, starts, n,
                     patch, H, W, image);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
