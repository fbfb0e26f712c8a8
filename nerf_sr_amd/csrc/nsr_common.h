// Shared host/device helpers of libnsr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nsr.h"

#define NSR_WAVE 64
// cross-TU helpers that are not part of the C ABI (include/nsr.h) stay out of the dynamic symbol table
#define NSR_INTERNAL __attribute__((visibility("hidden")))

#define NSR_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t nsr_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Development switches (A/B runs on one box: NSR_GEMM_TILE / NSR_GEMM_TK / NSR_GEMM_FULLN / NSR_REFINE_SEPARATE_MAX) are
// environment reads, and getenv is not safe against a concurrent setenv: the release library does not contain them
// (include/nsr.h promises no ambient process state; tests/test_abi.py checks that `getenv` is not even imported); build
// with -DNSR_DEV_SWITCHES to get them back.  The training step's path is an argument (NSR_F16X3_GEMM, include/nsr_train.h).
#ifdef NSR_DEV_SWITCHES
static inline const char* nsr_dev_env(const char* name) { return getenv(name); }
#else
static inline const char* nsr_dev_env(const char*) { return nullptr; }
#endif

// ---- tail of every packed weight blob (include/nsr.h "numerics status word"): 16 bytes behind the 16-byte-aligned
// payload of the precision's own layout: word 0 = sticky NSR_FLAG_* status, word 1 = colour-head options
// (bit 0: --gamma_correct, models/nerf_downX_model.py:271-276; bit 1: --color_activation none, models/networks.py:173-176),
// words 2, 3 reserved.
constexpr size_t kBlobTailBytes = 16;
constexpr unsigned kOptGamma = 1u, kOptColorNone = 2u;   // = NSR_OPT_* of include/nsr.h
static inline size_t nsr_blob_tail_offset(size_t payload_bytes) { return (payload_bytes + 15) & ~(size_t)15; }
NSR_INTERNAL size_t nsr_payload_bytes(int precision);   // nsr_mlp.hip
static inline unsigned* nsr_blob_tail(const void* packed_dev, int precision) {
  return reinterpret_cast<unsigned*>(const_cast<char*>(static_cast<const char*>(packed_dev)) +
                                     nsr_blob_tail_offset(nsr_payload_bytes(precision)));
}
// what the MLP kernels get: the tail (null: the training step's private blob has none).  Flags are raised with one
// atomic per offending lane (never on a healthy network); the option word is read by the kernel itself, so setting
// an option is an ordinary stream-ordered write.
struct NsrTail {
  unsigned* w;   // null or the 4-word tail
};
__device__ __forceinline__ void nsr_raise(const NsrTail& t, unsigned flags) {
  if (flags != 0u && t.w) atomicOr(t.w, flags);
}
__device__ __forceinline__ unsigned nsr_opts(const NsrTail& t) { return t.w ? __builtin_nontemporal_load(t.w + 1) : 0u; }
// the colour head's activation on the pre-activation s: nn.Sigmoid, or nn.Identity under --color_activation none
// (models/networks.py:173-180), then --gamma_correct's pow (nerf_downX_model.py:271-276; nsr_gamma below)
__device__ __forceinline__ float nsr_colour_activation(float s, unsigned opts) {
  return (opts & kOptColorNone) ? s : 1.0f / (1.0f + expf(-s));
}
// torch.max over the reference patches propagates NaN (models/networks.py:980-983); fmaxf drops it
__device__ __forceinline__ float nsr_max_nan(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }
// torch.relu keeps NaN (networks.py: nn.ReLU after every BatchNorm); fmaxf(NaN, 0) = 0 would hide a diverged feature
// (x <= 0: -0.0 becomes +0.0, so the result is +0, positive, +inf or NaN -- see nsr_max_relu)
#ifdef NSR_ABL_RELU_FMAX   // ablation: what keeping NaN costs (profiles/r4_refine_halo.txt)
__device__ __forceinline__ float nsr_relu_nan(float x) { return fmaxf(x, 0.0f); }
#else
__device__ __forceinline__ float nsr_relu_nan(float x) { return (x <= 0.0f) ? 0.0f : x; }
#endif
// NaN-propagating maximum of two nsr_relu_nan results in one instruction: on +0 / positive / +inf the unsigned order of the bit
// patterns is the numeric order, and a NaN of either sign is above +inf.  (With the general nsr_max_nan the compiler wraps every
// maximum of the epilogue in an EXEC-mask branch: 4,047 -> 8,089 instructions in conv_halo_kernel's grouped epilogue.)
__device__ __forceinline__ float nsr_max_relu(float a, float b) {
  const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  return __builtin_bit_cast(float, __builtin_elementwise_max(ua, ub));      // v_max_u32
}
__device__ __forceinline__ bool nsr_finite(float x) { return fabsf(x) <= 3.402823466e38f; }   // false for inf and NaN
// sigma_activation == 'softplus' (models/rendering.py:72): log(1 + exp(x - 1)), fp32 operation by operation
__device__ __forceinline__ float nsr_softplus_density(float x) { return logf(__fadd_rn(1.0f, expf(__fsub_rn(x, 1.0f)))); }
// --gamma_correct: out_rgbs = pow(out_rgbs, 1 / 2.2) on the per-sample colours (nerf_downX_model.py:271-276)
__device__ __forceinline__ float nsr_gamma(float c) { return powf(c, 1.0f / 2.2f); }

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

// ---- sin and cos of the positional encodings (arguments 2^k * x, |arg| < 2^15).  Three-constant Cody-Waite
// reduction by pi/2 carried by FMAs (products exact inside the fma, so this is immune to -ffp-contract), then
// the classic degree-7 / degree-8 minimax kernels on [-pi/4, pi/4]: abs error < 1.5e-7 over the whole range,
// ~28 instructions per pair against ~120 for the device library's full-range sincosf (which also carries a
// Payne-Hanek branch the encodings can never take).  21 pairs per sample point make this 7 % of the fused
// split-fp16 kernel's time.
__device__ __forceinline__ void nsr_sincos(float x, float& sn, float& cs) {
  const float k = rintf(__fmul_rn(x, 0.636619772367581343f));       // x * 2/pi
  float r = fmaf(k, -1.57079625129699707031f, x);                     // pi/2 = 0x3FC90FDA + 0x33A22168 + 0x27C234C4
  r = fmaf(k, -7.54978941586159635335e-08f, r);
  r = fmaf(k, -5.39030252995776476554e-15f, r);
  const int q = (int)k;
  const float r2 = __fmul_rn(r, r);
  float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fmaf(sp, r2, -1.6666654611e-1f);
  const float s0 = fmaf(__fmul_rn(sp, r2), r, r);                      // r + r^3 * S(r^2)
  float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fmaf(cp, r2, 4.166664568298827e-2f);
  const float c0 = fmaf(__fmul_rn(cp, r2), r2, fmaf(-0.5f, r2, 1.0f));  // 1 - r^2/2 + r^4 * C(r^2)
  const float ss = (q & 1) ? c0 : s0;
  const float cc = (q & 1) ? s0 : c0;
  sn = (q & 2) ? -ss : ss;
  cs = ((q + 1) & 2) ? -cc : cc;
}

// ---- ray records.  stride 8: [o(3), d(3), near, far] (nerf_downX: the view direction that is encoded IS d,
// models/nerf_downX_model.py:282-286); stride 11: the vanilla model's rows with a separate view direction in
// columns 8:11 (models/nerf_model.py:209-213, data/llff_dataset.py:337-341).
struct NsrRay {
  float o[3], d[3], near_, far_, v[3];
};
__device__ __forceinline__ NsrRay nsr_load_ray(const float* __restrict__ rays, int64_t r, int stride) {
  NsrRay q;
  const float* p = rays + r * stride;
  if (stride == 8) {
    const float4 a = reinterpret_cast<const float4*>(p)[0];
    const float4 b = reinterpret_cast<const float4*>(p)[1];
    q.o[0] = a.x; q.o[1] = a.y; q.o[2] = a.z; q.d[0] = a.w; q.d[1] = b.x; q.d[2] = b.y; q.near_ = b.z; q.far_ = b.w;
    q.v[0] = q.d[0]; q.v[1] = q.d[1]; q.v[2] = q.d[2];
  } else {
    q.o[0] = p[0]; q.o[1] = p[1]; q.o[2] = p[2]; q.d[0] = p[3]; q.d[1] = p[4]; q.d[2] = p[5];
    q.near_ = p[6]; q.far_ = p[7]; q.v[0] = p[8]; q.v[1] = p[9]; q.v[2] = p[10];
  }
  return q;
}
static inline bool nsr_ray_stride_ok(int stride) { return stride == 8 || stride == 11; }

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
