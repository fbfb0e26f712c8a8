// M1 / D2: the NeRF MLP (VanillaMLP.forward, models/networks.py:182-226) fused
// with cast_rays + both positional encodings (render_rays,
// models/nerf_downX_model.py:260-278).  fp32 parity path:
// v_mfma_f32_32x32x2_f32, exact fp32 products, fp32 accumulation.
//
// Structure (see nsr_mlp_layout.h for the fragment algebra):
//   * workgroup = 4 waves (one per SIMD), wave = 32 sample points, tile = 128 points;
//   * activations never leave registers: the D fragment of layer L is the B
//     operand of layer L+1 (transposed evaluation, weights as the A operand);
//   * the 2.27 MiB weight stream of one net is DMA'd global->LDS
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction) in 32 KiB chunks
//     through a 2-deep ring; each chunk feeds 128 MFMAs (8192 cycles) per wave,
//     one workgroup barrier per chunk;
//   * sigma (256->1) and rgb (128->3) heads are VALU dot products on the
//     register-resident activations; bias enters as the accumulator init.
#include "nsr_common.h"
#include "nsr_mlp_layout.h"
#include "nsr_composite.h"

using namespace nsr;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// weight packing (device side gather into the fragment-ordered stream)
// ---------------------------------------------------------------------------
struct PackPtrs {
  const float* p[NSR_N_STATE_TENSORS];
};

__device__ __forceinline__ int tensor_ld(int tensor) {
  switch (tensor) {
    case 0: return kPosCh;
    case 8: return kWidth + kPosCh;
    case 18: return kWidth + kDirCh;
    default: return kWidth;
  }
}

__global__ void __launch_bounds__(256) pack_fp32_kernel(PackPtrs w, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kStreamFloats + kAuxFloats) return;
  float v = 0.0f;
  if (idx < kStreamPieces * 256) {
    const int piece = idx >> 8, lane = (idx & 255) >> 2, j = idx & 3;
    int s = 0;
#pragma unroll
    for (int i = 1; i < kNumSegments; ++i)
      if (piece >= kSegmentsDev[i].piece0) s = i;
    const Segment seg = kSegmentsDev[s];
    const int local = piece - seg.piece0;
    const int slab = local / seg.nb, nb = local % seg.nb;
    const int t = 4 * slab + j, h = lane >> 5, n = 32 * nb + (lane & 31);
    const int col = seg_col(seg.src, t, h);
    if (col != kPad) v = w.p[seg.tensor][n * tensor_ld(seg.tensor) + seg.col0 + col];
  } else if (idx >= kStreamFloats) {
    const int a = idx - kStreamFloats;
    if (a < kAuxBiasFinal) v = w.p[2 * (a >> 8) + 1][a & 255];
    else if (a < kAuxBiasDir) v = w.p[17][a - kAuxBiasFinal];
    else if (a < kAuxSigmaW) v = w.p[19][a - kAuxBiasDir];
    else if (a < kAuxRgbW) v = w.p[20][a - kAuxSigmaW];
    else if (a < kAuxSigmaB) v = w.p[22][a - kAuxRgbW];
    else if (a < kAuxRgbB) v = w.p[21][0];
    else if (a < kAuxRgbB + 3) v = w.p[23][a - kAuxRgbB];
  }
  out[idx] = v;
}

// split-fp16 kernels (nsr_mlp_f16.hip)
extern "C" NSR_INTERNAL size_t nsr_f16x3_packed_bytes(void);
extern "C" NSR_INTERNAL int nsr_f16x3_pack(const float* const* w, void* packed_dev, void* stream);
extern "C" NSR_INTERNAL int nsr_f16x3_mlp_forward(const void* packed, const float* x, int64_t P, int sigma_only, float* out,
                                     unsigned* tail, void* stream);
extern "C" NSR_INTERNAL int nsr_f16x3_render_rays(const void* packed, const float* rays, int ray_stride, const float* z, int64_t R,
                                     int N, float* out, unsigned* tail, void* stream);

// single 16-bit operand paths (nsr_mlp_h1.hip); bf = 1: bf16, 0: fp16
extern "C" NSR_INTERNAL size_t nsr_h1_packed_bytes(void);
extern "C" NSR_INTERNAL int nsr_h1_pack(int bf, const float* const* w, void* packed_dev, void* stream);
extern "C" NSR_INTERNAL int nsr_h1_mlp_forward(int bf, const void* packed, const float* x, int64_t P, int sigma_only,
                                               float* out, unsigned* tail, void* stream);
extern "C" NSR_INTERNAL int nsr_h1_render_rays(int bf, const void* packed, const float* rays, int ray_stride,
                                               const float* z, int64_t R, int N, float* out, unsigned* tail, void* stream);
static inline bool precision_built(int precision) {
  return precision == NSR_FP32 || precision == NSR_F16X3 || precision == NSR_BF16 || precision == NSR_F16;
}
static inline bool precision_h1(int precision) { return precision == NSR_BF16 || precision == NSR_F16; }

NSR_INTERNAL size_t nsr_payload_bytes(int precision) {
  if (precision == NSR_FP32) return sizeof(float) * (size_t)(kStreamFloats + kAuxFloats);
  if (precision == NSR_F16X3) return nsr_f16x3_packed_bytes();
  if (precision_h1(precision)) return nsr_h1_packed_bytes();
  return 0;
}

extern "C" size_t nsr_packed_weights_bytes(int precision) {
  const size_t payload = nsr_payload_bytes(precision);
  return payload ? nsr_blob_tail_offset(payload) + kBlobTailBytes : 0;
}

// Range check of the 24 tensors the blob is made from (include/nsr.h, nsr_pack_weights); the tail is cleared by a
// memset enqueued in front of this kernel.  limit = largest |w| the precision's operand format carries (weights only;
// biases stay fp32 everywhere and only have to be finite).
struct CheckPtrs {
  const float* p[NSR_N_STATE_TENSORS];
  int n[NSR_N_STATE_TENSORS];
};
__global__ void __launch_bounds__(256) check_weights_kernel(CheckPtrs w, float limit, unsigned* tail) {
  bool bad = false;
  for (int t = blockIdx.y; t < NSR_N_STATE_TENSORS; t += gridDim.y) {
    const float lim = (t & 1) ? 3.402823466e38f : limit;      // odd entries of the state_dict are the biases
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w.n[t]; i += gridDim.x * blockDim.x)
      bad |= !(fabsf(w.p[t][i]) <= lim);                       // also true for NaN
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(tail, NSR_FLAG_WEIGHT_RANGE);
}

static const int kStateTensorSizes[NSR_N_STATE_TENSORS] = {
    kWidth * kPosCh, kWidth, kWidth * kWidth, kWidth, kWidth * kWidth, kWidth, kWidth * kWidth, kWidth,
    kWidth * (kWidth + kPosCh), kWidth, kWidth * kWidth, kWidth, kWidth * kWidth, kWidth, kWidth * kWidth, kWidth,
    kWidth * kWidth, kWidth, (kWidth / 2) * (kWidth + kDirCh), kWidth / 2, kWidth, 1, 3 * (kWidth / 2), 3};

// largest weight magnitude whose operand encoding is still finite: fp16 rounds to inf from 65,520 on
static float precision_weight_limit(int precision) {
  if (precision == NSR_F16X3) return 65519.996f / 64.0f;   // the stream carries 2^6 w (nsr_f16x3_core.h, kWScale)
  if (precision == NSR_F16) return 65519.996f;
  return 3.402823466e38f;
}

// both networks of a training step in ONE launch (round 6: the step's launch count; blockIdx.z = network)
__global__ void __launch_bounds__(256) check_weights2_kernel(CheckPtrs w0, CheckPtrs w1, float limit, unsigned* tail) {
  const CheckPtrs& w = blockIdx.z ? w1 : w0;
  bool bad = false;
  for (int t = blockIdx.y; t < NSR_N_STATE_TENSORS; t += gridDim.y) {
    const float lim = (t & 1) ? 3.402823466e38f : limit;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < w.n[t]; i += gridDim.x * blockDim.x)
      bad |= !(fabsf(w.p[t][i]) <= lim);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(tail, NSR_FLAG_WEIGHT_RANGE);
}
extern "C" NSR_INTERNAL int nsr_check_weights_range2(const float* const* w0, const float* const* w1, int precision, unsigned* word, void* stream) {
  CheckPtrs a, b;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w0[i] || !w1[i]) return NSR_ERR_INVALID_ARG;
    a.p[i] = w0[i]; b.p[i] = w1[i];
    a.n[i] = b.n[i] = kStateTensorSizes[i];
  }
  hipLaunchKernelGGL(check_weights2_kernel, dim3(16, NSR_N_STATE_TENSORS, 2), dim3(256), 0, nsr_stream(stream), a, b,
                     precision_weight_limit(precision), word);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// the weight-range check on its own (the training step runs it on the weights it re-packs every iteration)
extern "C" NSR_INTERNAL int nsr_check_weights_range(const float* const* w, int precision, unsigned* word, void* stream) {
  CheckPtrs cp;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w[i]) return NSR_ERR_INVALID_ARG;
    cp.p[i] = w[i];
    cp.n[i] = kStateTensorSizes[i];
  }
  hipLaunchKernelGGL(check_weights_kernel, dim3(16, NSR_N_STATE_TENSORS), dim3(256), 0, nsr_stream(stream), cp,
                     precision_weight_limit(precision), word);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_pack_weights_async(const float* const* w, void* packed_dev, int precision, void* stream) {
  if (!w || !packed_dev) return NSR_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(packed_dev) & 15) != 0) return NSR_ERR_INVALID_ARG;
  if (!precision_built(precision)) return NSR_ERR_UNSUPPORTED;
  CheckPtrs cp;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w[i]) return NSR_ERR_INVALID_ARG;
    cp.p[i] = w[i];
    cp.n[i] = kStateTensorSizes[i];
  }
  unsigned* tail = nsr_blob_tail(packed_dev, precision);
  if (hipMemsetAsync(tail, 0, kBlobTailBytes, nsr_stream(stream)) != hipSuccess) return NSR_ERR_LAUNCH;
  hipLaunchKernelGGL(check_weights_kernel, dim3(16, NSR_N_STATE_TENSORS), dim3(256), 0, nsr_stream(stream), cp,
                     precision_weight_limit(precision), tail);
  NSR_CHECK_LAUNCH();
  if (precision == NSR_F16X3) return nsr_f16x3_pack(w, packed_dev, stream);
  if (precision_h1(precision)) return nsr_h1_pack(precision == NSR_BF16, w, packed_dev, stream);
  PackPtrs pp;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) pp.p[i] = w[i];
  const int total = kStreamFloats + kAuxFloats;
  hipLaunchKernelGGL(pack_fp32_kernel, dim3((total + 255) / 256), dim3(256), 0, nsr_stream(stream), pp,
                     static_cast<float*>(packed_dev));
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_weights_status(const void* packed_dev, int precision, int clear, unsigned* flags_out, void* stream) {
  if (!packed_dev || !flags_out) return NSR_ERR_INVALID_ARG;
  if (!precision_built(precision)) return NSR_ERR_UNSUPPORTED;
  unsigned* tail = nsr_blob_tail(packed_dev, precision);
  hipStream_t st = nsr_stream(stream);
  unsigned host = 0;
  if (hipMemcpyAsync(&host, tail, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) return NSR_ERR_LAUNCH;
  if (clear && hipMemsetAsync(tail, 0, sizeof(unsigned), st) != hipSuccess) return NSR_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return NSR_ERR_LAUNCH;
  *flags_out = host;
  return NSR_OK;
}

extern "C" int nsr_pack_weights(const float* const* w, void* packed_dev, int precision, void* stream) {
  const int rc = nsr_pack_weights_async(w, packed_dev, precision, stream);
  if (rc != NSR_OK) return rc;
  unsigned flags = 0;
  const int rs = nsr_weights_status(packed_dev, precision, 0, &flags, stream);
  if (rs != NSR_OK) return rs;
  return (flags & NSR_FLAG_WEIGHT_RANGE) ? NSR_ERR_RANGE : NSR_OK;
}

extern "C" int nsr_weights_set_options(void* packed_dev, int precision, unsigned options, void* stream) {
  static_assert(kOptGamma == NSR_OPT_GAMMA && kOptColorNone == NSR_OPT_COLOR_NONE, "option bits of include/nsr.h");
  if (!packed_dev || (options & ~(NSR_OPT_GAMMA | NSR_OPT_COLOR_NONE)) != 0u) return NSR_ERR_INVALID_ARG;
  if (!precision_built(precision)) return NSR_ERR_UNSUPPORTED;
  unsigned* tail = nsr_blob_tail(packed_dev, precision);
  if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tail + 1), (int)options, 1, nsr_stream(stream)) != hipSuccess)
    return NSR_ERR_LAUNCH;
  return NSR_OK;
}
extern "C" int nsr_weights_set_gamma(void* packed_dev, int precision, int enable, void* stream) {
  return nsr_weights_set_options(packed_dev, precision, enable ? NSR_OPT_GAMMA : 0u, stream);
}

// ---------------------------------------------------------------------------
// fused MLP kernel, fp32 MFMA
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// LDS byte address (32-bit) of a __shared__ object
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p);
}

// One wave-instruction: 64 lanes x 16 B, global -> LDS (dst = wave-uniform M0 + OFF + lane*16), issued from
// inline asm with a wave-uniform 64-bit base + one 32-bit lane offset.  Why not
// __builtin_amdgcn_global_load_lds: with an LDS-DMA in flight hipcc's waitcnt pass waits lgkmcnt(0) before
// every MFMA group (see nsr_mlp_f16.hip); the asm form is invisible to it and is drained by hand
// (dma_drain) right before the barrier that publishes the chunk.
template <int OFF>
__device__ __forceinline__ void glds16_asm(const char* base_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
  // base through an in-statement SALU copy (a VALU-restored SGPR feeding VMEM needs wait states hipcc cannot add inside
  // inline asm): see nsr_f16x3_core.h
  unsigned long long tmp;
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_mov_b64 %0, %2\n\t"
      "global_load_lds_dwordx4 %1, %0 offset:%4"
      : "=&s"(tmp)
      : "v"(lane_off), "s"(base_uniform), "s"(lds_dst_uniform), "i"(OFF)
      : "memory");
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Stream {
  const float* base;   // packed stream
  int q;               // next chunk to CONSUME
  int q_end;           // chunks in this launch's stream
};

// enqueue chunk `q` into ring slot `ring_slot`: each wave moves a contiguous quarter (8 pieces = 8 KiB)
__device__ __forceinline__ void issue_chunk(const float* stream, int q, float* ring_slot, int wave, int lane) {
  const char* src = reinterpret_cast<const char*>(stream) + (size_t)q * kChunkBytes + wave * 8192;
  const unsigned dst = lds_addr(ring_slot) + (unsigned)wave * 8192u;
  const unsigned lane_off = (unsigned)lane * 16u;
  glds16_asm<0>(src, lane_off, dst);
  glds16_asm<1024>(src, lane_off, dst);
  glds16_asm<2048>(src, lane_off, dst);
  glds16_asm<3072>(src, lane_off, dst);
  glds16_asm<0>(src + 4096, lane_off, dst + 4096u);
  glds16_asm<1024>(src + 4096, lane_off, dst + 4096u);
  glds16_asm<2048>(src + 4096, lane_off, dst + 4096u);
  glds16_asm<3072>(src + 4096, lane_off, dst + 4096u);
}

// Consume one segment of the stream: acc[nb] += W_seg(nb, :) * B, B = b[0..STEPS-1].
// Segments always start on an even chunk, so the ring slot is (local chunk & 1).
template <int NB, int STEPS, int NREG>
__device__ __forceinline__ void run_segment(f32x16 (&acc)[NB], const float (&b)[NREG], Stream& st, float* ring,
                                            int wave, int lane) {
  constexpr int kSlabsPerChunk = kChunkPieces / NB;         // NB=8: 4, NB=4: 8
  constexpr int kSlabs = STEPS / 4;
  constexpr int kChunks = (kSlabs + kSlabsPerChunk - 1) / kSlabsPerChunk;
  static_assert(STEPS <= NREG, "B operand registers");
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    // chunk st.q has been in flight for a whole chunk time: drain + make it visible to all waves;
    // the same barrier proves every wave is done reading the other slot.
    dma_drain();
    __syncthreads();
    if (st.q + 1 < st.q_end) issue_chunk(st.base, st.q + 1, ring + ((c + 1) & 1) * (kChunkBytes / 4), wave, lane);
    const f32x4* buf = reinterpret_cast<const f32x4*>(ring + (c & 1) * (kChunkBytes / 4));
#pragma unroll
    for (int sl = 0; sl < kSlabsPerChunk; ++sl) {
      if (c * kSlabsPerChunk + sl >= kSlabs) break;   // compile-time after unrolling (short last chunk)
      f32x4 wf[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wf[nb] = buf[(sl * NB + nb) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bv = b[(c * kSlabsPerChunk + sl) * 4 + j];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nb][j], bv, acc[nb], 0, 0, 0);
      }
    }
    st.q += 1;
  }
}

// accumulator init = bias, in D-fragment order
template <int NB>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NB], const float* bias, int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * nb + 8 * q + 4 * h);
      acc[nb][4 * q + 0] = b4[0];
      acc[nb][4 * q + 1] = b4[1];
      acc[nb][4 * q + 2] = b4[2];
      acc[nb][4 * q + 3] = b4[3];
    }
}

// dot( regs[0..16*NB-1], w[feature(t, h)] ) for one lane-half; caller adds the halves
template <int NB>
__device__ __forceinline__ float half_dot(const float (&v)[16 * NB], const float* w, int h) {
  float s = 0.0f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + 32 * nb + 8 * q + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) s = fmaf(v[16 * nb + 4 * q + i], w4[i], s);
    }
  return s;
}

// torch.relu keeps NaN (fmaxf would turn it into 0 and hide a diverged trunk from the output check below)
__device__ __forceinline__ float relu_nan(float x) { return (x < 0.0f) ? 0.0f : x; }

// MODE 0: x is (P, 90) embedded rows.  MODE 1: x is rays (R, 8), z (R, N) given.
// NSC > 0 (64 or 128 = samples per ray, MODE 1): the tile's points are whole rays and the kernel composites them itself
// (nsr_composite.h); `out` may then be null.
template <int MODE, bool SIGMA_ONLY, int NSC = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_fp32_kernel(const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ zv,
                int64_t P, int N, int stride, float* __restrict__ out, NsrTail tail, NsrCompOut co = NsrCompOut{}) {
  __shared__ __attribute__((aligned(16))) float ring[2 * kChunkBytes / 4];   // 64 KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* aux = packed + kStreamFloats;

  Stream st;
  st.base = packed;
  st.q = 0;
  st.q_end = SIGMA_ONLY ? 60 : (kStreamPiecesPadded / kChunkPieces);
  issue_chunk(st.base, 0, ring, wave, lane);   // overlap the first 32 KiB with the encoding prologue

  const int64_t p = (int64_t)blockIdx.x * 128 + wave * 32 + m;
  const int64_t pc = p < P ? p : P - 1;

  float pe[32], de[16];
  unsigned flags = 0u;   // NSR_FLAG_* of this lane's point, raised once at the end
  if (MODE == 0) {
    const float* row = x + pc * kInCh;
    bool ok = true;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const int col = pecol(t, h);
      pe[t] = (col == kPad) ? 0.0f : row[col];
      ok &= nsr_finite(pe[t]);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int col = dircol(t, h);
      de[t] = (col == kPad) ? 0.0f : row[kPosCh + col];
      ok &= nsr_finite(de[t]);
    }
    if (!ok) flags |= NSR_FLAG_INPUT_RANGE;
  } else {
    const int64_t ray = pc / N;
    const NsrRay rq = nsr_load_ray(x, ray, stride);
    const float zk = zv[pc];
    const float d[3] = {rq.v[0], rq.v[1], rq.v[2]};           // the direction that is ENCODED
    // cast_rays (models/utils.py:14): o + z*d, separate multiply and add as ATen does
    const float v[3] = {__fadd_rn(rq.o[0], __fmul_rn(zk, rq.d[0])), __fadd_rn(rq.o[1], __fmul_rn(zk, rq.d[1])),
                        __fadd_rn(rq.o[2], __fmul_rn(zk, rq.d[2]))};
    if (!(nsr_finite(v[0]) && nsr_finite(v[1]) && nsr_finite(v[2]) && nsr_finite(d[0]) && nsr_finite(d[1]) && nsr_finite(d[2])))
      flags |= NSR_FLAG_INPUT_RANGE;
    pe[0] = h ? v[2] : v[0];
    pe[1] = h ? 0.0f : v[1];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float sn, cs;
        nsr_sincos(ldexpf(v[c], 5 * h + f), sn, cs);   // 2^k * x is exact; full-range sin/cos
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

  f32x16 acc[8];
  float act[128];
  float sigma = 0.0f;

  // ---- L1: pe(63) -> 256
  init_bias<8>(acc, aux + kAuxBias0, h);
  run_segment<8, 32, 32>(acc, pe, st, ring, wave, lane);
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) act[16 * nb + r] = relu_nan(acc[nb][r]);

  // ---- L2..L8 (+ xyz_encoding_final as "layer 8", no activation)
  constexpr int kLast = SIGMA_ONLY ? 7 : 8;
#pragma unroll 1
  for (int L = 1; L <= kLast; ++L) {
    init_bias<8>(acc, aux + kAuxBias0 + L * 256, h);
    if (L == 4) run_segment<8, 32, 32>(acc, pe, st, ring, wave, lane);   // skip: cat([pe, h])
    run_segment<8, 128, 128>(acc, act, st, ring, wave, lane);
    if (L < 8) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * nb + r] = relu_nan(acc[nb][r]);
    } else {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[16 * nb + r] = acc[nb][r];
    }
    if (L == 7) {   // density head on h8 (raw: the renderer applies the relu)
      float s = half_dot<8>(act, aux + kAuxSigmaW, h);
      s += __shfl_xor(s, 32, 64);
      sigma = s + aux[kAuxSigmaB];
    }
  }

  if (SIGMA_ONLY) {
    if (h == 0 && p < P) out[p] = sigma;
    if (!nsr_finite(sigma)) flags |= NSR_FLAG_OUTPUT_NONFINITE;
    if (p < P) nsr_raise(tail, flags);
    return;
  }

  // ---- dir_encoding: cat([g(256), de(27)]) -> 128, relu
  f32x16 acc4[4];
  init_bias<4>(acc4, aux + kAuxBiasDir, h);
  run_segment<4, 128, 128>(acc4, act, st, ring, wave, lane);
  run_segment<4, 16, 16>(acc4, de, st, ring, wave, lane);
  float cfe[64];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) cfe[16 * nb + r] = relu_nan(acc4[nb][r]);

  // ---- rgb head: 128 -> 3, sigmoid (or none: the network's option word)
  const unsigned opts = nsr_opts(tail);
  float rgb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s = half_dot<4>(cfe, aux + kAuxRgbW + 128 * k, h);
    s += __shfl_xor(s, 32, 64);
    s += aux[kAuxRgbB + k];
    rgb[k] = nsr_colour_activation(s, opts);
  }
  if (opts & kOptGamma) {
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = nsr_gamma(rgb[k]);
  }
  if (!(nsr_finite(rgb[0]) && nsr_finite(rgb[1]) && nsr_finite(rgb[2]) && nsr_finite(sigma))) flags |= NSR_FLAG_OUTPUT_NONFINITE;
  if (p < P) nsr_raise(tail, flags);
  if (out && h == 0 && p < P) reinterpret_cast<float4*>(out)[p] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
  if (NSC > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no weight DMA may still be landing in the ring that is re-used below
    composite_tile<(NSC > 0 ? NSC : 64)>(ring, h == 0, wave, m, lane, make_float4(rgb[0], rgb[1], rgb[2], sigma), zv[pc], P / (NSC > 0 ? NSC : 1), co, (int64_t)blockIdx.x);
  }
}

extern "C" int nsr_mlp_forward(const void* packed_dev, int precision, const float* x, int64_t P, int sigma_only,
                               float* out, void* stream) {
  if (P < 0 || !packed_dev) return NSR_ERR_INVALID_ARG;
  if (!precision_built(precision)) return NSR_ERR_UNSUPPORTED;
  if (P == 0) return NSR_OK;   // empty batch: nothing to read or write (pointers may be null)
  if (!x || !out) return NSR_ERR_INVALID_ARG;
  if (!sigma_only && (reinterpret_cast<uintptr_t>(out) & 15) != 0) return NSR_ERR_INVALID_ARG;
  unsigned* tail = nsr_blob_tail(packed_dev, precision);
  if (precision == NSR_F16X3) return nsr_f16x3_mlp_forward(packed_dev, x, P, sigma_only, out, tail, stream);
  if (precision_h1(precision)) return nsr_h1_mlp_forward(precision == NSR_BF16, packed_dev, x, P, sigma_only, out, tail, stream);
  const dim3 grid((unsigned)((P + 127) / 128)), block(256);
  const float* pk = static_cast<const float*>(packed_dev);
  if (sigma_only)
    hipLaunchKernelGGL((mlp_fp32_kernel<0, true>), grid, block, 0, nsr_stream(stream), pk, x, nullptr, P, 1, 8, out, NsrTail{nsr_blob_tail(packed_dev, precision)});
  else
    hipLaunchKernelGGL((mlp_fp32_kernel<0, false>), grid, block, 0, nsr_stream(stream), pk, x, nullptr, P, 1, 8, out, NsrTail{nsr_blob_tail(packed_dev, precision)});
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_render_rays(const void* packed_dev, int precision, const float* rays, int ray_stride, const float* z,
                               int64_t R, int n_samples, float* out, void* stream) {
  if (!packed_dev || R < 0 || n_samples <= 0 || !nsr_ray_stride_ok(ray_stride)) return NSR_ERR_INVALID_ARG;
  if (!precision_built(precision)) return NSR_ERR_UNSUPPORTED;
  if (R == 0) return NSR_OK;
  if (!rays || !z || !out) return NSR_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(out) & 15) != 0 || (ray_stride == 8 && (reinterpret_cast<uintptr_t>(rays) & 15) != 0))
    return NSR_ERR_INVALID_ARG;
  unsigned* tail = nsr_blob_tail(packed_dev, precision);
  if (precision == NSR_F16X3) return nsr_f16x3_render_rays(packed_dev, rays, ray_stride, z, R, n_samples, out, tail, stream);
  if (precision_h1(precision))
    return nsr_h1_render_rays(precision == NSR_BF16, packed_dev, rays, ray_stride, z, R, n_samples, out, tail, stream);
  const int64_t P = R * n_samples;
  const dim3 grid((unsigned)((P + 127) / 128)), block(256);
  hipLaunchKernelGGL((mlp_fp32_kernel<1, false>), grid, block, 0, nsr_stream(stream),
                     static_cast<const float*>(packed_dev), rays, z, P, n_samples, ray_stride, out, NsrTail{tail});
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

/* D2 + V1 in one launch, see include/nsr.h */
extern "C" NSR_INTERNAL int nsr_f16x3_render_composite(const void* packed, const float* rays, int ray_stride, const float* z,
                                                       int64_t R, int N, float* raw, const NsrCompOut* co, unsigned* tail, void* stream);

extern "C" int nsr_render_rays_composited(const void* packed_dev, int precision, const float* rays, int ray_stride, const float* z,
                                          int64_t R, int n_samples, int white_bkgd, float* raw, float* comp_rgb, float* depth,
                                          float* opacity, float* weights, void* stream) {
  if (!packed_dev || R < 0 || n_samples <= 0 || !nsr_ray_stride_ok(ray_stride) || (white_bkgd & ~(NSR_WHITE_BKGD | NSR_SIGMA_SOFTPLUS)) != 0)
    return NSR_ERR_INVALID_ARG;
  if (!precision_built(precision)) return NSR_ERR_UNSUPPORTED;
  if ((precision != NSR_FP32 && precision != NSR_F16X3) || (n_samples != 64 && n_samples != 128)) return NSR_ERR_UNSUPPORTED;
  if (R == 0) return NSR_OK;
  if (!rays || !z) return NSR_ERR_INVALID_ARG;
  if ((raw && (reinterpret_cast<uintptr_t>(raw) & 15) != 0) || (ray_stride == 8 && (reinterpret_cast<uintptr_t>(rays) & 15) != 0))
    return NSR_ERR_INVALID_ARG;
  const NsrCompOut co{comp_rgb, depth, opacity, weights, white_bkgd};
  unsigned* tail = nsr_blob_tail(packed_dev, precision);
  if (precision == NSR_F16X3) return nsr_f16x3_render_composite(packed_dev, rays, ray_stride, z, R, n_samples, raw, &co, tail, stream);
  const int64_t P = R * n_samples;
  const dim3 grid((unsigned)((P + 127) / 128)), block(256);
  const float* pk = static_cast<const float*>(packed_dev);
  if (n_samples == 64)
    hipLaunchKernelGGL((mlp_fp32_kernel<1, false, 64>), grid, block, 0, nsr_stream(stream), pk, rays, z, P, n_samples, ray_stride, raw, NsrTail{tail}, co);
  else
    hipLaunchKernelGGL((mlp_fp32_kernel<1, false, 128>), grid, block, 0, nsr_stream(stream), pk, rays, z, P, n_samples, ray_stride, raw, NsrTail{tail}, co);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
