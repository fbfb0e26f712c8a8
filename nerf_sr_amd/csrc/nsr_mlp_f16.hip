// M1 / D2 on the fp16 matrix pipe with split operands ("f16x3", NSR_F16X3):
// every fp32 value v is carried as hi = f16(v), lo = f16(v - hi) and each product is
// formed as a_hi*b_hi + (a_hi*b_lo + a_lo*b_hi) on v_mfma_f32_32x32x16_f16 with fp32
// accumulation — products exact to ~2^-21, i.e. fp32-grade results (measured vs the
// fp32 oracle: <= 3e-5 RGB) at 3/16 of the fp32-MFMA cycle cost.
//
// Same register algebra as the fp32 kernel (nsr_mlp_layout.h): a wave owns 32 sample
// points, activations never leave registers, weights stream global -> LDS by DMA.
// Differences: the stream is consumed one 32-feature OUTPUT block at a time (chunk =
// all k-steps of that block + its bias), so that a finished block is re-split into the
// next layer's hi/lo operand registers on the VALU while the next block occupies the
// matrix pipe; two register sets alternate roles layer by layer.
#include "nsr_common.h"
#include "nsr_mlp_layout.h"

using namespace nsr;
using namespace nsr::hx;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------
struct PackPtrsH {
  const float* p[NSR_N_STATE_TENSORS];
};

__device__ __forceinline__ int tensor_ld_h(int tensor) {
  switch (tensor) {
    case 0: return kPosCh;
    case 8: return kWidth + kPosCh;
    case 18: return kWidth + kDirCh;
    default: return kWidth;
  }
}

__device__ __forceinline__ unsigned pack_hl(float a, float b, int part) {
  _Float16 ha = (_Float16)a, hb = (_Float16)b;               // round to nearest even
  if (part) {
    ha = (_Float16)(a - (float)ha);
    hb = (_Float16)(b - (float)hb);
  }
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}

// one thread per 32-bit word of the blob
__global__ void __launch_bounds__(256) pack_f16x3_kernel(PackPtrsH w, unsigned* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int stream_words = kPiecesTotal * 256;
  if (idx >= stream_words + hx::kAuxFloats) return;
  unsigned v = 0u;
  if (idx < stream_words) {
    const int piece = idx >> 8, word = idx & 255;
    int q = 0;
    for (int i = 1; i < kChunks; ++i)
      if (piece >= chunk_info(i).piece0) q = i;
    const Chunk c = chunk_info(q);
    const int local = piece - c.piece0;
    const int npieces = chunk_pieces(c.steps, c.nnb);
    if (local == npieces - 1) {
      // bias piece: fp32 bias of the chunk's output blocks, 32 per block
      if (word < 32 * c.nnb) v = __float_as_uint(w.p[c.tensor + 1][32 * c.nb0 + word]);
    } else {
      const int g = local / (2 * c.steps), rem = local % (2 * c.steps);
      const int s = rem >> 1, part = rem & 1;
      const int lane = word >> 2, jj = word & 3;
      const int n = 32 * (c.nb0 + g) + (lane & 31), h = lane >> 5;
      const int ld = tensor_ld_h(c.tensor);
      float f[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = column_of(c.tensor, s, 2 * jj + e, h);
        f[e] = (col == kPad) ? 0.0f : w.p[c.tensor][n * ld + col];
      }
      v = pack_hl(f[0], f[1], part);
    }
  } else {
    const int a = idx - stream_words;
    float f = 0.0f;
    if (a < hx::kAuxRgbW) f = w.p[20][a];
    else if (a < hx::kAuxSigmaB) f = w.p[22][a - hx::kAuxRgbW];
    else if (a < hx::kAuxRgbB) f = w.p[21][0];
    else if (a < hx::kAuxRgbB + 3) f = w.p[23][a - hx::kAuxRgbB];
    v = __float_as_uint(f);
  }
  out[idx] = v;
}

extern "C" size_t nsr_f16x3_packed_bytes(void) { return 4 * (size_t)(kPiecesTotal * 256 + hx::kAuxFloats); }

extern "C" int nsr_f16x3_pack(const float* const* w, void* packed_dev, void* stream) {
  PackPtrsH pp;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w[i]) return NSR_ERR_INVALID_ARG;
    pp.p[i] = w[i];
  }
  const int total = kPiecesTotal * 256 + hx::kAuxFloats;
  hipLaunchKernelGGL(pack_f16x3_kernel, dim3((total + 255) / 256), dim3(256), 0, nsr_stream(stream), pp,
                     static_cast<unsigned*>(packed_dev));
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// ---------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const float* gsrc_lane, float* lds_base_wave) {
  __builtin_amdgcn_global_load_lds((glb_ptr_t)gsrc_lane, (lds_ptr_t)lds_base_wave, 16, 0, 0);
}

struct Loader {
  const float* stream;   // packed blob viewed as 32-bit words
  int q;                 // chunk being CONSUMED
  int q_end;
  int wave, lane;
  // descriptor of chunk q+1 (what the current chunk's body is prefetching)
  const float* next_src;
  int next_pieces;
};

__device__ __forceinline__ void loader_prepare_next(Loader& ld) {
  const int qn = ld.q + 1;
  if (qn < ld.q_end) {
    const Chunk c = chunk_info(qn);
    ld.next_pieces = chunk_pieces(c.steps, c.nnb);
    ld.next_src = ld.stream + (size_t)c.piece0 * 256 + ld.lane * 4;
  } else {
    ld.next_pieces = 0;
    ld.next_src = ld.stream;
  }
}

// issue DMA piece number 4*i + wave of the next chunk into `slot` (no-op past its end)
__device__ __forceinline__ void loader_issue(const Loader& ld, float* slot, int i) {
  const int p = 4 * i + ld.wave;
  if (p < ld.next_pieces) glds16(ld.next_src + p * 256, slot + p * 256);
}

__device__ __forceinline__ h8 as_h8(const u32x4& v) { return __builtin_bit_cast(h8, v); }

// NSTEP k-steps of one output block: am += A_hi*B_hi, ac += A_hi*B_lo + A_lo*B_hi.
// Prefetches two DMA pieces of the next chunk per k-step for steps < 6 (ISSUE).
template <int NSTEP, int B0, bool ISSUE, int NB>
__device__ __forceinline__ void mma_steps(f32x16& am, f32x16& ac, const u32x4 (&bh)[NB], const u32x4 (&bl)[NB],
                                          const u32x4* a_pieces, const Loader& ld, float* next_slot, int issue0) {
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const u32x4 ah = a_pieces[(2 * s) * 64];
    const u32x4 al = a_pieces[(2 * s + 1) * 64];
    am = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah), as_h8(bh[B0 + s]), am, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah), as_h8(bl[B0 + s]), ac, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al), as_h8(bh[B0 + s]), ac, 0, 0, 0);
    if (ISSUE && issue0 + s < 6) {
      loader_issue(ld, next_slot, 2 * (issue0 + s));
      loader_issue(ld, next_slot, 2 * (issue0 + s) + 1);
    }
  }
}

// accumulator init from the chunk's bias piece (fp32, D-fragment order)
__device__ __forceinline__ void init_bias(f32x16& am, const float* bias32, int h) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias32 + 8 * qd + 4 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) am[4 * qd + i] = b4[i];
  }
}

// v -> (hi, lo) fp16 pairs packed two per 32-bit register (round toward zero for hi; lo takes the rest)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const auto ph = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  const auto pl = __builtin_amdgcn_cvt_pkrtz(x0 - (float)ph[0], x1 - (float)ph[1]);
  hi = __builtin_bit_cast(unsigned, ph);
  lo = __builtin_bit_cast(unsigned, pl);
}

// finished block (am + ac), optional relu -> k-steps 2nb, 2nb+1 of the next layer's operands
__device__ __forceinline__ void split_block(const f32x16& v, u32x4& h0, u32x4& l0, u32x4& h1, u32x4& l1) {
  unsigned a[4], b[4], c[4], d[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    split2(v[2 * p], v[2 * p + 1], a[p], b[p]);
    split2(v[8 + 2 * p], v[8 + 2 * p + 1], c[p], d[p]);
  }
  h0 = u32x4{a[0], a[1], a[2], a[3]};
  l0 = u32x4{b[0], b[1], b[2], b[3]};
  h1 = u32x4{c[0], c[1], c[2], c[3]};
  l1 = u32x4{d[0], d[1], d[2], d[3]};
}

// sum_r v[r] * w[feature(r, h)] over one block (features 8q + 4h + i)
__device__ __forceinline__ float block_dot(const f32x16& v, const float* w32, int h, float s) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(w32 + 8 * qd + 4 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) s = fmaf(v[4 * qd + i], w4[i], s);
  }
  return s;
}

// One 256 -> 256 trunk layer L (1..8; 8 = xyz_encoding_final): in (bh, bl) -> out (oh, ol).
// L == 4 prepends the 4 positional-encoding k-steps (skip connection); L == 7 also feeds the
// density head from the fp32 block results.
__device__ __forceinline__ void trunk_layer(int L, const u32x4 (&bh)[16], const u32x4 (&bl)[16], u32x4 (&oh)[16],
                                            u32x4 (&ol)[16], const u32x4 (&peh)[4], const u32x4 (&pel)[4],
                                            Loader& ld, float* ring, const float* aux, int h, float& sigma_acc) {
  const float lower = (L < 8) ? 0.0f : -__builtin_inff();   // relu on L1..L8, none on xyz_encoding_final
  const int skip = (L == 4) ? 8 : 0;                        // pieces taken by the pe k-steps
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    float* slot = ring + (nb & 1) * kSlotFloats;
    float* next_slot = ring + ((nb + 1) & 1) * kSlotFloats;
    __syncthreads();                       // chunk ld.q landed (vmcnt(0)) and the other slot is free
    loader_prepare_next(ld);
    const u32x4* a0 = reinterpret_cast<const u32x4*>(slot) + ld.lane;
    f32x16 am, ac = {};
    init_bias(am, slot + (32 + skip) * 256, h);   // bias piece follows the 2*steps weight pieces
    if (L == 4) mma_steps<4, 0, false, 4>(am, ac, peh, pel, a0, ld, next_slot, 0);
    mma_steps<16, 0, true, 16>(am, ac, bh, bl, a0 + skip * 64, ld, next_slot, 0);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaxf(am[r] + ac[r], lower);
    if (L == 7) sigma_acc = block_dot(v, aux + hx::kAuxSigmaW + 32 * nb, h, sigma_acc);
    split_block(v, oh[2 * nb], ol[2 * nb], oh[2 * nb + 1], ol[2 * nb + 1]);
    ld.q += 1;
  }
}

template <int MODE, bool SIGMA_ONLY, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_f16x3_kernel(const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ zv,
                 int64_t P, int N, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float ring[2 * kSlotFloats];   // 2 x 41 KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* aux = packed + kPiecesTotal * 256;

  Loader ld;
  ld.stream = packed;
  ld.q = -1;
  ld.q_end = SIGMA_ONLY ? kChunksSigmaOnly : kChunks;
  ld.wave = wave;
  ld.lane = lane;
  loader_prepare_next(ld);                 // chunk 0
#pragma unroll
  for (int i = 0; i < 11; ++i) loader_issue(ld, ring, i);
  ld.q = 0;

  const int64_t p = (int64_t)blockIdx.x * 128 + wave * 32 + m;
  const int64_t pc = p < P ? p : P - 1;

  float pe[32], de[16];
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
    const int64_t ray = (NS > 0) ? pc / NS : pc / N;
    const float4 ra = reinterpret_cast<const float4*>(x + ray * 8)[0];
    const float4 rb = reinterpret_cast<const float4*>(x + ray * 8)[1];
    const float zk = zv[pc];
    const float d[3] = {ra.w, rb.x, rb.y};
    const float v[3] = {__fadd_rn(ra.x, __fmul_rn(zk, d[0])), __fadd_rn(ra.y, __fmul_rn(zk, d[1])),
                        __fadd_rn(ra.z, __fmul_rn(zk, d[2]))};
    pe[0] = h ? v[2] : v[0];
    pe[1] = h ? 0.0f : v[1];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float sn, cs;
        sincosf(ldexpf(v[c], 5 * h + f), &sn, &cs);
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
        sincosf(ldexpf(d[c], 2 * h + f), &sn, &cs);
        de[2 + 6 * f + c] = sn;
        de[2 + 6 * f + 3 + c] = cs;
      }
    de[14] = 0.0f;
    de[15] = 0.0f;
  }
  u32x4 peh[4], pel[4], deh[2], del[2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    unsigned a[4], b[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) split2(pe[8 * s + 2 * pr], pe[8 * s + 2 * pr + 1], a[pr], b[pr]);
    peh[s] = u32x4{a[0], a[1], a[2], a[3]};
    pel[s] = u32x4{b[0], b[1], b[2], b[3]};
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    unsigned a[4], b[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) split2(de[8 * s + 2 * pr], de[8 * s + 2 * pr + 1], a[pr], b[pr]);
    deh[s] = u32x4{a[0], a[1], a[2], a[3]};
    del[s] = u32x4{b[0], b[1], b[2], b[3]};
  }

  u32x4 bh[16], bl[16], oh[16], ol[16];

  // ---- L1: two chunks of four output blocks, 4 k-steps each
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float* slot = ring + (c & 1) * kSlotFloats;
    float* next_slot = ring + ((c + 1) & 1) * kSlotFloats;
    __syncthreads();
    loader_prepare_next(ld);
    const u32x4* a0 = reinterpret_cast<const u32x4*>(slot) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = 4 * c + g;
      f32x16 am, ac = {};
      init_bias(am, slot + 32 * 256 + 32 * g, h);
      mma_steps<4, 0, true, 4>(am, ac, peh, pel, a0 + g * 8 * 64, ld, next_slot, (g == 0) ? 0 : (g == 1 ? 4 : 6));
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = fmaxf(am[r] + ac[r], 0.0f);
      split_block(v, bh[2 * nb], bl[2 * nb], bh[2 * nb + 1], bl[2 * nb + 1]);
    }
    ld.q += 1;
  }

  // ---- L2..L8 (+ xyz_encoding_final), two layers per trip so the register sets swap roles
  float sigma = 0.0f;
  constexpr int kPairs = SIGMA_ONLY ? 3 : 4;
#pragma unroll 1
  for (int pair = 0; pair < kPairs; ++pair) {
    const int L = 1 + 2 * pair;
    trunk_layer(L, bh, bl, oh, ol, peh, pel, ld, ring, aux, h, sigma);
    trunk_layer(L + 1, oh, ol, bh, bl, peh, pel, ld, ring, aux, h, sigma);
  }
  if (SIGMA_ONLY) trunk_layer(7, bh, bl, oh, ol, peh, pel, ld, ring, aux, h, sigma);
  sigma += __shfl_xor(sigma, 32, 64);
  sigma += aux[hx::kAuxSigmaB];
  if (SIGMA_ONLY) {
    if (h == 0 && p < P) out[p] = sigma;
    return;
  }

  // ---- dir_encoding (cat([g, de]) -> 128, relu) fused with the rgb head (128 -> 3, sigmoid)
  float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    float* slot = ring + (nb & 1) * kSlotFloats;
    float* next_slot = ring + ((nb + 1) & 1) * kSlotFloats;
    __syncthreads();
    loader_prepare_next(ld);
    const u32x4* a0 = reinterpret_cast<const u32x4*>(slot) + lane;
    f32x16 am, ac = {};
    init_bias(am, slot + 36 * 256, h);
    mma_steps<16, 0, true, 16>(am, ac, bh, bl, a0, ld, next_slot, 0);
    mma_steps<2, 0, false, 2>(am, ac, deh, del, a0 + 32 * 64, ld, next_slot, 0);
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaxf(am[r] + ac[r], 0.0f);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = block_dot(v, aux + hx::kAuxRgbW + 128 * k + 32 * nb, h, rgb[k]);
    ld.q += 1;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s = rgb[k];
    s += __shfl_xor(s, 32, 64);
    s += aux[hx::kAuxRgbB + k];
    rgb[k] = 1.0f / (1.0f + expf(-s));
  }
  if (h == 0 && p < P) reinterpret_cast<float4*>(out)[p] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
}

template <int MODE, bool SIGMA_ONLY>
static int launch_f16x3(const void* packed, const float* x, const float* z, int64_t P, int N, float* out,
                        hipStream_t st) {
  const dim3 grid((unsigned)((P + 127) / 128)), block(256);
  const float* pk = static_cast<const float*>(packed);
  if (MODE == 1 && N == 64)
    hipLaunchKernelGGL((mlp_f16x3_kernel<MODE, SIGMA_ONLY, 64>), grid, block, 0, st, pk, x, z, P, N, out);
  else if (MODE == 1 && N == 128)
    hipLaunchKernelGGL((mlp_f16x3_kernel<MODE, SIGMA_ONLY, 128>), grid, block, 0, st, pk, x, z, P, N, out);
  else
    hipLaunchKernelGGL((mlp_f16x3_kernel<MODE, SIGMA_ONLY, 0>), grid, block, 0, st, pk, x, z, P, N, out);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

extern "C" int nsr_f16x3_mlp_forward(const void* packed, const float* x, int64_t P, int sigma_only, float* out,
                                     void* stream) {
  return sigma_only ? launch_f16x3<0, true>(packed, x, nullptr, P, 1, out, nsr_stream(stream))
                    : launch_f16x3<0, false>(packed, x, nullptr, P, 1, out, nsr_stream(stream));
}

extern "C" int nsr_f16x3_render_rays(const void* packed, const float* rays, const float* z, int64_t R, int N,
                                     float* out, void* stream) {
  return launch_f16x3<1, false>(packed, rays, z, R * N, N, out, nsr_stream(stream));
}
