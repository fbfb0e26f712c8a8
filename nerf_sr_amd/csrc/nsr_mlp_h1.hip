// M1 / D2 with plain 16-bit operands — the FAST path: NSR_BF16 (v_mfma_f32_32x32x16_bf16) and NSR_F16
// (v_mfma_f32_32x32x16_f16), one MFMA per product, fp32 accumulation, everything outside the contraction
// (encodings, bias, activations, heads, compositing) in fp32.  NOT a parity path: operand rounding (2^-9 for
// bf16, 2^-12 for f16) moves rendered colours by ~5e-3 / ~6e-4 (PSNR vs the reference ~45-60 dB); the
// contract-grade paths are NSR_F16X3 and NSR_FP32.
//
// Same design as nsr_mlp_f16.hip (a wave owns 32 sample points, activations stay in registers as the MFMA B
// operand of the next layer, weights stream global -> LDS by DMA through a 3-slot ring), with ONE difference:
// a chunk carries TWO 32-feature output blocks per k-step instead of the (hi, lo) halves of one, so each k-step
// is two independent MFMAs sharing the B operand and the stream keeps the same 2-pieces-per-k-step shape.
#include "nsr_common.h"
#include "nsr_mlp_layout.h"
#include "nsr_mlp_stream.h"
#include "nsr_mlp_encode.h"

using namespace nsr;
using namespace nsr::stream;

namespace h1 {

// ---- stream layout ----------------------------------------------------------------------------------------
// chunk = [ A(nb0, s=0), A(nb1, s=0), A(nb0, 1), A(nb1, 1), ..., bias piece ]
// The bias piece is itself an A fragment: lane (row i, half 0) carries (hi, lo) of bias[nb0][i] in k-slots 0, 1
// and of bias[nb1][i] in k-slots 2, 3, zero elsewhere; one MFMA against a constant B (ones in k-slots 0, 1 resp.
// 2, 3) initialises the accumulator with the bias, exact to 2^-22 (fp16) / 2^-17 (bf16) -- ONE LDS read per
// chunk instead of eight broadcast reads (this kernel is LDS-bandwidth-bound: every MFMA needs a 1 KiB
// fragment and the LDS moves 128 B/clk).
//   0        : L1, four block pairs of 4 k-steps each + one bias piece per pair
//   1 .. 32  : trunk layer L = 1..8 (L2..L8, xyz_encoding_final), block pair pb = 0..3 ; L == 4: 4 + 16 k-steps
//   33       : density head paired with an all-zero block
//   34, 35   : dir_encoding block pairs, 16 + 2 k-steps
constexpr int kChunks = 36;
constexpr int kSlotPieces = 41;
constexpr int kSlotFloats = kSlotPieces * 256;
constexpr int kSlotBytes = kSlotFloats * 4;
constexpr int kL1Pieces = 36;
constexpr int kSigmaPiece0 = 1124, kDirPiece0 = 1157;
constexpr int kPiecesTotal = 1231;
constexpr int kAuxRgbW = 0, kAuxRgbB = 384, kAuxFloats = 448;
constexpr int kBar = 8;
constexpr int kPF = 3;
// waves per workgroup: 8 = two per SIMD (the kernel fits 256 registers), so one wave's LDS reads, conversions
// and DMA issue run in the shadow of its partner's MFMAs
#ifndef NSR_H1_WAVES
#define NSR_H1_WAVES 8
#endif
constexpr int kWaves = NSR_H1_WAVES;
constexpr int kTile = 32 * kWaves;                           // sample points per workgroup
constexpr int kMinIssue = 32 / kWaves;                       // pieces every wave owns of every chunk
constexpr int kMaxIssue = (kSlotPieces + kWaves - 1) / kWaves;   // ... and at most
constexpr int kIssuePerStep = (kMaxIssue + 5) / 6;           // DMA issue slots: k-steps kBar .. kBar + 5

struct Chunk {
  int tensor, nb0, steps, piece0, pieces;
  bool l1, sigma;
};
NSR_HD int trunk_pieces(int L) { return (L == 4) ? 41 : 33; }
NSR_HD int trunk_base(int L) { return (L <= 3) ? 36 + 132 * (L - 1) : (L == 4 ? 432 : 596 + 132 * (L - 5)); }
NSR_HD Chunk chunk_info(int q) {
  Chunk c{};
  if (q == 0) {
    c.tensor = 0; c.nb0 = 0; c.steps = 4; c.piece0 = 0; c.pieces = kL1Pieces; c.l1 = true;
  } else if (q <= 32) {
    const int L = 1 + (q - 1) / 4, pb = (q - 1) % 4;
    c.tensor = 2 * L; c.nb0 = 2 * pb; c.steps = (L == 4) ? 20 : 16;
    c.pieces = trunk_pieces(L); c.piece0 = trunk_base(L) + c.pieces * pb;
  } else if (q == 33) {
    c.tensor = 20; c.nb0 = 0; c.steps = 16; c.piece0 = kSigmaPiece0; c.pieces = 33; c.sigma = true;
  } else {
    c.tensor = 18; c.nb0 = 2 * (q - 34); c.steps = 18; c.piece0 = kDirPiece0 + 37 * (q - 34); c.pieces = 37;
  }
  return c;
}

__device__ __forceinline__ ChunkRef mkref(int piece0, int pieces, int wave) { return make_ref<kWaves>(piece0, pieces, wave); }
__device__ __forceinline__ void issue(const Loader& ld, int i) { loader_issue<kMinIssue>(ld, i); }
__device__ __forceinline__ ChunkRef layer_ref(int L, int pb, int wave) {
  return mkref(trunk_base(L) + trunk_pieces(L) * pb, trunk_pieces(L), wave);
}
__device__ __forceinline__ ChunkRef sigma_ref(int wave) { return mkref(kSigmaPiece0, 33, wave); }
__device__ __forceinline__ ChunkRef dir_ref(int pb, int wave) { return mkref(kDirPiece0 + 37 * pb, 37, wave); }
// past the end: chunk 0 again into the idle slot (first eight DMA issues stay branch-free; drained at exit)
__device__ __forceinline__ ChunkRef end_ref(int wave) { return mkref(0, 32, wave); }

constexpr unsigned kRelu = 0u, kNoAct = 0x80008000u;

// ---- element conversion ---------------------------------------------------------------------------------
template <bool BF>
__device__ __forceinline__ unsigned pack2(float a, float b) {   // round to nearest even, a in the low half
  if (BF) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, h2{(_Float16)a, (_Float16)b});
  }
}

// ---- packing -------------------------------------------------------------------------------------------
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
template <bool BF>
__global__ void __launch_bounds__(256) pack_kernel(PackPtrs w, unsigned* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int stream_words = kPiecesTotal * 256;
  if (idx >= stream_words + kAuxFloats) return;
  unsigned v = 0u;
  if (idx < stream_words) {
    const int piece = idx >> 8, word = idx & 255;
    int q = 0;
    for (int i = 1; i < kChunks; ++i)
      if (piece >= chunk_info(i).piece0) q = i;
    const Chunk c = chunk_info(q);
    const int local = piece - c.piece0;
    const int n_bias = c.l1 ? 4 : 1;                      // bias pieces at the end of the chunk
    if (local >= c.pieces - n_bias) {
      // bias fragment: lane (i, half 0), word jj < 2 = (hi, lo) of the bias of row i of block 2 * pair + jj
      const int pair = local - (c.pieces - n_bias);
      const int lane = word >> 2, jj = word & 3;
      if (lane < 32 && jj < 2) {
        float bv = 0.0f;
        if (c.sigma) bv = (lane == 0 && jj == 0) ? w.p[21][0] : 0.0f;      // one real row, partner block none
        else bv = w.p[c.tensor + 1][32 * (c.nb0 + 2 * pair + jj) + lane];
        const unsigned hi2 = pack2<BF>(bv, 0.0f);
        const float hi = BF ? __uint_as_float(hi2 << 16) : (float)__builtin_bit_cast(_Float16, (unsigned short)(hi2 & 0xffffu));
        v = pack2<BF>(hi, bv - hi);
      }
    } else {
      int g, s;
      if (c.l1) { g = 2 * (local >> 3) + (local & 1); s = (local & 7) >> 1; }   // four pairs x 4 k-steps x 2 blocks
      else { g = local & 1; s = local >> 1; }
      const int lane = word >> 2, jj = word & 3;
      const int n = 32 * (c.nb0 + g) + (lane & 31), h = lane >> 5;
      float f[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = hx::column_of(c.tensor, s, 2 * jj + e, h);
        const bool real_row = !c.sigma || n == 0;          // density head: row 0 of block 0 only
        f[e] = (col == kPad || !real_row) ? 0.0f : w.p[c.tensor][n * tensor_ld(c.tensor) + col];
      }
      v = pack2<BF>(f[0], f[1]);
    }
  } else {
    const int a = idx - stream_words;
    float f = 0.0f;
    if (a < kAuxRgbB) f = w.p[22][a];
    else if (a < kAuxRgbB + 3) f = w.p[23][a - kAuxRgbB];
    v = __float_as_uint(f);
  }
  out[idx] = v;
}

// ---- kernel ----------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <bool BF>
__device__ __forceinline__ f32x16 mma(const u32x4& a, const u32x4& b, const f32x16& c) {
  if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

struct Acc2 {
  f32x16 a0, a1;   // the chunk's two output blocks
};
struct Pre2 {      // head of a k-step sequence: fragments of its first kPF k-steps (+ the chunk's bias fragment)
  u32x4 f0[kPF], f1[kPF];
  u32x4 bias;
};
__device__ __forceinline__ void prefetch_frag(Pre2& pre, int k, unsigned seq_addr) {
  const u32x4* a = lds_vec(seq_addr);
  pre.f0[k] = a[(2 * k) * 64];
  pre.f1[k] = a[(2 * k + 1) * 64];
}
// constant B operands of the bias MFMA: ones in k-slots (0, 1) resp. (2, 3) of the lanes of half 0
template <bool BF>
__device__ __forceinline__ void bias_operands(int h, u32x4& b0, u32x4& b1) {
  const unsigned ones = h == 0 ? (BF ? 0x3f803f80u : 0x3c003c00u) : 0u;
  b0 = u32x4{ones, 0u, 0u, 0u};
  b1 = u32x4{0u, ones, 0u, 0u};
}
template <bool BF>
__device__ __forceinline__ void init_acc(Acc2& acc, const u32x4& bias_frag, const u32x4& b0, const u32x4& b1) {
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  acc.a1 = mma<BF>(bias_frag, b1, zero);
  acc.a0 = mma<BF>(bias_frag, b0, zero);
}

// NSTEP k-steps of a block pair; same pipeline discipline as block_mma in nsr_mlp_f16.hip
template <bool BF, int NSTEP, int BAR, class BOf, class Hook, class Next>
__device__ __forceinline__ void pair_mma(Acc2& acc, const Pre2& pre, unsigned a_addr, Loader& ld, const ChunkRef& c2,
                                         BOf&& b_of, Hook&& hook, Next&& next) {
  static_assert(NSTEP >= kPF, "sequence shorter than the prefetch depth");
  const u32x4* a_pieces = lds_vec(a_addr);
  u32x4 f0[NSTEP], f1[NSTEP];
#pragma unroll
  for (int s = 0; s < kPF; ++s) {
    f0[s] = pre.f0[s];
    f1[s] = pre.f1[s];
  }
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    if (s == BAR) loader_publish(ld, c2);
    if (s + kPF < NSTEP) {
      f0[s + kPF] = a_pieces[(2 * (s + kPF)) * 64];
      f1[s + kPF] = a_pieces[(2 * (s + kPF) + 1) * 64];
    } else {
      next(s + kPF - NSTEP);
    }
    const u32x4 b = b_of(s);
    acc.a1 = mma<BF>(f1[s], b, acc.a1);     // the younger load first: one lgkmcnt wait serves both
    acc.a0 = mma<BF>(f0[s], b, acc.a0);
    hook(s);
    if (BAR >= 0 && s >= BAR && s < BAR + 6) {
#pragma unroll
      for (int i = 0; i < kIssuePerStep; ++i) issue(ld, kIssuePerStep * (s - BAR) + i);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// pair Q (0..15) of a finished block pair: accumulator registers 2P, 2P+1 (P = Q & 7) of block Q >> 3 ->
// activation -> 16-bit pair -> k-step register (2 per block) of the consuming layer
template <bool BF, int Q>
__device__ __forceinline__ void conv_pair(const Acc2& p, unsigned lower, u32x4& o0, u32x4& o1, u32x4& o2, u32x4& o3) {
  constexpr int P = Q & 7;
  // round first, relu second: on the packed 16-bit pair relu is ONE integer max with kRelu = 0 (negative
  // floats, -0 included, are negative int16 in both formats), and rounding is monotone with round(0) = 0, so
  // the result equals round(relu(x)); kNoAct = int16 minimum makes the same instruction a no-op.  asm volatile
  // pins the work into the MFMA shadow of this k-step.
  unsigned r = (Q < 8) ? pack2<BF>(p.a0[2 * P], p.a0[2 * P + 1]) : pack2<BF>(p.a1[2 * P], p.a1[2 * P + 1]);
  asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(r), "v"(lower));
  if (Q < 4) o0[P & 3] = r;
  else if (Q < 8) o1[P & 3] = r;
  else if (Q < 12) o2[P & 3] = r;
  else o3[P & 3] = r;
}
template <bool BF>
__device__ __forceinline__ void conv_q(int q, const Acc2& p, unsigned lower, u32x4& o0, u32x4& o1, u32x4& o2, u32x4& o3) {
  switch (q) {
    case 0: conv_pair<BF, 0>(p, lower, o0, o1, o2, o3); break;
    case 1: conv_pair<BF, 1>(p, lower, o0, o1, o2, o3); break;
    case 2: conv_pair<BF, 2>(p, lower, o0, o1, o2, o3); break;
    case 3: conv_pair<BF, 3>(p, lower, o0, o1, o2, o3); break;
    case 4: conv_pair<BF, 4>(p, lower, o0, o1, o2, o3); break;
    case 5: conv_pair<BF, 5>(p, lower, o0, o1, o2, o3); break;
    case 6: conv_pair<BF, 6>(p, lower, o0, o1, o2, o3); break;
    case 7: conv_pair<BF, 7>(p, lower, o0, o1, o2, o3); break;
    case 8: conv_pair<BF, 8>(p, lower, o0, o1, o2, o3); break;
    case 9: conv_pair<BF, 9>(p, lower, o0, o1, o2, o3); break;
    case 10: conv_pair<BF, 10>(p, lower, o0, o1, o2, o3); break;
    case 11: conv_pair<BF, 11>(p, lower, o0, o1, o2, o3); break;
    case 12: conv_pair<BF, 12>(p, lower, o0, o1, o2, o3); break;
    case 13: conv_pair<BF, 13>(p, lower, o0, o1, o2, o3); break;
    case 14: conv_pair<BF, 14>(p, lower, o0, o1, o2, o3); break;
    case 15: conv_pair<BF, 15>(p, lower, o0, o1, o2, o3); break;
    default: break;
  }
}
// 16 pairs over the k-steps 0..13 of a 16-step sequence (k-steps 0 and 1 take two), so that even the operands
// of k-steps 12..15 (blocks 6, 7 of the previous layer) are complete before they are read
template <bool BF>
__device__ __forceinline__ void pending_step(int s, const Acc2& p, unsigned lower, u32x4& o0, u32x4& o1, u32x4& o2,
                                             u32x4& o3) {
  if (s < 2) { conv_q<BF>(2 * s, p, lower, o0, o1, o2, o3); conv_q<BF>(2 * s + 1, p, lower, o0, o1, o2, o3); }
  else if (s < 14) conv_q<BF>(s + 2, p, lower, o0, o1, o2, o3);
}
// colour head: pair Q of a finished dir_encoding block pair (relu) dotted with the three rgb rows
template <int Q>
__device__ __forceinline__ void rgb_pair(const Acc2& p, const float* w64, int h, float (&rgb)[3]) {
  constexpr int P = Q & 7, r = 2 * P;
  const float x0 = fmaxf(Q < 8 ? p.a0[r] : p.a1[r], 0.0f);
  const float x1 = fmaxf(Q < 8 ? p.a0[r + 1] : p.a1[r + 1], 0.0f);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float2 w2 = *reinterpret_cast<const float2*>(w64 + 128 * k + 32 * (Q >> 3) + 8 * (r >> 2) + 4 * h + (r & 3));
    rgb[k] = fmaf(x1, w2.y, fmaf(x0, w2.x, rgb[k]));
  }
}
__device__ __forceinline__ void rgb_q(int q, const Acc2& p, const float* w64, int h, float (&rgb)[3]) {
  switch (q) {
    case 0: rgb_pair<0>(p, w64, h, rgb); break;   case 1: rgb_pair<1>(p, w64, h, rgb); break;
    case 2: rgb_pair<2>(p, w64, h, rgb); break;   case 3: rgb_pair<3>(p, w64, h, rgb); break;
    case 4: rgb_pair<4>(p, w64, h, rgb); break;   case 5: rgb_pair<5>(p, w64, h, rgb); break;
    case 6: rgb_pair<6>(p, w64, h, rgb); break;   case 7: rgb_pair<7>(p, w64, h, rgb); break;
    case 8: rgb_pair<8>(p, w64, h, rgb); break;   case 9: rgb_pair<9>(p, w64, h, rgb); break;
    case 10: rgb_pair<10>(p, w64, h, rgb); break; case 11: rgb_pair<11>(p, w64, h, rgb); break;
    case 12: rgb_pair<12>(p, w64, h, rgb); break; case 13: rgb_pair<13>(p, w64, h, rgb); break;
    case 14: rgb_pair<14>(p, w64, h, rgb); break; case 15: rgb_pair<15>(p, w64, h, rgb); break;
    default: break;
  }
}

// head of the NEXT chunk (its first fragments + its bias fragment), prefetched in the last three k-steps
__device__ __forceinline__ void prefetch_next_chunk(Pre2& nxt, int k, const Loader& ld, unsigned bias_off) {
  prefetch_frag(nxt, k, ld.slot_next + ld.lane_off);
  if (k == 1) nxt.bias = lds_vec(ld.slot_next + bias_off + ld.lane_off)[0];
}

// One 256 -> 256 trunk layer L (1..8; 8 = xyz_encoding_final): in (bin) -> out (bout), four block-pair chunks.
// `pend` = the pair that finished last (blocks 6, 7 of the previous layer on entry).
template <bool BF>
__device__ __forceinline__ void trunk_layer(int L, u32x4 (&bin)[16], u32x4 (&bout)[16], const u32x4* stash, Loader& ld,
                                            const u32x4& bc0, const u32x4& bc1, Acc2& pend, Pre2& pre,
                                            const ChunkRef& after0, const ChunkRef& after1) {
  const unsigned lower = (L < 8) ? kRelu : kNoAct;
  const ChunkRef ref0 = layer_ref(L, 0, ld.wave);
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    ChunkRef c1 = ref0, c2 = ref0;
    c1.piece0 += ref0.pieces * (pb + 1);
    c2.piece0 += ref0.pieces * (pb + 2);
    if (pb == 3) c1 = after0;
    if (pb == 2) c2 = after0;
    if (pb == 3) c2 = after1;
    Acc2 cur;
    init_acc<BF>(cur, pre.bias, bc0, bc1);
    unsigned a_addr = ld.slot_cur + ld.lane_off;
    const unsigned next_bias = (unsigned)(c1.pieces - 1) * 1024u;
    Pre2 nxt;
    if (L == 4) {
      // skip connection: the encoded position (4 fragments per lane) was parked in LDS by the prologue
      u32x4 pe4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pe4[i] = stash[i * 64];
      Pre2 mid;
      pair_mma<BF, 4, -1>(
          cur, pre, a_addr, ld, c2, [&](int s) -> u32x4 { return pe4[s]; }, [&](int) {},
          [&](int k) { prefetch_frag(mid, k, a_addr + 8 * 1024); });
#pragma unroll
      for (int k = 0; k < kPF; ++k) { pre.f0[k] = mid.f0[k]; pre.f1[k] = mid.f1[k]; }
      a_addr += 8 * 1024;
    }
    pair_mma<BF, 16, kBar>(
        cur, pre, a_addr, ld, c2, [&](int s) -> u32x4 { return bin[s]; },
        [&](int s) {
          if (pb == 0)   // blocks 6, 7 of the previous layer (always relu'd) -> k-steps 12..15 of THIS layer's input
            pending_step<BF>(s, pend, kRelu, bin[12], bin[13], bin[14], bin[15]);
          else
            pending_step<BF>(s, pend, lower, bout[4 * pb - 4], bout[4 * pb - 3], bout[4 * pb - 2], bout[4 * pb - 1]);
        },
        [&](int k) { prefetch_next_chunk(nxt, k, ld, next_bias); });
    pend = cur;
    pre = nxt;
    loader_advance(ld);
  }
}

template <int MODE, bool SIGMA_ONLY, bool BF>
__global__ void __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(kWaves / 4, kWaves / 4)))
mlp_h1_kernel(const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ zv, int64_t P,
              int N, int stride, float* __restrict__ out, NsrTail tail) {
  // 3 x 41 KiB weight ring + per-wave stash of the encoded position (4 fragments x 64 lanes x 16 B) + colour head
  // (8 waves: 125,952 + 32,768 + 1,792 = 160,512 B of the 163,840)
  constexpr int kStash0 = 3 * kSlotFloats, kAux0 = kStash0 + kWaves * 4 * 256;
  __shared__ __attribute__((aligned(16))) float ring[kAux0 + kAuxFloats];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* aux = ring + kAux0;
  for (int i = threadIdx.x; i < kAuxFloats; i += 64 * kWaves) ring[kAux0 + i] = packed[kPiecesTotal * 256 + i];

  Loader ld;
  ld.stream = packed;
  ld.wave = wave;
  ld.lane_off = (unsigned)lane * 16u;
  ld.slot_cur = lds_addr(ring);
  ld.slot_next = ld.slot_cur + kSlotBytes;
  ld.slot_free = ld.slot_cur + 2 * kSlotBytes;
  loader_prepare_dma(ld, mkref(0, kL1Pieces, wave), ld.slot_cur);      // L1
#pragma unroll
  for (int i = 0; i < kMaxIssue; ++i) issue(ld, i);
  loader_prepare_dma(ld, layer_ref(1, 0, wave), ld.slot_next);         // first chunk of L2
#pragma unroll
  for (int i = 0; i < kMaxIssue; ++i) issue(ld, i);

  const int64_t p = (int64_t)blockIdx.x * kTile + wave * 32 + m;
  const int64_t pc = p < P ? p : P - 1;
  float pe[32], de[16];
  encode_point<MODE>(x, zv, pc, N, stride, h, pe, de);
  // status word (include/nsr.h): the fast paths check what enters and what leaves -- inputs inside the operand format's
  // range (an fp16 operand beyond 65,504 is inf; bf16 has fp32's range) and finite outputs; hidden activations are not
  // tracked here (NSR_F16 / NSR_BF16 are not parity paths)
  unsigned flags = 0u;
  {
    const float lim = BF ? 3.0e38f : 65504.0f;
    bool ok = true;
#pragma unroll
    for (int t = 0; t < 32; ++t) ok &= fabsf(pe[t]) <= lim;
#pragma unroll
    for (int t = 0; t < 16; ++t) ok &= fabsf(de[t]) <= lim;
    if (!ok) flags |= NSR_FLAG_INPUT_RANGE;
  }
  u32x4 pe4[4], de2[2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    pe4[s] = u32x4{pack2<BF>(pe[8 * s], pe[8 * s + 1]), pack2<BF>(pe[8 * s + 2], pe[8 * s + 3]),
                   pack2<BF>(pe[8 * s + 4], pe[8 * s + 5]), pack2<BF>(pe[8 * s + 6], pe[8 * s + 7])};
#pragma unroll
  for (int s = 0; s < 2; ++s)
    de2[s] = u32x4{pack2<BF>(de[8 * s], de[8 * s + 1]), pack2<BF>(de[8 * s + 2], de[8 * s + 3]),
                   pack2<BF>(de[8 * s + 4], de[8 * s + 5]), pack2<BF>(de[8 * s + 6], de[8 * s + 7])};
  u32x4* stash = reinterpret_cast<u32x4*>(ring + kStash0) + wave * 4 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 4; ++s) stash[s * 64] = pe4[s];

  u32x4 ba[16], bb[16];
  u32x4 bc0, bc1;
  bias_operands<BF>(h, bc0, bc1);
  Acc2 pend;
  Pre2 pre;

  // ---- L1: one chunk of four block pairs, 4 k-steps each; a pair is converted during the next pair
  {
    loader_publish(ld, layer_ref(1, 1, wave));     // chunk j+2 = second chunk of L2
    const unsigned a_chunk = ld.slot_cur + ld.lane_off;
    Pre2 nxt;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const unsigned a_addr = a_chunk + g * 8 * 1024;
      Pre2 mine;
#pragma unroll
      for (int k = 0; k < kPF; ++k) prefetch_frag(mine, k, a_addr);
      mine.bias = lds_vec(ld.slot_cur + (32 + g) * 1024 + ld.lane_off)[0];
      Acc2 cur;
      init_acc<BF>(cur, mine.bias, bc0, bc1);
      pair_mma<BF, 4, -1>(
          cur, mine, a_addr, ld, end_ref(wave), [&](int s) -> u32x4 { return pe4[s]; },
          [&](int s) {
            const int i = 4 * g + s;        // DMA of chunk j+2: one piece per k-step over the chunk's 16 k-steps
            if (i < kMaxIssue) issue(ld, i);
            if (g > 0) {                    // previous pair (blocks 2g-2, 2g-1): four register pairs per k-step
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4)
                conv_q<BF>(4 * s + q4, pend, kRelu, ba[4 * g - 4], ba[4 * g - 3], ba[4 * g - 2], ba[4 * g - 1]);
            }
          },
          [&](int k) {
            if (g == 3) prefetch_next_chunk(nxt, k, ld, 32u * 1024u);   // head of the first trunk chunk
          });
      pend = cur;
    }
    pre = nxt;
    loader_advance(ld);
  }

  // ---- L2..L8 (+ xyz_encoding_final), two layers per trip so the register sets swap roles
  constexpr int kPairs = SIGMA_ONLY ? 3 : 4;
#pragma unroll 1
  for (int pair = 0; pair < kPairs; ++pair) {
    const int L = 1 + 2 * pair;
    trunk_layer<BF>(L, ba, bb, stash, ld, bc0, bc1, pend, pre, layer_ref(L + 1, 0, wave), layer_ref(L + 1, 1, wave));
    const bool last = !SIGMA_ONLY && pair == kPairs - 1;
    const ChunkRef a0 = last ? sigma_ref(wave) : layer_ref(L + 2, 0, wave);
    const ChunkRef a1 = last ? dir_ref(0, wave) : layer_ref(L + 2, 1, wave);
    trunk_layer<BF>(L + 1, bb, ba, stash, ld, bc0, bc1, pend, pre, a0, a1);
  }
  if (SIGMA_ONLY) trunk_layer<BF>(7, ba, bb, stash, ld, bc0, bc1, pend, pre, sigma_ref(wave), end_ref(wave));

  // ---- density head over h8 (= bb: the input of xyz_encoding_final, still intact), paired with a zero block.
  // Pending: xyz_encoding_final's blocks 6, 7 (-> ba, no activation) or, sigma_only, L8's (-> bb, relu).
  float sigma;
  {
    Acc2 cur;
    init_acc<BF>(cur, pre.bias, bc0, bc1);
    Pre2 nxt;
    pair_mma<BF, 16, kBar>(
        cur, pre, ld.slot_cur + ld.lane_off, ld, SIGMA_ONLY ? end_ref(wave) : dir_ref(1, wave),
        [&](int s) -> u32x4 { return bb[s]; },
        [&](int s) {
          if (SIGMA_ONLY) pending_step<BF>(s, pend, kRelu, bb[12], bb[13], bb[14], bb[15]);
          else pending_step<BF>(s, pend, kNoAct, ba[12], ba[13], ba[14], ba[15]);
        },
        [&](int k) {
          if (!SIGMA_ONLY) prefetch_next_chunk(nxt, k, ld, 36u * 1024u);
        });
    sigma = cur.a0[0];
    pre = nxt;
    loader_advance(ld);
  }
  if (SIGMA_ONLY) {
    if (h == 0 && p < P) out[p] = sigma;
    if (!nsr_finite(sigma)) flags |= NSR_FLAG_OUTPUT_NONFINITE;
    if (p < P) nsr_raise(tail, flags);
    dma_drain();   // no LDS-DMA may be in flight when the workgroup's LDS is released
    return;
  }

  // ---- dir_encoding (cat([g, de]) -> 128, relu), two block pairs, fused with the rgb head (128 -> 3, sigmoid)
  float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    Acc2 cur;
    init_acc<BF>(cur, pre.bias, bc0, bc1);
    Pre2 nxt;
    pair_mma<BF, 18, kBar>(
        cur, pre, ld.slot_cur + ld.lane_off, ld, end_ref(wave),
        [&](int s) -> u32x4 { return (s < 16) ? ba[s & 15] : de2[s & 1]; },
        [&](int s) {
          if (pb > 0) rgb_q(s, pend, aux + kAuxRgbW, h, rgb);     // blocks 0, 1: one pair per k-step 0..15
        },
        [&](int k) {
          if (pb == 0) prefetch_next_chunk(nxt, k, ld, 36u * 1024u);
        });
    pend = cur;
    pre = nxt;
    loader_advance(ld);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) rgb_q(q, pend, aux + kAuxRgbW + 64, h, rgb);   // blocks 2, 3
  const unsigned opts = nsr_opts(tail);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s = rgb[k];
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
  if (h == 0 && p < P) reinterpret_cast<float4*>(out)[p] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
  dma_drain();     // no LDS-DMA may be in flight when the workgroup's LDS is released
}

template <int MODE, bool SIGMA_ONLY>
static int launch(bool bf, const void* packed, const float* x, const float* z, int64_t P, int N, int stride, float* out,
                  unsigned* tail_w, hipStream_t st) {
  const NsrTail tail{tail_w};
  const dim3 grid((unsigned)((P + kTile - 1) / kTile)), block(64 * kWaves);
  const float* pk = static_cast<const float*>(packed);
  if (bf) hipLaunchKernelGGL((mlp_h1_kernel<MODE, SIGMA_ONLY, true>), grid, block, 0, st, pk, x, z, P, N, stride, out, tail);
  else hipLaunchKernelGGL((mlp_h1_kernel<MODE, SIGMA_ONLY, false>), grid, block, 0, st, pk, x, z, P, N, stride, out, tail);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

}  // namespace h1

extern "C" NSR_INTERNAL size_t nsr_h1_packed_bytes(void) { return 4 * (size_t)(h1::kPiecesTotal * 256 + h1::kAuxFloats); }

extern "C" NSR_INTERNAL int nsr_h1_pack(int bf, const float* const* w, void* packed_dev, void* stream) {
  h1::PackPtrs pp;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w[i]) return NSR_ERR_INVALID_ARG;
    pp.p[i] = w[i];
  }
  const int total = h1::kPiecesTotal * 256 + h1::kAuxFloats;
  if (bf)
    hipLaunchKernelGGL(h1::pack_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, nsr_stream(stream), pp,
                       static_cast<unsigned*>(packed_dev));
  else
    hipLaunchKernelGGL(h1::pack_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, nsr_stream(stream), pp,
                       static_cast<unsigned*>(packed_dev));
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" NSR_INTERNAL int nsr_h1_mlp_forward(int bf, const void* packed, const float* x, int64_t P, int sigma_only,
                                               float* out, unsigned* tail, void* stream) {
  return sigma_only ? h1::launch<0, true>(bf != 0, packed, x, nullptr, P, 1, 8, out, tail, nsr_stream(stream))
                    : h1::launch<0, false>(bf != 0, packed, x, nullptr, P, 1, 8, out, tail, nsr_stream(stream));
}

extern "C" NSR_INTERNAL int nsr_h1_render_rays(int bf, const void* packed, const float* rays, int ray_stride,
                                               const float* z, int64_t R, int N, float* out, unsigned* tail, void* stream) {
  return h1::launch<1, false>(bf != 0, packed, rays, z, R * N, N, ray_stride, out, tail, nsr_stream(stream));
}
