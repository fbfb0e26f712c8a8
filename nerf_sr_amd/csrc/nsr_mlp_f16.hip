// M1 / D2 on the fp16 matrix pipe with split operands ("f16x3", NSR_F16X3):
// every fp32 value v is carried as hi = RNE_f16(v), lo = RNE_f16(v - hi) and each product is
// formed as a_hi*b_hi + (a_hi*b_lo + a_lo*b_hi) on v_mfma_f32_32x32x16_f16 with fp32
// accumulation, at 3/16 of the fp32-MFMA cycle cost.  Two details keep the split at its full
// 22 bits, which is what puts the rendered colours as close to the fp32 oracle as the fp32-MFMA
// kernel is (262,144 rays of the four BASELINE geometries: median |dRGB| 1e-7, worst non-exempt ray
// 8.3e-5; round 1, without them: 5e-7 / 1.3e-4):
//   * weights and biases are multiplied by 2^6 before they are split (exact): the lo part of a
//     typical weight (|w| ~ 0.05 -> lo ~ 2^-16) otherwise sits on fp16's subnormal floor (2^-24)
//     and loses three bits.  Accumulators therefore hold 64 x the layer output; the factor is
//     removed, exactly, inside the re-split of a finished block (an exponent subtract on the packed
//     fp16 pair, see "Re-split" below);
//   * hi is rounded to nearest (v_cvt_pk_f16_f32), so |lo| <= 2^-12 |v| (round 1: cvt_pkrtz, 2^-11 |v|).
//
// Same register algebra as the fp32 kernel (nsr_mlp_layout.h): a wave owns 32 sample
// points, activations never leave registers, weights stream global -> LDS by DMA.
// Differences: the stream is consumed one 32-feature OUTPUT block at a time (chunk =
// all k-steps of that block + its bias), so that a finished block is re-split into the
// next layer's hi/lo operand registers on the VALU while the next block occupies the
// matrix pipe; two register sets alternate roles layer by layer.
#include "nsr_f16x3_core.h"
#include "nsr_composite.h"
#include <type_traits>
#include <utility>

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


// one thread per 32-bit word of the blob
template <bool TWO>
__device__ __forceinline__ void pack_f16x3_body(const PackPtrsH& w, unsigned* __restrict__ out);
__global__ void __launch_bounds__(256) pack_f16x3_kernel(PackPtrsH w, unsigned* __restrict__ out) { pack_f16x3_body<false>(w, out); }
// both networks of a training step in one launch (round 6; blockIdx.y = network)
__global__ void __launch_bounds__(256) pack_f16x3_2_kernel(PackPtrsH w0, unsigned* __restrict__ out0, PackPtrsH w1, unsigned* __restrict__ out1) {
  if (blockIdx.y) pack_f16x3_body<true>(w1, out1);
  else pack_f16x3_body<true>(w0, out0);
}
template <bool TWO>
__device__ __forceinline__ void pack_f16x3_body(const PackPtrsH& w, unsigned* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int stream_words = kPiecesTotal * 256;
  if (idx >= stream_words + hx::kAuxFloats) return;
  unsigned v = 0u;
  if (idx < stream_words) {
    const int piece = idx >> 8, word = idx & 255;
    const Chunk c = chunk_info(chunk_of_piece(piece));
    const int local = piece - c.piece0;
    const int npieces = chunk_pieces(c.steps, c.nnb);
    if (local == npieces - 1) {
      // bias piece: fp32 bias (x 2^6, like the weights) of the chunk's output blocks, 32 per block
      const int n_rows = (c.tensor == 20) ? 1 : 32 * c.nnb;     // sigma head: one real output row
      if (word < n_rows) v = __float_as_uint(kWScale * w.p[c.tensor + 1][32 * c.nb0 + word]);
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
        const bool real_row = (c.tensor != 20) || n == 0;
        f[e] = (col == kPad || !real_row) ? 0.0f : kWScale * w.p[c.tensor][n * ld + col];
      }
      v = pack_hl(f[0], f[1], part);
    }
  } else {
    const int a = idx - stream_words;
    float f = 0.0f;
    if (a < hx::kAuxRgbB) f = kWInvScale * w.p[22][a];     // the colour head reads 64 x dir_encoding's output
    else if (a < hx::kAuxRgbB + 3) f = w.p[23][a - hx::kAuxRgbB];
    v = __float_as_uint(f);
  }
  out[idx] = v;
}

extern "C" NSR_INTERNAL size_t nsr_f16x3_packed_bytes(void) { return 4 * (size_t)(kPiecesTotal * 256 + hx::kAuxFloats); }

extern "C" NSR_INTERNAL int nsr_f16x3_pack(const float* const* w, void* packed_dev, void* stream) {
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

extern "C" NSR_INTERNAL int nsr_f16x3_pack2(const float* const* w0, void* packed0, const float* const* w1, void* packed1, void* stream) {
  PackPtrsH a, b;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w0[i] || !w1[i]) return NSR_ERR_INVALID_ARG;
    a.p[i] = w0[i];
    b.p[i] = w1[i];
  }
  const int total = kPiecesTotal * 256 + hx::kAuxFloats;
  hipLaunchKernelGGL(pack_f16x3_2_kernel, dim3((total + 255) / 256, 2), dim3(256), 0, nsr_stream(stream), a,
                     static_cast<unsigned*>(packed0), b, static_cast<unsigned*>(packed1));
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// trunk layer L (1..8: L2..L8, xyz_encoding_final), output block nb: chunk ids 2 + 8(L-1) + nb (see chunk_info)
__device__ __forceinline__ ChunkRef layer_ref(int L, int nb, int wave) {
  const int pieces = (L == 4) ? 41 : 33;
  const int base = (L <= 3) ? 66 + 264 * (L - 1) : (L == 4 ? 858 : 1186 + 264 * (L - 5));
  return make_ref(base + pieces * nb, pieces, wave);
}
__device__ __forceinline__ ChunkRef sigma_ref(int wave) { return make_ref(2242, 33, wave); }
__device__ __forceinline__ ChunkRef dir_ref(int nb, int wave) { return make_ref(2275 + 37 * nb, 37, wave); }
// past the end of the sequence chunk 0 is re-fetched into the idle slot (32 KiB of dead traffic, twice per
// tile) so that the first eight DMA issues of every chunk need no bounds test; the kernel drains before exit
__device__ __forceinline__ ChunkRef end_ref(int wave) { return make_ref(0, 32, wave); }
// chunks 0 and 1 of the stream (L1): what the last two chunks of a tile fetch for the next tile of the persistent loop
__device__ __forceinline__ ChunkRef first_ref(int c, int wave) { return make_ref(33 * c, 33, wave); }


// Re-split of the pending block: accumulator pair P (registers 2P, 2P+1; P = 0..7) -> activation -> (hi, lo) fp16
// pairs in the operand registers of the consuming layer: block nb becomes k-steps 2nb (P < 4 -> h0 / l0) and 2nb + 1
// (P >= 4 -> h1 / l1), element P & 3.  The accumulators hold 64 x the layer output (weight scale, see the header).
// Six VALU instructions per pair, only two of them on the slow mixed-precision path:
//   x0, x1 = max(acc, 0)                (raw v_max: fmaxf() would add a canonicalising v_max per operand)
//   hi64   = RNE_f16(x0), RNE_f16(x1)   (one v_cvt_pk_f16_f32)
//   hi     = hi64 / 64                  (one v_pk_sub_u16 ... clamp on the two exponent fields: 6 << 10 off each bit
//                                        pattern, saturating at 0.  Exact whenever the result is a normal number; for
//                                        |x| < 2^-8 the pattern lands in / below the subnormal encodings and simply
//                                        means another small number -- harmless, because lo is computed from the bits
//                                        hi actually holds, so hi + lo is x / 64 either way)
//   lo     = RNE_f16(x * 2^-6 - hi)     (v_fma_mixlo / mixhi with an f16 source; the fma result is exact in fp32)
// xyz_encoding_final has no ReLU and negative values do not survive the unsigned exponent trick: there hi comes from
// two v_fma_mix (RNE_f16(x * 2^-6)) instead of the cvt / sub pair.
// The work is issued as 17 HALF-STEPS of three instructions; the lo of pair P - 1 is interleaved with the hi of pair P.
// asm volatile pins each half-step into its k-step (MFMA shadow); LLVM would otherwise sink the work to its first
// use, i.e. serialise all eight blocks' conversions at the layer end.
struct PairTmp {
  float x0, x1;
  unsigned hi;
};
struct Resplit {
  PairTmp t[2];   // pairs alternate between the two sets: pair P - 1 is still live while pair P starts
};
template <int P>
__device__ __forceinline__ void put(unsigned v, u32x4& d0, u32x4& d1) {
  if (P < 4) d0[P & 3] = v; else d1[P & 3] = v;
}
#ifdef NSR_ABL_NO_AMAX   // ablation (scripts/ablate.sh): what the activation-range tracking costs
#define NSR_AMAX_RELU
#define NSR_AMAX_NOACT
#else
#define NSR_AMAX_RELU "\n\tv_max3_f32 %3, %0, %1, %3"
#define NSR_AMAX_NOACT "\n\tv_max3_f32 %3, |%0|, |%1|, %3"
#endif
// amax: running maximum of |64 x activation| over everything this lane re-splits (one v_max3 per pair): at 65,520 the
// hi part rounds to inf -> saturates at 1024 after the exponent subtract (NSR_FLAG_ACTIVATION_RANGE, include/nsr.h)
template <int P, bool RELU>
__device__ __forceinline__ void resplit_a(const Acc& p, float lower, Resplit& r, float& amax) {
  PairTmp& t = r.t[P & 1];
#ifdef NSR_ABL_NO_CONVERT
  asm volatile("v_max_f32 %0, %2, %4\n\tv_max_f32 %1, %3, %4" : "=&v"(t.x0), "=&v"(t.x1) : "v"(p.m[2 * P]), "v"(p.m[2 * P + 1]), "v"(lower));
  t.hi = __float_as_uint(t.x0);
#else
  if (RELU)
    asm volatile(
        "v_max_f32 %0, %4, 0\n\t"
        "v_max_f32 %1, %5, 0\n\t"
        "v_cvt_pk_f16_f32 %2, %0, %1"
        NSR_AMAX_RELU
        : "=&v"(t.x0), "=&v"(t.x1), "=&v"(t.hi), "+v"(amax)
        : "v"(p.m[2 * P]), "v"(p.m[2 * P + 1]));
  else
    asm volatile(
        "v_max_f32 %0, %4, %6\n\t"
        "v_max_f32 %1, %5, %6\n\t"
        "v_fma_mixlo_f16 %2, %0, %7, 0 op_sel_hi:[0,0,0]"
        NSR_AMAX_NOACT
        : "=&v"(t.x0), "=&v"(t.x1), "=&v"(t.hi), "+v"(amax)
        : "v"(p.m[2 * P]), "v"(p.m[2 * P + 1]), "v"(lower), "v"(kWInvScale));
#endif
}
template <int P, bool RELU>   // P = 0..8: finishes hi of pair P (P < 8) and makes lo of pair P - 1 (P > 0)
__device__ __forceinline__ void resplit_b(Resplit& r, u32x4& h0, u32x4& l0, u32x4& h1, u32x4& l1) {
  PairTmp& cur = r.t[P & 1];
  const PairTmp& prev = r.t[(P & 1) ^ 1];
#ifdef NSR_ABL_NO_CONVERT
  if (P < 8) put<(P < 8 ? P : 0)>(cur.hi, h0, h1);
  if (P > 0) put<(P > 0 ? P - 1 : 0)>(__float_as_uint(prev.x1), l0, l1);
#else
  constexpr unsigned kExp6 = 0x18001800u;   // 6 in both fp16 exponent fields
  unsigned lo = 0;
  if (P == 0) {
    if (RELU) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(cur.hi) : "v"(kExp6));
    else asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(cur.hi) : "v"(cur.x1), "v"(kWInvScale));
  } else if (P < 8) {
    if (RELU)
      asm volatile(
          "v_fma_mixlo_f16 %1, %3, %5, -%2 op_sel_hi:[0,0,1]\n\t"
          "v_pk_sub_u16 %0, %0, %6 clamp\n\t"
          "v_fma_mixhi_f16 %1, %4, %5, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
          : "+v"(cur.hi), "=&v"(lo)
          : "v"(prev.hi), "v"(prev.x0), "v"(prev.x1), "v"(kWInvScale), "v"(kExp6));
    else
      asm volatile(
          "v_fma_mixlo_f16 %1, %4, %6, -%3 op_sel_hi:[0,0,1]\n\t"
          "v_fma_mixhi_f16 %0, %2, %6, 0 op_sel_hi:[0,0,0]\n\t"
          "v_fma_mixhi_f16 %1, %5, %6, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
          : "+v"(cur.hi), "=&v"(lo)
          : "v"(cur.x1), "v"(prev.hi), "v"(prev.x0), "v"(prev.x1), "v"(kWInvScale));
  } else {
    asm volatile(
        "v_fma_mixlo_f16 %0, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo)
        : "v"(prev.hi), "v"(prev.x0), "v"(prev.x1), "v"(kWInvScale));
  }
  if (P < 8) put<(P < 8 ? P : 0)>(cur.hi, h0, h1);
  if (P > 0) put<(P > 0 ? P - 1 : 0)>(lo, l0, l1);
#endif
}
// half-step hs (0..16) of the pending block
template <bool RELU>
__device__ __forceinline__ void pending_half_t(int hs, const Acc& p, float lower, Resplit& r, u32x4& h0, u32x4& l0, u32x4& h1,
                                               u32x4& l1, float& amax) {
  switch (hs) {
#define NSR_HS(P)                                                  \
    case 2 * P: resplit_a<P, RELU>(p, lower, r, amax); break;      \
    case 2 * P + 1: resplit_b<P, RELU>(r, h0, l0, h1, l1); break;
    NSR_HS(0) NSR_HS(1) NSR_HS(2) NSR_HS(3) NSR_HS(4) NSR_HS(5) NSR_HS(6) NSR_HS(7)
#undef NSR_HS
    case 16: resplit_b<8, RELU>(r, h0, l0, h1, l1); break;
    default: break;
  }
}
template <bool RELU>
__device__ __forceinline__ void pending_half(int hs, const Acc& p, Resplit& r, u32x4& h0, u32x4& l0, u32x4& h1, u32x4& l1,
                                             float& amax) {
  pending_half_t<RELU>(hs, p, RELU ? 0.0f : -__builtin_inff(), r, h0, l0, h1, l1, amax);
}
// Schedule over the gaps of a 16-step chunk (block_mma3): half-step i (0..15) in gap 1 + (i & 1) of k-step i >> 1, the
// last one (two instructions) in gap 0 of k-step 8, next to that gap's fragment reads.  One half-step (3-4 VALU) per
// MFMA gap, finished before the publish point and the DMA gaps of k-steps 8..13, and long before the operands of k-steps
// 14, 15 (block 7 of the previous layer) are used.
template <bool RELU>
__device__ __forceinline__ void pending_gap(int s, int g, const Acc& p, Resplit& r, u32x4& h0, u32x4& l0, u32x4& h1, u32x4& l1,
                                            float& amax) {
  if (g == 0) {
    if (s == 8) pending_half<RELU>(16, p, r, h0, l0, h1, l1, amax);
    return;
  }
  const int i = 2 * s + g - 1;
  if (i < 16) pending_half<RELU>(i, p, r, h0, l0, h1, l1, amax);
}
// colour head: pair P of a finished dir_encoding block (relu) dotted with rgb row K; the weights come in through `w2`,
// read from LDS one k-step earlier (rgb_load) so that their latency is not exposed inside a 32-cycle gap
template <int P>
__device__ __forceinline__ float2 pair_rgb_w(const float* w32, int h, int k) {
  constexpr int r = 2 * P;                       // registers r, r+1 <-> features 8*(r>>2) + 4h + (r&3), +1
  return *reinterpret_cast<const float2*>(w32 + 128 * k + 8 * (r >> 2) + 4 * h + (r & 3));
}
template <int P>
__device__ __forceinline__ void pair_rgb_k(const Acc& p, float2 w2, float& acc) {
  const float x0 = fmaxf(p.m[2 * P], 0.0f);
  const float x1 = fmaxf(p.m[2 * P + 1], 0.0f);
  acc = fmaf(x1, w2.y, fmaf(x0, w2.x, acc));
}
// k-step s, gap g: channel g of pair s - S0 is accumulated, the weights of pair s - S0 + 1 are fetched
template <int S0>
__device__ __forceinline__ void rgb_gap(int s, int g, const Acc& p, const float* w32, int h, float (&rgb)[3], float2 (&w2)[3]) {
  const int P = s - S0;
  switch (P) {
#define NSR_RG(Q) case Q: pair_rgb_k<Q>(p, w2[g], rgb[g]); break;
    NSR_RG(0) NSR_RG(1) NSR_RG(2) NSR_RG(3) NSR_RG(4) NSR_RG(5) NSR_RG(6) NSR_RG(7)
#undef NSR_RG
    default: break;
  }
  switch (P + 1) {
#define NSR_RW(Q) case Q: w2[g] = pair_rgb_w<Q>(w32, h, g); break;
    NSR_RW(0) NSR_RW(1) NSR_RW(2) NSR_RW(3) NSR_RW(4) NSR_RW(5) NSR_RW(6) NSR_RW(7)
#undef NSR_RW
    default: break;
  }
}


// TRAIN: vector-memory operations EVERY storing block issues behind its last DMA piece: the two unit stores (every second
// block adds the sign dword; counting it would let the publish point's vmcnt leave a DMA piece in flight after the blocks
// that do not -- under-counting only waits for a store that is a whole chunk old)
constexpr int kTrainYoung = 2;
// TRAIN: dir_encoding's blocks feed the colour head from their fp32 accumulators (rgb_gap), so nothing makes their fp16
// operand form -- but the colour head's weight gradient wants it like any other layer's input: hi of pair P = RNE_f16 of
// relu(acc) / 64, the re-split's first half
template <int P>
__device__ __forceinline__ void relu_hi_pair(const Acc& p, u32x4& d0, u32x4& d1) {
  float x0, x1;
  unsigned hi;
  asm volatile(
      "v_max_f32 %0, %3, 0\n\t"
      "v_max_f32 %1, %4, 0\n\t"
      "v_cvt_pk_f16_f32 %2, %0, %1\n\t"
      "v_pk_sub_u16 %2, %2, %5 clamp"
      : "=&v"(x0), "=&v"(x1), "=&v"(hi)
      : "v"(p.m[2 * P]), "v"(p.m[2 * P + 1]), "v"(0x18001800u));
  put<P>(hi, d0, d1);
}
__device__ __forceinline__ void relu_hi_step(int P, const Acc& p, u32x4& d0, u32x4& d1) {
  switch (P) {
#define NSR_RH(Q) case Q: relu_hi_pair<Q>(p, d0, d1); break;
    NSR_RH(0) NSR_RH(1) NSR_RH(2) NSR_RH(3) NSR_RH(4) NSR_RH(5) NSR_RH(6) NSR_RH(7)
#undef NSR_RH
    default: break;
  }
}

constexpr int kConvStep0 = 6;   // colour head: pending dir block is consumed in k-steps 6..13 (one pair each)

// prefetch of the NEXT chunk (sequence position j+1) during the last three k-steps of chunk j
__device__ __forceinline__ void prefetch_next_chunk(Pre& nxt, int k, int g, const Loader& ld, unsigned bias_off, int h) {
  if (g == 0) prefetch_frag(nxt, k, ld.slot_next + ld.lane_off);
  if (g == 2 && k == 1) prefetch_bias(nxt, ld.slot_next + bias_off, h);
}

// ---------------------------------------------------------------------------
// Encoding of the NEXT tile's sample point in the matrix shadow of the current tile (persistent kernel, MODE 1).
// cast_rays + both positional encodings + the (hi, lo) split of a point are ~700 VALU instructions; at the top of a
// tile they run with the matrix pipe idle (5,700 cycles of a 150,000-cycle tile, profiles/r4_timeline_*.json).  Here the
// same arithmetic -- statement for statement nsr_sincos / split2, so the operands are bit-identical -- is cut into
// kEncPieces pieces of 3..6 instructions, one per k-step (gap 0, next to the two fragment reads) of
// xyz_encoding_final's eight chunks, the density block and the first dir_encoding block.  The encoded position goes to
// the wave's LDS stash (which the current tile stopped reading after L5), the encoded direction waits in 16 registers.
// Every piece ends in an empty asm volatile on what it produced: LLVM would otherwise sink the arithmetic to its first
// use (the end of the window).
// Piece program: 0 = loads (ray record, z; volatile), 1 = cast_rays + range flags, 2 = the raw-coordinate pairs, then seven
// groups (position frequencies 0..4, direction frequencies 0, 1) of 3 x 6 sincos pieces + 3 split pieces.
// ---------------------------------------------------------------------------
constexpr int kEncGroup0 = 3, kEncGroupPieces = 21, kEncPieces = kEncGroup0 + 7 * kEncGroupPieces;   // 150
// All state is scalars selected by compare chains (pick3 / put3 / put8): the piece index is a constant only AFTER the
// k-step loops are unrolled, and an array indexed by a not-yet-constant would send the struct to scratch memory -- or,
// with a 150-way switch per call site, push the loop bodies past the unroller's size limit.
struct Enc {
  float o0, o1, o2, d0, d1, d2, w0, w1, w2;   // ray record: origin, direction, the direction that is ENCODED
  float zk;
  float v0, v1, v2;                  // the point
  float x, k, r, r2, s0, t2, hh;     // sincos in flight
  int q;
  float a0, a1, a2, a3, a4, a5;      // sin c0..2, cos c0..2 of the current frequency
  unsigned flags;
  unsigned dh0, dh1, dh2, dh3, dh4, dh5, dh6, dh7;   // the encoded direction: pairs 0..7, hi and lo
  unsigned dl0, dl1, dl2, dl3, dl4, dl5, dl6, dl7;
  unsigned th, tl;                   // a split pair between its two half-pieces
};
struct EncIn {
  const float* rays;
  const float* zv;
  int64_t pc;        // the next tile's point of this lane (clamped)
  int64_t ray;
  int stride;
  int h;
  unsigned* stash;   // this lane's 16 B column of the wave's stash, viewed as words: fragment f, element e at stash[f * 256 + e]
};
typedef float vf32x4 __attribute__((ext_vector_type(4)));
#define NSR_PIN1(a) asm volatile("" : "+v"(a))
#define NSR_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define NSR_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
__device__ __forceinline__ float pick3(int c, float a, float b, float d) { return c == 0 ? a : (c == 1 ? b : d); }
__device__ __forceinline__ void put3(int c, float v, float& a, float& b, float& d) {
  if (c == 0) a = v; else if (c == 1) b = v; else d = v;
}
// (x0, x1) -> packed RNE fp16 pair hi and the RNE fp16 pair of the exact residuals (== split2)
__device__ __forceinline__ void split2_asm(float x0, float x1, unsigned& hi, unsigned& lo) {
  asm volatile(
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(hi), "=&v"(lo)
      : "v"(x0), "v"(x1));
}
__device__ __forceinline__ void enc_put_dir(Enc& e, int pair, unsigned hi, unsigned lo) {
  if (pair == 0) { e.dh0 = hi; e.dl0 = lo; }
  if (pair == 1) { e.dh1 = hi; e.dl1 = lo; }
  if (pair == 2) { e.dh2 = hi; e.dl2 = lo; }
  if (pair == 3) { e.dh3 = hi; e.dl3 = lo; }
  if (pair == 4) { e.dh4 = hi; e.dl4 = lo; }
  if (pair == 5) { e.dh5 = hi; e.dl5 = lo; }
  if (pair == 6) { e.dh6 = hi; e.dl6 = lo; }
  if (pair == 7) { e.dh7 = hi; e.dl7 = lo; }
}
// part: 0 / 1 = first / second half of the piece (2-4 instructions each: with the two fragment reads a gap then holds
// five fillers, what a lone wave hides behind one MFMA), 2 = the whole piece (first tile)
__device__ __forceinline__ void enc_piece(int i, int part, Enc& e, const EncIn& in) {
  if (i < 0 || i >= kEncPieces) return;
  const bool p0 = part != 1, p1 = part != 0;
  if (i == 0) {
    if (!p0) return;
    // plain loads, held in place by compiler-level memory fences (a volatile load would make hipcc wait for every
    // one of them on the spot); hipcc's own vmcnt wait lands at the first use, piece 1, most of a tile later
    asm volatile("" ::: "memory");
    const NsrRay q = nsr_load_ray(in.rays, in.ray, in.stride);
    e.zk = in.zv[in.pc];
    e.o0 = q.o[0]; e.o1 = q.o[1]; e.o2 = q.o[2]; e.d0 = q.d[0]; e.d1 = q.d[1]; e.d2 = q.d[2];
    e.w0 = q.v[0]; e.w1 = q.v[1]; e.w2 = q.v[2];
    asm volatile("" ::: "memory");
    return;
  }
  if (i == 1) {   // cast_rays (models/utils.py:14: separate multiply and add) + the operand-range check of the raw coordinates
    if (p0) {
      e.v0 = __fadd_rn(e.o0, __fmul_rn(e.zk, e.d0));
      e.v1 = __fadd_rn(e.o1, __fmul_rn(e.zk, e.d1));
      e.v2 = __fadd_rn(e.o2, __fmul_rn(e.zk, e.d2));
      NSR_PIN3(e.v0, e.v1, e.v2);
    }
    if (p1) {
      const bool ok = fabsf(e.v0) <= 65504.0f && fabsf(e.w0) <= 65504.0f && fabsf(e.v1) <= 65504.0f && fabsf(e.w1) <= 65504.0f &&
                      fabsf(e.v2) <= 65504.0f && fabsf(e.w2) <= 65504.0f;
      e.flags = ok ? 0u : (unsigned)NSR_FLAG_INPUT_RANGE;
      NSR_PIN1(e.flags);
    }
    return;
  }
  if (i == 2) {   // pairs 0 of both encodings: the raw coordinates (x, y | z, 0); pair 7 of the direction: padding
    unsigned hi, lo;
    if (p0) {
      split2_asm(in.h ? e.v2 : e.v0, in.h ? 0.0f : e.v1, hi, lo);
      in.stash[0] = hi;
      in.stash[4 * 256] = lo;
    }
    if (p1) {
      split2_asm(in.h ? e.w2 : e.w0, in.h ? 0.0f : e.w1, hi, lo);
      e.dh0 = hi; e.dl0 = lo;
      e.dh7 = 0u; e.dl7 = 0u;
    }
    return;
  }
  const int g = (i - kEncGroup0) / kEncGroupPieces, w = (i - kEncGroup0) % kEncGroupPieces;
  const bool pos = g < 5;
  const int f = pos ? g : g - 5;
  if (w < 18) {
    const int c = w / 6, sub = w % 6;
    // nsr_sincos (nsr_common.h), statement for statement
    if (sub == 0) {
      if (p0) {
        e.x = ldexpf(pos ? pick3(c, e.v0, e.v1, e.v2) : pick3(c, e.w0, e.w1, e.w2), (pos ? 5 : 2) * in.h + f);
        e.k = __fmul_rn(e.x, 0.636619772367581343f);
        NSR_PIN2(e.x, e.k);
      }
      if (p1) {
        e.k = rintf(e.k);
        e.r = fmaf(e.k, -1.57079625129699707031f, e.x);
        NSR_PIN2(e.k, e.r);
      }
    } else if (sub == 1) {
      if (p0) {
        e.r = fmaf(e.k, -7.54978941586159635335e-08f, e.r);
        e.r = fmaf(e.k, -5.39030252995776476554e-15f, e.r);
        NSR_PIN1(e.r);
      }
      if (p1) {
        e.q = (int)e.k;
        e.r2 = __fmul_rn(e.r, e.r);
        NSR_PIN2(e.q, e.r2);
      }
    } else if (sub == 2) {
      if (p0) {
        float sp = fmaf(e.r2, -1.9515295891e-4f, 8.3321608736e-3f);
        e.s0 = fmaf(sp, e.r2, -1.6666654611e-1f);
        NSR_PIN1(e.s0);
      }
      if (p1) {
        e.s0 = fmaf(__fmul_rn(e.s0, e.r2), e.r, e.r);
        NSR_PIN1(e.s0);
      }
    } else if (sub == 3) {
      if (p0) {
        float cp = fmaf(e.r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
        e.t2 = fmaf(cp, e.r2, 4.166664568298827e-2f);
        NSR_PIN1(e.t2);
      }
      if (p1) {
        e.t2 = __fmul_rn(e.t2, e.r2);
        e.hh = fmaf(-0.5f, e.r2, 1.0f);
        NSR_PIN2(e.t2, e.hh);
      }
    } else if (sub == 4) {
      if (p0) {
        e.hh = fmaf(e.t2, e.r2, e.hh);          // c0
        NSR_PIN1(e.hh);
      }
      if (p1) {
        const float ss = (e.q & 1) ? e.hh : e.s0;
        const float cc = (e.q & 1) ? e.s0 : e.hh;
        e.s0 = ss;
        e.t2 = cc;
        NSR_PIN2(e.s0, e.t2);
      }
    } else {
      if (p0) {
        float sn = (e.q & 2) ? -e.s0 : e.s0;
        NSR_PIN1(sn);
        put3(c, sn, e.a0, e.a1, e.a2);
      }
      if (p1) {
        float cs = ((e.q + 1) & 2) ? -e.t2 : e.t2;
        NSR_PIN1(cs);
        put3(c, cs, e.a3, e.a4, e.a5);
      }
    }
    return;
  }
  const int j = w - 18;                    // values 2j, 2j + 1 of the frequency -> pair 1 + 3f + j of the encoding
  const int pair = 1 + 3 * f + j;
  if (p0) split2_asm(pick3(j, e.a0, e.a2, e.a4), pick3(j, e.a1, e.a3, e.a5), e.th, e.tl);
  if (p1) {
    if (pos) {
      in.stash[(pair >> 2) * 256 + (pair & 3)] = e.th;
      in.stash[(4 + (pair >> 2)) * 256 + (pair & 3)] = e.tl;
    } else {
      enc_put_dir(e, pair, e.th, e.tl);
    }
  }
}
// Slot schedule: slot t = gap 0 of k-step t of the window [xyz_encoding_final | density | dir_encoding block 0] (128 + 16 +
// 18 slots) runs piece t - 2 whole.  The loads (piece 0) go out in L8's last chunk; their first use (slot kEncSlot0 =
// k-step 3 of a chunk) comes five k-steps after the previous chunk's last DMA piece, which has landed: hipcc's vmcnt(0)
// at that use finds nothing to wait for.  (Measured alternatives, same box: half-pieces spread over L8 .. dir block 1,
// two per piece: the state then lives through sixteen chunks at 420 registers, hipcc parks it in AGPRs and scratch,
// +1.6 %; whole pieces here: the reference point.)
constexpr int kEncSlot0 = 3;
static_assert(kEncSlot0 + kEncPieces - 2 < 128 + 16 + 18, "the encoding does not fit its window");
__device__ __forceinline__ void enc_slot(int t, Enc& e, const EncIn& in) {
  if (t >= kEncSlot0) enc_piece(t - kEncSlot0 + 1, 2, e, in);
}
// pieces I0 .. I1 - 1 back to back (the workgroup's first tile): recursion, not a loop -- a 150-trip loop is not unrolled,
// and a run-time piece index turns the compare chains above into scratch-memory arrays
template <int I0, int I1>
__device__ __forceinline__ void enc_all(Enc& e, const EncIn& in) {
  if constexpr (I0 < I1) {
    enc_piece(I0, 2, e, in);
    enc_all<I0 + 1, I1>(e, in);
  }
}

// One 256 -> 256 trunk layer L (1..8; 8 = xyz_encoding_final): in (bh, bl) -> out (oh, ol).
// L == 4 prepends the 4 positional-encoding k-steps (skip connection).  `pend` is the block that
// finished last (block 7 of the previous layer on entry; block 7 of this layer on exit): it is
// activated and re-split in the MFMA shadow of the FOLLOWING block's k-steps 0..13.  `pre` carries the
// prefetched head of the next chunk across chunk (and layer) boundaries.
// TRAIN: the pending block's fp16 `hi` operand registers -- the activation the next layer multiplies, which is also the
// operand of the weight gradients -- are written to the training panels (panel L - 1 holds trunk layer L's output, see
// nsr_f16x3_core.h): its two units as two 16-byte stores in k-steps 14 and 15 (+ the block's sign word), i.e. BEHIND the
// chunk's last DMA piece (k-step 13).  Vector-memory operations complete in issue order, so the next publish point --
// which must see that DMA landed -- may leave those stores in flight (block_mma's YOUNGER); they have a whole further
// chunk to reach HBM.  (Rounds 2-4 stored the sixteen raw fp32 accumulators of a block here.)
// ENC: the layer's gap-0 slots also carry encoding pieces of the next tile's point: 1 (L8) = the loads (piece 0) in the
// last chunk, 2 (xyz_encoding_final) = slots 16 nb + s of the piece schedule (enc_slot).
template <bool RELU_OUT, bool TRAIN = false, int ENC = 0>   // relu on L2..L8 (true), none on xyz_encoding_final (L == 8: false)
__device__ __forceinline__ void trunk_layer(int L, u32x4 (&bh)[16], u32x4 (&bl)[16], u32x4 (&oh)[16], u32x4 (&ol)[16],
                                            const u32x4* stash, Loader& ld, int h, Acc& pend, Pre& pre,
                                            const ChunkRef& after0, const ChunkRef& after1, float& amax, unsigned& sbits,
                                            const PanelRef& tr = PanelRef{}, unsigned voff0 = 0, unsigned voff1 = 0, Enc* enc = nullptr,
                                            const EncIn* ein = nullptr
#ifdef NSR_ABL_TIMELINE
                                            , unsigned long long* ld_tk_buf = nullptr
#endif
                                            ) {
  const ChunkRef ref0 = layer_ref(L, 0, ld.wave);           // this layer's chunks: piece0 advances by `pieces`
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    // chunks j+1 (prefetched from at the end of this one) and j+2 (DMA'd during this one)
    ChunkRef c1 = ref0, c2 = ref0;
    c1.piece0 += ref0.pieces * (nb + 1);
    c2.piece0 += ref0.pieces * (nb + 2);
    if (nb == 7) c1 = after0;
    if (nb == 6) c2 = after0;
    if (nb == 7) c2 = after1;
    Acc cur;
    cur.m = pre.bias;
    Resplit ptmp;
    unsigned a_addr = ld.slot_cur + ld.lane_off;
#ifdef NSR_ABL_TIMELINE
    ld.tk = (L == 7 && nb == 3) ? ld_tk_buf : nullptr;
    if (L == 7 && nb == 4) ld_tk_buf[5] = tl_now();
#endif
    const unsigned next_bias = (unsigned)(c1.pieces - 1) * 1024u;
    Pre nxt;
    if (L == 4) {
      // skip connection: the encoded position was parked in LDS by the prologue (8 fragments per lane);
      // its 4 k-steps run first and hand the act part's first fragments over through `mid`
      u32x4 pe8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pe8[i] = stash[i * 64];
      Pre mid;
      block_mma3<4, -1>(
          cur, pre, a_addr, ld, c2, [&](int s, int part) -> u32x4 { return pe8[4 * part + s]; }, [&](int, int) {},
          [&](int k, int g) {
            if (g == 0) prefetch_frag(mid, k, a_addr + 8 * 1024);
          });
#pragma unroll
      for (int k = 0; k < kPF; ++k) {
        pre.ah[k] = mid.ah[k];
        pre.al[k] = mid.al[k];
      }
      a_addr += 8 * 1024;
    }
    block_mma3<16, kBar, (TRAIN ? kTrainYoung : 0)>(
        cur, pre, a_addr, ld, c2, [&](int s, int part) -> u32x4 { return part ? bl[s] : bh[s]; },
        [&](int s, int g) {
          if (ENC == 1 && g == 0 && nb == 7 && s == 0) enc_piece(0, 2, *enc, *ein);
          if (ENC == 2 && g == 0) enc_slot(16 * nb + s, *enc, *ein);
          if (nb == 0)
            // block 7 of the previous layer (always relu'd: the previous layer is L1..L7) -> k-steps 14, 15
            // of THIS layer's input, needed only at the end of this chunk
            pending_gap<true>(s, g, pend, ptmp, bh[14], bl[14], bh[15], bl[15], amax);
          else
            pending_gap<RELU_OUT>(s, g, pend, ptmp, oh[2 * nb - 2], ol[2 * nb - 2], oh[2 * nb - 1], ol[2 * nb - 1], amax);
          if (TRAIN && s >= 8 && g == 2) {
            const char* blk = (nb == 0) ? panel_block(tr, L - 1, 7) : panel_block(tr, L, nb - 1);
            // the pending block's hi registers were finished in k-step 7 (half-step 15)
            if (nb == 0) {
              if (s == 14) unit_store<0>(bh[14], blk, voff0);
              if (s == 15) unit_store<1>(bh[15], blk, voff1);
            } else {
              if (s == 14) unit_store<0>(oh[2 * nb - 2], blk, voff0);
              if (s == 15) unit_store<1>(oh[2 * nb - 1], blk, voff1);
            }
            // sign bits of the pending block if a ReLU follows it (block 7 of the previous layer always does; this layer's
            // blocks do unless it is xyz_encoding_final); `sbits` runs on across blocks and layers, the dword is stored when
            // its second (odd) block is complete
            if (nb == 0 || RELU_OUT) {
              sign_push(sbits, pend.m[2 * (s - 8)]);
              sign_push(sbits, pend.m[2 * (s - 8) + 1]);
              if (s == 15 && (nb == 0 || ((nb - 1) & 1)))
                sign_store(sbits, (nb == 0) ? sign_block(tr.sgn, tr.group, L - 1, 7) : sign_block(tr.sgn, tr.group, L, nb - 1),
                           ld.lane_off >> 2);
            }
          }
        },
        [&](int k, int g) { prefetch_next_chunk(nxt, k, g, ld, next_bias, h); },
        // the block before the first trunk block is L1's last one, whose stores interleave with its DMA
        TRAIN && L == 1 && nb == 0);
    pend = cur;
    pre = nxt;
    loader_advance(ld);
  }
}

// NSR_ABL_TIMELINE (development build only, scripts/timeline.py): s_memtime stamps at the phase boundaries of every
// workgroup's wave, fetched with nsr_dbg_timeline(); never defined in the product build.
#ifdef NSR_ABL_TIMELINE
constexpr int kTlGroups = 65536, kTlSlots = 10;
__device__ unsigned long long nsr_tl[kTlGroups * 4 * kTlSlots];
__device__ unsigned long long nsr_tk[kTlGroups * 4 * 8];
#define NSR_TL(k) (tl[k] = tl_now())
extern "C" int nsr_dbg_ksteps(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nsr_tk), bytes < sizeof(nsr_tk) ? bytes : sizeof(nsr_tk), 0, hipMemcpyDeviceToHost);
}
extern "C" int nsr_dbg_timeline(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nsr_tl), bytes < sizeof(nsr_tl) ? bytes : sizeof(nsr_tl), 0, hipMemcpyDeviceToHost);
}
#else
#define NSR_TL(k) ((void)0)
#endif

// COMP: the tile's points are whole rays (MODE 1, NS = 64 or 128) and the kernel composites them itself (V1 fused into
// D2 + M1: the (R, N, 4) network output never goes to HBM); `out` may then be null.
// TRAIN: the forward pass of the training step (nsr_train.hip): additionally keeps every layer's pre-activations for the
// backward pass in the training panels `pan` (n_groups = 4 * n_tiles point groups).
//
// PERSISTENT TILE LOOP (round 4).  A workgroup takes tiles blockIdx.x, blockIdx.x + gridDim.x, ... of 128 points; the ray
// kernels (MODE 1, not TRAIN) are launched with one workgroup per CU.  Across a tile boundary
//   * the weight ring keeps streaming: the last two chunks of a tile fetch chunks 0 and 1 of the stream (the next tile's
//     L1) instead of idling, and the last block prefetches L1's first fragments: no DMA drain, no cold start;
//   * the next tile's encoding (cast_rays, 21 sincos, hi/lo split: the old 5,700-cycle prologue) has already run in the
//     matrix shadow of xyz_encoding_final / density / the first dir_encoding block (enc_piece above);
//   * the compositing epilogue has its own 2.5 KiB of LDS, so the ring is never drained for it;
//   * workgroup dispatch, the colour-head block load and the first-chunk DMA wait are paid once per CU, not per tile.
template <int MODE, bool SIGMA_ONLY, int NS, bool COMP = false, bool TRAIN = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_f16x3_kernel(const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ zv,
                 int64_t P, int N, int stride, float* __restrict__ out, NsrTail tail, NsrCompOut co = NsrCompOut{},
                 float* pan = nullptr, unsigned* sgn = nullptr) {
  // 3 x 41 KiB weight ring + per-wave stash of the encoded position (8 fragments x 64 lanes x 16 B = 8 KiB
  // per wave) + the colour-head block (rgb weights and bias, 448 floats) + the compositor's staging area (640 floats):
  // 163,072 B of the CU's 163,840
  constexpr int kStash0 = 3 * kSlotFloats, kAux0 = kStash0 + 4 * 8 * 256, kComp0 = kAux0 + hx::kAuxFloats;
  __shared__ __attribute__((aligned(16))) float ring[kComp0 + 640];
  // NSR_PERSISTENT (build flag, off in the product): the ray kernels loop over tiles, one workgroup per CU.
  // NSR_ENC_OVERLAP (with it): the next tile's encoding rides in this tile's matrix shadow.  Both are complete and
  // bit-identical to the default, and both measured no faster on this power-limited kernel (DESIGN section 3.1,
  // profiles/r4_persistent_ab.json): same box, fine pass 61.8 ms one tile per workgroup / 61.9 persistent / 62.2 with the
  // overlapped encoding.
#ifdef NSR_PERSISTENT
  constexpr bool PERSIST = (MODE == 1) && !TRAIN && !SIGMA_ONLY;
#else
  constexpr bool PERSIST = false;
#endif
#ifdef NSR_ENC_OVERLAP
  constexpr bool OVERLAP = PERSIST;
#else
  constexpr bool OVERLAP = false;
#endif
  const int lane = threadIdx.x & 63;
  int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* aux = ring + kAux0;          // LDS copy, visible after the first barrier
#ifdef NSR_ABL_TIMELINE
  unsigned long long tl[8], tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int i = threadIdx.x; i < hx::kAuxFloats; i += 256) ring[kAux0 + i] = packed[kPiecesTotal * 256 + i];

  Loader ld;
  ld.stream = packed;
  ld.wave = wave;
  ld.lane_off = (unsigned)lane * 16u;
#ifdef NSR_ABL_TIMELINE
  ld.tk = nullptr;
#endif
  ld.slot_cur = lds_addr(ring);
  ld.slot_next = ld.slot_cur + kSlotBytes;
  ld.slot_free = ld.slot_cur + 2 * kSlotBytes;
  // chunks 0 and 1 stream in behind the first tile's encoding
  loader_prepare_dma(ld, first_ref(0, wave), ld.slot_cur);
#pragma unroll
  for (int i = 0; i < 11; ++i) loader_issue(ld, i);
  loader_prepare_dma(ld, first_ref(1, wave), ld.slot_next);
#pragma unroll
  for (int i = 0; i < 11; ++i) loader_issue(ld, i);

  const int64_t n_tiles = (P + 127) / 128;
  const int samples = (NS > 0) ? NS : N;
  // colour-head options of the packed network (nsr_common.h); TRAIN: word 1 of the step's status block, of which the kernel
  // honours the activation only (the step applies --gamma_correct itself, between this launch and the compositor)
  const unsigned opts = nsr_opts(tail) & (TRAIN ? kOptColorNone : ~0u);
  // (the inference instantiations never use them; their voff0 is the expression rounds 2-4 had here, which keeps the register
  // allocation -- and with it the whole ISA of those kernels -- bit-identical to the measured round-4 build)
  const unsigned voff0 = TRAIN ? unit_voff(m, h, 0) : 4u * (unsigned)(m + 128 * h), voff1 = TRAIN ? unit_voff(m, h, 1) : 0u;   // panel stores: this lane's slot in either unit of a block
  // the wave's stash of the split position encoding: L5 (skip) re-reads it, which frees 32 registers in the loop
  u32x4* stash = reinterpret_cast<u32x4*>(ring + kStash0) + wave * 8 * 64 + lane;
  u32x4 deh[2], del[2];
  unsigned flags = 0u;     // NSR_FLAG_* of this lane's point, raised once at the end of its tile
  Enc enc;
  EncIn ein;
  ein.rays = x;
  ein.zv = zv;
  ein.stride = stride;
  ein.h = h;
  ein.stash = reinterpret_cast<unsigned*>(stash);
  if (OVERLAP) {   // the workgroup's first tile: the one encoding that is exposed
    const int64_t p0 = (int64_t)blockIdx.x * 128 + wave * 32 + m;
    ein.pc = p0 < P ? p0 : P - 1;
    ein.ray = ein.pc / samples;
    enc_all<0, kEncPieces>(enc, ein);
    deh[0] = u32x4{enc.dh0, enc.dh1, enc.dh2, enc.dh3};
    deh[1] = u32x4{enc.dh4, enc.dh5, enc.dh6, enc.dh7};
    del[0] = u32x4{enc.dl0, enc.dl1, enc.dl2, enc.dl3};
    del[1] = u32x4{enc.dl4, enc.dl5, enc.dl6, enc.dl7};
    flags = enc.flags;
  }
  Pre l1pre;
  bool first_tile = true;

  // the kernels that do not overlap the encoding are launched with one tile per workgroup and take the body once: no
  // loop-carried state, the register budget of the straight-line kernel (their panel stores take SGPR bases in inline asm)
  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) {
    dma_drain();
    return;
  }
#pragma unroll 1
  do {
#ifndef NSR_ABL_NO_TILE_LAUNDER
  // chunk descriptors derive from ld.wave: making it opaque once per tile keeps hipcc from hoisting seventy descriptor
  // offsets out of the tile loop (and spilling them to VGPR lanes)
  if (PERSIST) {
    asm volatile("" : "+s"(wave));
    ld.wave = wave;
  }
#endif
  NSR_TL(0);
  const int64_t p = tile * 128 + wave * 32 + m;
  const int64_t pc = p < P ? p : P - 1;
  PanelRef tr{};
  if (TRAIN) {
    tr.base = reinterpret_cast<char*>(pan);
    tr.n_groups = n_tiles * 4;
    tr.group = tile * 4 + wave;
    tr.sgn = sgn;
  }
  float amax = 0.0f;       // see resplit_a
  u32x4 peh[4], pel[4];
  if (OVERLAP) {
    // this tile's encoding was made during the previous tile (or above); the next tile's is made during this one
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      peh[s] = stash[s * 64];
      pel[s] = stash[(4 + s) * 64];
    }
    const int64_t pn = p + (int64_t)gridDim.x * 128;
    ein.pc = pn < P ? pn : P - 1;          // past the last tile: a harmless re-encoding of the last point
    ein.ray = ein.pc / samples;
  } else {
  // ---- encoding at the top of the tile (VanillaMLP.forward's embedded rows; the training step)
  flags = 0u;
  float pe[32], de[16];
  if (MODE == 0) {
    const float* row = x + pc * kInCh;
    bool ok = true;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const int col = pecol(t, h);
      pe[t] = (col == kPad) ? 0.0f : row[col];
      ok &= fabsf(pe[t]) <= 65504.0f;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int col = dircol(t, h);
      de[t] = (col == kPad) ? 0.0f : row[kPosCh + col];
      ok &= fabsf(de[t]) <= 65504.0f;
    }
    if (!ok) flags |= NSR_FLAG_INPUT_RANGE;
  } else {
    const int64_t ray = pc / ((NS > 0) ? NS : N);
    const NsrRay rq = nsr_load_ray(x, ray, stride);
    const float zk = zv[pc];
    const float d[3] = {rq.v[0], rq.v[1], rq.v[2]};           // the direction that is ENCODED
    const float v[3] = {__fadd_rn(rq.o[0], __fmul_rn(zk, rq.d[0])), __fadd_rn(rq.o[1], __fmul_rn(zk, rq.d[1])),
                        __fadd_rn(rq.o[2], __fmul_rn(zk, rq.d[2]))};
    // the raw coordinates are operands too (columns 0..2 of both encodings): fp16's range, and false for NaN
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 3; ++c) ok = ok && fabsf(v[c]) <= 65504.0f && fabsf(d[c]) <= 65504.0f;
    if (!ok) flags |= NSR_FLAG_INPUT_RANGE;
    pe[0] = h ? v[2] : v[0];
    pe[1] = h ? 0.0f : v[1];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float sn, cs;
        nsr_sincos(ldexpf(v[c], 5 * h + f), sn, cs);
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


  if (TRAIN) {   // the encodings themselves (their hi halves): operands of the weight gradients of L1, L5 (skip part) and
                 // dir_encoding; the direction panel's second 32 rows are zeros (the narrowest weight-gradient tile is 64)
    const char* pblk = panel_block(tr, 10, 0);
    const char* dblk = panel_block(tr, 11, 0);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    unit_store<0>(peh[0], pblk, voff0);
    unit_store<1>(peh[1], pblk, voff1);
    unit_store<0>(peh[2], pblk + 2048, voff0);
    unit_store<1>(peh[3], pblk + 2048, voff1);
    unit_store<0>(deh[0], dblk, voff0);
    unit_store<1>(deh[1], dblk, voff1);
    unit_store<0>(zero4, dblk + 2048, voff0);
    unit_store<1>(zero4, dblk + 2048, voff1);
  }

  // park the split position encoding in LDS for L5
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    stash[s * 64] = peh[s];
    stash[(4 + s) * 64] = pel[s];
  }
  }   // !OVERLAP
  if (!PERSIST || first_tile) {
    // the workgroup's first tile: chunks 0 and 1 (streamed in behind the encoding) + the colour-head block are in LDS
    // for everybody; the head of L1's first block.  Later tiles get all of this from their predecessor.
    dma_drain();
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPF; ++k) prefetch_frag(l1pre, k, ld.slot_cur + ld.lane_off);
    prefetch_bias(l1pre, ld.slot_cur + 32 * 1024, h);
    first_tile = false;
  }

  u32x4 bh[16], bl[16], oh[16], ol[16];
  Acc pend;
  Pre pre;
  unsigned sbits = 0u;     // TRAIN: sign bits of the pending blocks, two blocks per stored dword (nsr_f16x3_core.h)

  NSR_TL(1);
  // ---- L1: two chunks of four output blocks, 4 k-steps each; block b is re-split during block b+1 (17 half-steps
  // over the block's 12 MFMA gaps: 1 / 2 / 1-2 per gap), and the head of block b+1 (first fragments + bias) is read
  // during block b's last three k-steps.  Publish point at the chunk start (chunk 0 / 1 were issued above; chunk j+2
  // is fetched here, one piece per k-step).
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    // c == 0: publishes chunk 1 (fetched during the previous tile's last block, or before the loop) and frees the slot
    // of that block; chunk 0 was published there, and its head is already in `l1pre`
    loader_publish(ld, layer_ref(1, c, wave));    // chunk j+2 = first / second chunk of L2
    const unsigned a_chunk = ld.slot_cur + ld.lane_off;
    const unsigned next_bias = 32u * 1024u;       // L1 chunk 1 and L2's chunks: 32 weight pieces, then the bias
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = 4 * c + g;
      const unsigned a_addr = a_chunk + g * 8 * 1024;
      Pre nxt;
      Acc cur;
      cur.m = l1pre.bias;
      Resplit ptmp;
      block_mma3<4, -1>(
          cur, l1pre, a_addr, ld, first_ref(0, wave), [&](int s, int part) -> u32x4 { return part ? pel[s] : peh[s]; },
          [&](int s, int gp) {
            const int i = 4 * g + s;        // DMA of chunk j+2: one piece per k-step over the chunk's 16 k-steps
            if (gp == 0 && i < 11) loader_issue(ld, i);
            if (nb > 0) {
              // half-steps of the pending block: k-step 0 takes 0 | 1, 2 | 3, 4; k-step s >= 1 takes 4s+1 | 4s+2, 4s+3 | 4s+4
              const int first = (s == 0) ? (gp == 0 ? 0 : 2 * gp - 1) : 4 * s + (gp == 0 ? 1 : (gp == 1 ? 2 : 4));
              const int count = (gp == 1 || (s == 0 && gp == 2)) ? 2 : 1;
#pragma unroll
              for (int q = 0; q < count; ++q)
                pending_half<true>(first + q, pend, ptmp, bh[2 * nb - 2], bl[2 * nb - 2], bh[2 * nb - 1], bl[2 * nb - 1], amax);
              if (TRAIN && gp == 2) {
                const char* blk = panel_block(tr, 0, nb - 1);
                // hi of pairs 0..3 is complete after k-step 1 (half-step 7), of pairs 4..7 after gap 1 of k-step 3 (half-step 15)
                if (s == 2) unit_store<0>(bh[2 * nb - 2], blk, voff0);
                if (s == 3) unit_store<1>(bh[2 * nb - 1], blk, voff1);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) sign_push(sbits, pend.m[4 * s + q4]);
                if (s == 3 && ((nb - 1) & 1)) sign_store(sbits, sign_block(tr.sgn, tr.group, 0, nb - 1), ld.lane_off >> 2);
              }
            }
          },
          [&](int k, int gp) {
            // head of the next block: same chunk | chunk 1 (published together with chunk 0) | the first trunk chunk
            // (published at this chunk's start)
            const unsigned frag = (g < 3) ? a_addr + 8 * 1024 : ld.slot_next + ld.lane_off;
            const unsigned bias = (g < 3) ? ld.slot_cur + 32 * 1024 + 128 * (g + 1)
                                          : ld.slot_next + (c == 0 ? 32u * 1024u : next_bias);
            if (gp == 0) prefetch_frag(nxt, k, frag);
            if (gp == 2 && k == 1) prefetch_bias(nxt, bias, h);
          });
      pend = cur;
      l1pre = nxt;
    }
    loader_advance(ld);
  }
  pre = l1pre;

  NSR_TL(2);
  // ---- L2..L8 (+ xyz_encoding_final), two layers per trip so the register sets swap roles.  Three trips of relu
  // layers (L2..L7); the last pair (L8 + xyz_encoding_final, which has no relu) is peeled so that the activation is a
  // compile-time property of every re-split.
#pragma unroll 1
  for (int pair = 0; pair < 3; ++pair) {
    const int L = 1 + 2 * pair;
    trunk_layer<true, TRAIN>(L, bh, bl, oh, ol, stash, ld, h, pend, pre, layer_ref(L + 1, 0, wave), layer_ref(L + 1, 1, wave), amax, sbits, tr, voff0, voff1);
    trunk_layer<true, TRAIN>(L + 1, oh, ol, bh, bl, stash, ld, h, pend, pre, layer_ref(L + 2, 0, wave), layer_ref(L + 2, 1, wave), amax,
                             sbits, tr, voff0, voff1);
  }
  if (SIGMA_ONLY) {   // xyz_encoding_final is not evaluated: L8 is followed by the density head, then nothing
    trunk_layer<true>(7, bh, bl, oh, ol, stash, ld, h, pend, pre, sigma_ref(wave), first_ref(0, wave), amax, sbits);
  } else {
#ifdef NSR_ABL_TIMELINE
    trunk_layer<true, TRAIN, (OVERLAP ? 1 : 0)>(7, bh, bl, oh, ol, stash, ld, h, pend, pre, layer_ref(8, 0, wave), layer_ref(8, 1, wave), amax, sbits, tr, voff0, voff1, &enc, &ein, tk);
    ld.tk = nullptr;
#else
    trunk_layer<true, TRAIN, (OVERLAP ? 1 : 0)>(7, bh, bl, oh, ol, stash, ld, h, pend, pre, layer_ref(8, 0, wave), layer_ref(8, 1, wave), amax, sbits, tr, voff0, voff1, &enc, &ein);
#endif
    trunk_layer<false, TRAIN, (OVERLAP ? 2 : 0)>(8, oh, ol, bh, bl, stash, ld, h, pend, pre, sigma_ref(wave), dir_ref(0, wave), amax, sbits, tr, voff0, voff1, &enc, &ein);
  }

  NSR_TL(3);
  // ---- density head: sigma.weight as row 0 of one more 32-row block over h8 (= oh/ol: the input of
  // xyz_encoding_final, still intact).  The pending block is xyz_encoding_final's last one (-> bh, no
  // activation), or L8's last one in a sigma_only launch (-> oh, relu).
  float sigma;
  {
    Acc cur;
    cur.m = pre.bias;
    Resplit ptmp;
    const unsigned next_bias = 36u * 1024u;       // dir_encoding chunks: 36 weight pieces, then the bias
    Pre nxt;
#ifdef NSR_ABL_NO_DENSITY_MMA   // ablation (profiles/r5_headline_experiments.json): the density block without its 48 MFMAs
    constexpr bool kDensityMma = false;
#else
    constexpr bool kDensityMma = true;
#endif
    block_mma3<16, kBar, (TRAIN ? kTrainYoung : 0), true, kDensityMma>(
        cur, pre, ld.slot_cur + ld.lane_off, ld, SIGMA_ONLY ? first_ref(1, wave) : dir_ref(1, wave),
        [&](int s, int part) -> u32x4 { return part ? ol[s] : oh[s]; },
        [&](int s, int g) {
          if (OVERLAP && g == 0) enc_slot(128 + s, enc, ein);
          if (SIGMA_ONLY)
            pending_gap<true>(s, g, pend, ptmp, oh[14], ol[14], oh[15], ol[15], amax);
          else
            pending_gap<false>(s, g, pend, ptmp, bh[14], bl[14], bh[15], bl[15], amax);
          if (TRAIN && s >= 14 && g == 2) {   // xyz_encoding_final's last block (no sign bits: nothing is masked by it)
            const char* blk = panel_block(tr, 8, 7);
            if (s == 14) unit_store<0>(bh[14], blk, voff0);
            else unit_store<1>(bh[15], blk, voff1);
          }
        },
        [&](int k, int g) {
          // sigma_only: the tile ends here, the next chunk is the next tile's L1 chunk 0 (bias behind 32 weight pieces)
          prefetch_next_chunk(nxt, k, g, ld, SIGMA_ONLY ? 32u * 1024u : next_bias, h);
        });
    sigma = cur.m[0] * kWInvScale;           // row 0 of the block lives in register 0 of the h == 0 lanes
    pre = nxt;
    loader_advance(ld);
  }
  if (SIGMA_ONLY) {
    if (h == 0 && p < P) out[p] = sigma;
    if (amax >= 65520.0f) flags |= NSR_FLAG_ACTIVATION_RANGE;
    if (!nsr_finite(sigma)) flags |= NSR_FLAG_OUTPUT_NONFINITE;
    if (p < P) nsr_raise(tail, flags);
    l1pre = pre;
    continue;
  }

  NSR_TL(4);
  // ---- dir_encoding (cat([g, de]) -> 128, relu) fused with the rgb head (128 -> 3, sigmoid)
  float rgb[3] = {0.0f, 0.0f, 0.0f};
  float2 w2[3];          // colour-head weights of the pair consumed in the next k-step (rgb_gap)
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    Acc cur;
    cur.m = pre.bias;
    u32x4 dh0 = {0u, 0u, 0u, 0u}, dh1 = {0u, 0u, 0u, 0u};   // TRAIN: hi halves of the pending dir block (relu_hi_pair)
    const unsigned next_bias = 36u * 1024u;
    Pre nxt;
    // in flight behind the previous block's DMA: the density block's 2 stores | nothing | a dir block's 2 + sign word
    const unsigned a_seq = ld.slot_cur + ld.lane_off;
    const ChunkRef c2 = nb < 2 ? dir_ref(nb + 2, wave) : first_ref(nb - 2, wave);   // ... then the next tile's L1
    auto b_of = [&](int s, int part) -> u32x4 {
      return (s < 16) ? (part ? bl[s & 15] : bh[s & 15]) : (part ? del[s & 1] : deh[s & 1]);
    };
    auto hook = [&](int s, int g) {
      if (OVERLAP && nb == 0 && g == 0) enc_slot(144 + s, enc, ein);
      // the pending dir block is consumed in k-steps 6..13, one pair per k-step, one colour channel per gap
      if (nb > 0) rgb_gap<kConvStep0>(s, g, pend, aux + hx::kAuxRgbW + 32 * (nb - 1), h, rgb, w2);
      if (TRAIN && nb > 0 && s < 8 && g == 1) relu_hi_step(s, pend, dh0, dh1);
      if (TRAIN && nb > 0 && s >= 8 && s < 16 && g == 2) {
        const char* blk = panel_block(tr, 9, nb - 1);
        if (s == 14) unit_store<0>(dh0, blk, voff0);
        if (s == 15) unit_store<1>(dh1, blk, voff1);
        sign_push(sbits, pend.m[2 * (s - 8)]);
        sign_push(sbits, pend.m[2 * (s - 8) + 1]);
        if (s == 15 && ((nb - 1) & 1)) sign_store(sbits, sign_block(tr.sgn, tr.group, 9, nb - 1), ld.lane_off >> 2);
      }
    };
    auto next = [&](int k, int g) {   // the last block hands over to the next tile's L1 (bias behind 32 weight pieces)
      if (nb < 3 || PERSIST) prefetch_next_chunk(nxt, k, g, ld, nb < 3 ? next_bias : 32u * 1024u, h);
    };
    // one-tile workgroups fetch nothing behind the last two blocks (the persistent build fetches the next tile's L1 there)
    constexpr bool kFetch = PERSIST;
    if (nb >= 2 && !kFetch) {
      if (!TRAIN) block_mma3<18, kBar, 0, false>(cur, pre, a_seq, ld, c2, b_of, hook, next);
      else block_mma3<18, kBar, kTrainYoung, false>(cur, pre, a_seq, ld, c2, b_of, hook, next);
    } else if (!TRAIN || nb == 1) block_mma3<18, kBar>(cur, pre, a_seq, ld, c2, b_of, hook, next);
    else if (nb == 0) block_mma3<18, kBar, 2>(cur, pre, a_seq, ld, c2, b_of, hook, next);
    else block_mma3<18, kBar, kTrainYoung>(cur, pre, a_seq, ld, c2, b_of, hook, next);
    pend = cur;
    pre = nxt;
    loader_advance(ld);
  }
  NSR_TL(5);
  {
    const float* w32 = aux + hx::kAuxRgbW + 32 * 3;
#pragma unroll
    for (int g = 0; g < 3; ++g) w2[g] = pair_rgb_w<0>(w32, h, g);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int g = 0; g < 3; ++g) rgb_gap<0>(s, g, pend, w32, h, rgb, w2);
  }
  if (TRAIN) {   // the last dir block (its accumulators were read by the colour head just above: no MFMA is in flight)
    const char* blk = panel_block(tr, 9, 3);
    u32x4 dh0, dh1;
#pragma unroll
    for (int P = 0; P < 8; ++P) relu_hi_step(P, pend, dh0, dh1);
#pragma unroll
    for (int r = 0; r < 16; ++r) sign_push(sbits, pend.m[r]);
    unit_store<0>(dh0, blk, voff0);
    unit_store<1>(dh1, blk, voff1);
    sign_store(sbits, sign_block(tr.sgn, tr.group, 9, 3), ld.lane_off >> 2);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s = rgb[k];
    s += __shfl_xor(s, 32, 64);
    s += aux[hx::kAuxRgbB + k];
    rgb[k] = nsr_colour_activation(s, opts);
  }
  if (opts & kOptGamma) {
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = nsr_gamma(rgb[k]);
  }
  if (amax >= 65520.0f) flags |= NSR_FLAG_ACTIVATION_RANGE;
  if (!(nsr_finite(rgb[0]) && nsr_finite(rgb[1]) && nsr_finite(rgb[2]) && nsr_finite(sigma))) flags |= NSR_FLAG_OUTPUT_NONFINITE;
  if (p < P) nsr_raise(tail, flags);
  if (out && h == 0 && p < P) reinterpret_cast<float4*>(out)[p] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
  NSR_TL(6);
  // the compositor's staging area is its own: at least L1's two publish barriers lie between two uses, and the weight
  // ring keeps streaming underneath
  if (COMP) composite_tile<(COMP ? NS : 64), true>(ring + kComp0, h == 0, wave, m, lane, make_float4(rgb[0], rgb[1], rgb[2], sigma),
                                                  zv[pc], P / NS, co, tile);
#ifdef NSR_ABL_TIMELINE
  NSR_TL(7);
  if (lane == 0 && tile < kTlGroups) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    unsigned long long* dst = nsr_tl + ((size_t)tile * 4 + wave) * kTlSlots;
    for (int k = 0; k < 8; ++k) dst[k] = tl[k];
    dst[8] = hwid;
    dst[9] = xcc;
    for (int k = 0; k < 8; ++k) nsr_tk[((size_t)tile * 4 + wave) * 8 + k] = tk[k];
  }
#endif
  // hand-over to the next tile: the head of its first L1 block, and the encoding made in this tile's shadow
  l1pre = pre;
  if (OVERLAP) {
    deh[0] = u32x4{enc.dh0, enc.dh1, enc.dh2, enc.dh3};
    deh[1] = u32x4{enc.dh4, enc.dh5, enc.dh6, enc.dh7};
    del[0] = u32x4{enc.dl0, enc.dl1, enc.dl2, enc.dl3};
    del[1] = u32x4{enc.dl4, enc.dl5, enc.dl6, enc.dl7};
    flags = enc.flags;
  }
  } while (PERSIST && (tile += gridDim.x) < n_tiles);
  dma_drain();     // no LDS-DMA may be in flight when the workgroup's LDS is released (the last tile fetched chunks 0, 1 again)
}

// one workgroup per CU for the persistent ray kernels (they are LDS- and register-bound to that anyway)
static int nsr_cu_count() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
    n = 256;
  return n;
}
static dim3 persistent_grid(int64_t P) {
  const int64_t n_tiles = (P + 127) / 128;
#ifdef NSR_PERSISTENT
  const int64_t cus = nsr_cu_count();
  return dim3((unsigned)(n_tiles < cus ? n_tiles : cus));
#else
  return dim3((unsigned)n_tiles);
#endif
}

template <int MODE, bool SIGMA_ONLY>
static int launch_f16x3(const void* packed, const float* x, const float* z, int64_t P, int N, int stride, float* out,
                        unsigned* tail_w, hipStream_t st) {
  const NsrTail tail{tail_w};
  // the ray kernels loop over tiles, one workgroup per CU; the embedded-row kernels keep one tile per workgroup
  const dim3 grid = (MODE == 1 && !SIGMA_ONLY) ? persistent_grid(P) : dim3((unsigned)((P + 127) / 128)), block(256);
  const float* pk = static_cast<const float*>(packed);
  if (MODE == 1 && N == 64)
    hipLaunchKernelGGL((mlp_f16x3_kernel<MODE, SIGMA_ONLY, 64>), grid, block, 0, st, pk, x, z, P, N, stride, out, tail);
  else if (MODE == 1 && N == 128)
    hipLaunchKernelGGL((mlp_f16x3_kernel<MODE, SIGMA_ONLY, 128>), grid, block, 0, st, pk, x, z, P, N, stride, out, tail);
  else
    hipLaunchKernelGGL((mlp_f16x3_kernel<MODE, SIGMA_ONLY, 0>), grid, block, 0, st, pk, x, z, P, N, stride, out, tail);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

extern "C" NSR_INTERNAL int nsr_f16x3_mlp_forward(const void* packed, const float* x, int64_t P, int sigma_only, float* out,
                                     unsigned* tail, void* stream) {
  return sigma_only ? launch_f16x3<0, true>(packed, x, nullptr, P, 1, 8, out, tail, nsr_stream(stream))
                    : launch_f16x3<0, false>(packed, x, nullptr, P, 1, 8, out, tail, nsr_stream(stream));
}

// render_rays + VolumetricRenderer.forward in one launch (n_samples 64 or 128); raw (R * N, 4) optional
extern "C" NSR_INTERNAL int nsr_f16x3_render_composite(const void* packed, const float* rays, int ray_stride, const float* z,
                                                       int64_t R, int N, float* raw, const NsrCompOut* co, unsigned* tail_w,
                                                       void* stream) {
  const NsrTail tail{tail_w};
  const int64_t P = R * N;
  const dim3 grid = persistent_grid(P), block(256);
  const float* pk = static_cast<const float*>(packed);
  if (N == 64)
    hipLaunchKernelGGL((mlp_f16x3_kernel<1, false, 64, true>), grid, block, 0, nsr_stream(stream), pk, rays, z, P, N, ray_stride, raw, tail, *co);
  else if (N == 128)
    hipLaunchKernelGGL((mlp_f16x3_kernel<1, false, 128, true>), grid, block, 0, nsr_stream(stream), pk, rays, z, P, N, ray_stride, raw, tail, *co);
  else
    return NSR_ERR_UNSUPPORTED;
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

// forward pass of the training step: raw (R * N, 4) network outputs + the training panels (see nsr_f16x3_core.h);
// `pan` holds the twelve fp16 panels of ceil(R N / 128) * 4 point groups (nsr_f16x3_train_panel_bytes)
extern "C" NSR_INTERNAL int64_t nsr_f16x3_train_panel_bytes(int64_t P) {
  const int64_t n_groups = ((P + 127) / 128) * 4;
  return panel_set_bytes(n_groups);
}
extern "C" NSR_INTERNAL int64_t nsr_f16x3_train_sign_words(int64_t P) { return sign_panel_words(((P + 127) / 128) * 4); }
extern "C" NSR_INTERNAL int nsr_f16x3_train_forward(const void* packed, const float* rays, int ray_stride, const float* z, int64_t R,
                                                    int N, float* raw, void* pan, unsigned* sgn, unsigned* status, void* stream) {
  const int64_t P = R * N;
  const dim3 grid((unsigned)((P + 127) / 128)), block(256);
  hipLaunchKernelGGL((mlp_f16x3_kernel<1, false, 0, false, true>), grid, block, 0, nsr_stream(stream),
                     static_cast<const float*>(packed), rays, z, P, N, ray_stride, raw, NsrTail{status}, NsrCompOut{}, static_cast<float*>(pan), sgn);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

extern "C" NSR_INTERNAL int nsr_f16x3_render_rays(const void* packed, const float* rays, int ray_stride, const float* z, int64_t R,
                                     int N, float* out, unsigned* tail, void* stream) {
  return launch_f16x3<1, false>(packed, rays, z, R * N, N, ray_stride, out, tail, nsr_stream(stream));
}
