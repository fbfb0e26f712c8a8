// M1 / D2 on the fp16 matrix pipe with split operands ("f16x3", NSR_F16X3), block-PAIR schedule.
// EXPERIMENT, not the product: built and dispatched only with -DNSR_F16X3_PAIR (python -m nerf_sr_amd.build --variant
// pair -DNSR_F16X3_PAIR).  Measured on MI355X (profiles/r2_f16x3_pair_experiment.md): same results as the product
// kernel (all parity tests pass), fine pass 66.7 ms against 60.7-62 ms: the premise below -- that back-to-back MFMAs
// on ONE accumulator stall -- is wrong for this part (a bare loop issues them at the full rate), so the pair schedule
// only adds 2 % of bias MFMAs and a clumpier DMA issue pattern.
//
// Arithmetic: every fp32 value v is carried as hi = RNE_f16(v), lo = RNE_f16(v - hi); a product is
// a_hi*b_hi + (a_hi*b_lo + a_lo*b_hi) on v_mfma_f32_32x32x16_f16 with fp32 accumulation.  Two details keep
// the split at its full 22 bits:
//   * the weights (and biases) are multiplied by 2^kScaleLog2 before they are split (exact), so that the lo
//     part of a typical weight (|w| ~ 0.05: lo ~ 2^-16) stays clear of fp16's subnormal floor (2^-24), which
//     otherwise costs three bits; accumulators therefore hold 2^kScaleLog2 x the layer output and the factor
//     is removed, exactly, inside the conversion of a finished block (v_fma_mix: f16(x * 2^-k));
//   * hi is rounded to nearest (the same v_fma_mix), so |lo| <= 2^-12 |v| instead of 2^-11 |v|.
//
// Schedule: same register algebra as the fp32 kernel (nsr_mlp_layout.h): a wave owns 32 sample points,
// activations never leave registers (the D fragment of an output block IS the B operand of two k-steps of the
// next layer), weights stream global -> LDS by DMA through a 3-slot ring.  A chunk of the stream carries TWO
// 32-feature output blocks over HALF of the layer's k-steps: consecutive MFMAs alternate between the two
// blocks' accumulators, so no MFMA waits for the result of the one issued before it (back-to-back MFMAs on
// ONE accumulator stall the pipe as soon as any other instruction sits between them: the round-1 kernel,
// three dependent MFMAs per k-step, kept the matrix pipe 66 % busy), and the VALU re-split of a finished pair
// rides between the MFMAs of the following pair.  The bias enters through the matrix pipe (one fragment
// holding (hi, lo) of both blocks' biases against a constant ones-operand).
#include "../nsr_common.h"
#include "../nsr_mlp_layout.h"
#include "../nsr_mlp_stream.h"
#include "../nsr_mlp_encode.h"

using namespace nsr;
using namespace nsr::stream;

namespace fp {

constexpr int kScaleLog2 = 6;                       // |w| < 1023 stays inside fp16; lo floor 2^-31 absolute
constexpr float kScale = 64.0f, kInvScale = 1.0f / 64.0f;

// ---- stream layout ----------------------------------------------------------------------------------------
// k-step piece group: [ A_hi(b0), A_lo(b0), A_hi(b1), A_lo(b1) ]   (b0, b1 = the chunk's two output blocks)
// chunk  = k-step groups of its half of the layer, then (first halves only) ONE bias fragment
//   0, 1     : L1, two block pairs of 4 k-steps each per chunk + one bias fragment per pair (34 pieces)
//   then     : trunk layer L = 1..8 (L2..L8, xyz_encoding_final), pair pb = 0..3, half hf = 0, 1
//              (8 + 8 k-steps: 33 / 32 pieces; L == 4, the skip layer: each half is led by two of the four k-steps over the
//              encoded position, 10 + 10: 41 / 40)
//   then     : density head: ONE row block whose k-steps 0..7 / 8..15 play the two "blocks" (33 pieces)
//   then     : dir_encoding, pair pb = 0, 1, halves of 9 k-steps (37 / 36)
constexpr int kChunks = 2 + 8 * 8 + 1 + 4;          // 71
constexpr int kSlotPieces = 41;
constexpr int kSlotFloats = kSlotPieces * 256;
constexpr int kSlotBytes = kSlotFloats * 4;
constexpr int kL1ChunkPieces = 34;
constexpr int kSigmaPiece0 = 68 + 7 * 260 + 324;   // 2212
constexpr int kDirPiece0 = kSigmaPiece0 + 33;      // 2245
constexpr int kPiecesTotal = kDirPiece0 + 2 * 73;  // 2391
constexpr int kAuxRgbW = 0, kAuxRgbB = 384, kAuxFloats = 448;
constexpr int kPF = 2;                              // fragment prefetch depth in k-steps (one k-step = 6 MFMAs)
constexpr int kBar = 3;                             // publish point of the next chunk (k-step)

NSR_HD int trunk_steps(int L) { return L == 4 ? 10 : 8; }                   // k-steps per half
NSR_HD int trunk_pair_pieces(int L) { return 8 * trunk_steps(L) + 1; }      // both halves + the bias fragment
NSR_HD int trunk_base(int L) { return L <= 3 ? 68 + 260 * (L - 1) : (L == 4 ? 848 : 1172 + 260 * (L - 5)); }

struct Chunk {
  int tensor;    // weight tensor (state_dict index); bias = tensor + 1
  int nb0;       // first output block of the chunk's first pair
  int npairs;    // 2 for L1, else 1
  int k0, steps; // first k-step of the layer this chunk covers, number of k-steps
  int nbias;     // bias fragments at the end of the chunk
  int piece0, pieces;
  bool sigma;
};
NSR_HD Chunk chunk_info(int q) {
  Chunk c{};
  if (q < 2) {
    c.tensor = 0; c.nb0 = 4 * q; c.npairs = 2; c.k0 = 0; c.steps = 4; c.nbias = 2;
    c.piece0 = kL1ChunkPieces * q; c.pieces = kL1ChunkPieces;
  } else if (q < 66) {
    const int L = 1 + (q - 2) / 8, r = (q - 2) % 8, pb = r >> 1, hf = r & 1;
    c.tensor = 2 * L; c.nb0 = 2 * pb; c.npairs = 1; c.steps = trunk_steps(L); c.k0 = hf * c.steps;
    c.nbias = hf ? 0 : 1;
    c.pieces = 4 * c.steps + c.nbias;
    c.piece0 = trunk_base(L) + trunk_pair_pieces(L) * pb + (hf ? 4 * c.steps + 1 : 0);
  } else if (q == 66) {
    c.tensor = 20; c.nb0 = 0; c.npairs = 1; c.k0 = 0; c.steps = 8; c.nbias = 1; c.piece0 = kSigmaPiece0; c.pieces = 33;
    c.sigma = true;
  } else {
    const int r = q - 67, pb = r >> 1, hf = r & 1;
    c.tensor = 18; c.nb0 = 2 * pb; c.npairs = 1; c.steps = 9; c.k0 = 9 * hf; c.nbias = hf ? 0 : 1;
    c.pieces = 36 + c.nbias;
    c.piece0 = kDirPiece0 + 73 * pb + (hf ? 37 : 0);
  }
  return c;
}

__device__ __forceinline__ ChunkRef mkref(int piece0, int pieces, int wave) { return make_ref<4>(piece0, pieces, wave); }
// chunk c (0..7 = 2 * pb + hf) of trunk layer L
__device__ __forceinline__ ChunkRef layer_ref(int L, int c, int wave) {
  const int st = trunk_steps(L);
  return mkref(trunk_base(L) + trunk_pair_pieces(L) * (c >> 1) + ((c & 1) ? 4 * st + 1 : 0), 4 * st + ((c & 1) ? 0 : 1), wave);
}
__device__ __forceinline__ ChunkRef sigma_ref(int wave) { return mkref(kSigmaPiece0, 33, wave); }
__device__ __forceinline__ ChunkRef dir_ref(int c, int wave) {   // c = 2 * pb + hf
  return mkref(kDirPiece0 + 73 * (c >> 1) + ((c & 1) ? 37 : 0), (c & 1) ? 36 : 37, wave);
}
// past the end of the sequence chunk 0 is re-fetched into the idle slot (the first DMA issues of every chunk stay
// branch-free); the kernel drains before exit
__device__ __forceinline__ ChunkRef end_ref(int wave) { return mkref(0, 32, wave); }
__device__ __forceinline__ void issue(const Loader& ld, int i) { loader_issue<8>(ld, i); }

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
// (a, b) -> packed fp16 pair of their hi parts (part 0) or lo parts (part 1); round to nearest even
__device__ __forceinline__ unsigned pack_hl(float a, float b, int part) {
  _Float16 ha = (_Float16)a, hb = (_Float16)b;
  if (part) {
    ha = (_Float16)(a - (float)ha);
    hb = (_Float16)(b - (float)hb);
  }
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}

// one thread per 32-bit word of the blob
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
    const int lane = word >> 2, jj = word & 3;
    if (local >= c.pieces - c.nbias) {
      // bias fragment: lane (row i, half 0), word jj < 2 = (hi, lo) of 2^k * bias of row i of block 2 * pair + jj
      const int pair = local - (c.pieces - c.nbias);
      if (lane < 32 && jj < 2) {
        float bv;
        if (c.sigma) bv = (lane == 0 && jj == 0) ? w.p[21][0] : 0.0f;      // one real row; the K-split partner adds 0
        else bv = w.p[c.tensor + 1][32 * (c.nb0 + 2 * pair + jj) + lane];
        bv *= kScale;
        const _Float16 hi = (_Float16)bv;
        const _Float16 lo = (_Float16)(bv - (float)hi);
        v = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
      }
    } else {
      const int per_pair = 4 * c.steps;
      const int pair = local / per_pair, rem = local % per_pair;
      const int s = rem >> 2, blk = (rem >> 1) & 1, part = rem & 1;
      // density head: "block" blk is the K half (k-steps 8 * blk ..) of the single row block
      // k-step of the layer (hx::column_of numbering) behind position s of this chunk
      int ks = c.k0 + s;
      if (c.sigma) ks = s + 8 * blk;
      else if (c.tensor == 8) ks = (s < 2) ? 2 * (c.k0 / 10) + s : 4 + 8 * (c.k0 / 10) + (s - 2);
      const int nb = c.sigma ? 0 : c.nb0 + 2 * pair + blk;
      const int n = 32 * nb + (lane & 31), h = lane >> 5;
      float f[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = hx::column_of(c.tensor, ks, 2 * jj + e, h);
        const bool real_row = !c.sigma || n == 0;
        f[e] = (col == kPad || !real_row) ? 0.0f : kScale * w.p[c.tensor][n * tensor_ld(c.tensor) + col];
      }
      v = pack_hl(f[0], f[1], part);
    }
  } else {
    const int a = idx - stream_words;
    float f = 0.0f;
    if (a < kAuxRgbB) f = kInvScale * w.p[22][a];      // the colour head reads 2^k-scaled dir_encoding outputs
    else if (a < kAuxRgbB + 3) f = w.p[23][a - kAuxRgbB];
    v = __float_as_uint(f);
  }
  out[idx] = v;
}

// ---- kernel ----------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

struct Acc2 {
  f32x16 a0, a1;   // the chunk's two output blocks (2^kScaleLog2 x bias + sum)
};
// head of a k-step sequence: the fragment groups of its first kPF k-steps (+ the chunk's bias fragment)
struct Pre2 {
  u32x4 f[kPF][4];
  u32x4 bias;
};
__device__ __forceinline__ void prefetch_group(Pre2& pre, int k, unsigned seq_addr) {
  const u32x4* a = lds_vec(seq_addr);
#pragma unroll
  for (int i = 0; i < 4; ++i) pre.f[k][i] = a[(4 * k + i) * 64];
}
// constant B operands of the bias MFMA: ones in k-slots (0, 1) resp. (2, 3) of the lanes of half 0
__device__ __forceinline__ void bias_operands(int h, u32x4& b0, u32x4& b1) {
  const unsigned ones = h == 0 ? 0x3c003c00u : 0u;
  b0 = u32x4{ones, 0u, 0u, 0u};
  b1 = u32x4{0u, ones, 0u, 0u};
}
__device__ __forceinline__ void init_acc(Acc2& acc, const u32x4& bias_frag, const u32x4& b0, const u32x4& b1) {
  const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  acc.a1 = mma(bias_frag, b1, zero);
  acc.a0 = mma(bias_frag, b0, zero);
}

// NSTEP k-steps of a block pair: 6 MFMAs per k-step alternating between the two accumulators, fragment groups
// software-pipelined kPF k-steps ahead (the first kPF come in through `pre`), order pinned by sched_barrier.
// a_addr: LDS byte address (+ lane*16) of the sequence's first piece.  b_of(s, part, blk) yields the k-step's
// activation operand for block blk (the same for both blocks except in the K-split density head); hook(s) runs
// beside the MFMAs of k-step s; BAR >= 0 places the next chunk's publish point + the DMA issue of the chunk after
// it; next(k), k = 0..kPF-1, runs in the last kPF k-steps and prefetches the head of the following sequence.
template <int NSTEP, int BAR, class BOf, class Hook, class Next>
__device__ __forceinline__ void pair_mma(Acc2& acc, const Pre2& pre, unsigned a_addr, Loader& ld, const ChunkRef& c2,
                                         BOf&& b_of, Hook&& hook, Next&& next) {
  static_assert(NSTEP >= kPF, "sequence shorter than the prefetch depth");
  const u32x4* a_pieces = lds_vec(a_addr);
  u32x4 f[NSTEP][4];
#pragma unroll
  for (int s = 0; s < kPF; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) f[s][i] = pre.f[s][i];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    if (s == BAR) loader_publish(ld, c2);
    if (s + kPF < NSTEP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) f[s + kPF][i] = a_pieces[(4 * (s + kPF) + i) * 64];
    } else {
      next(s + kPF - NSTEP);
    }
    const u32x4 bh0 = b_of(s, 0, 0), bl0 = b_of(s, 1, 0), bh1 = b_of(s, 0, 1), bl1 = b_of(s, 1, 1);
    // the youngest fragment load first: one lgkmcnt wait serves the whole k-step
    acc.a1 = mma(f[s][3], bh1, acc.a1);     // a_lo * b_hi
    acc.a0 = mma(f[s][1], bh0, acc.a0);
    acc.a1 = mma(f[s][2], bl1, acc.a1);     // a_hi * b_lo
    acc.a0 = mma(f[s][0], bl0, acc.a0);
    acc.a1 = mma(f[s][2], bh1, acc.a1);     // a_hi * b_hi
    acc.a0 = mma(f[s][0], bh0, acc.a0);
    hook(s);
    if (BAR >= 0 && s >= BAR && s < BAR + 4) {   // DMA of chunk j+2: three pieces per k-step over four k-steps
      issue(ld, 3 * (s - BAR));
      issue(ld, 3 * (s - BAR) + 1);
      issue(ld, 3 * (s - BAR) + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- re-split of a finished block: accumulator pair P (registers 2P, 2P+1) -> activation -> (hi, lo) fp16 pairs
// in the operand registers of the consuming layer: block nb becomes k-steps 2nb (P < 4 -> h0 / l0) and 2nb + 1
// (P >= 4 -> h1 / l1), element P & 3.  Two halves (A = activation + hi, B = lo) so that the work spreads thinly
// over the k-steps of the following pair.  asm volatile pins each half into its k-step (LLVM would otherwise
// sink all of it to the first use, i.e. serialise it at the layer end).
struct PairTmp {
  float x0, x1;
  unsigned hi;
};
template <int P>
__device__ __forceinline__ void pair_half_a(const f32x16& m, float lower, float ks, PairTmp& t, u32x4& h0, u32x4& h1) {
  // activation = max(x, lower) (raw v_max: fmaxf() would add a canonicalising v_max per operand), then
  // hi = RNE_f16(x * 2^-k) for both values of the pair
  asm volatile("v_max_f32 %0, %2, %4\n\tv_max_f32 %1, %3, %4"
               : "=&v"(t.x0), "=&v"(t.x1)
               : "v"(m[2 * P]), "v"(m[2 * P + 1]), "v"(lower));
  asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
               : "=&v"(t.hi)
               : "v"(t.x0), "v"(t.x1), "v"(ks));
  if (P < 4) h0[P & 3] = t.hi; else h1[P & 3] = t.hi;
}
template <int P>
__device__ __forceinline__ void pair_half_b(const PairTmp& t, float ks, u32x4& l0, u32x4& l1) {
  // lo = RNE_f16(x * 2^-k - hi): the fma result is exact in fp32 (hi is within 2^-12 of x * 2^-k)
  unsigned lo;
  asm volatile(
      "v_fma_mixlo_f16 %0, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %3, %4, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo)
      : "v"(t.hi), "v"(t.x0), "v"(t.x1), "v"(ks));
  if (P < 4) l0[P & 3] = lo; else l1[P & 3] = lo;
}
template <int HP>   // half-step HP (0..15) of ONE block
__device__ __forceinline__ void block_half(const f32x16& m, float lower, float ks, PairTmp& t, u32x4& h0, u32x4& l0,
                                           u32x4& h1, u32x4& l1) {
  if (HP & 1) pair_half_b<(HP >> 1)>(t, ks, l0, l1);
  else pair_half_a<(HP >> 1)>(m, lower, ks, t, h0, h1);
}
// half-step hp (0..31) of a finished PAIR: block 0 -> operand k-steps (h0, l0), (h1, l1); block 1 -> (h2, l2), (h3, l3)
struct Dst4 {
  u32x4 &h0, &l0, &h1, &l1, &h2, &l2, &h3, &l3;
};
__device__ __forceinline__ void pending_half(int hp, const Acc2& p, float lower, float ks, PairTmp& t, const Dst4& d) {
#define NSR_HP(N)                                                                  \
  case N: block_half<N>(p.a0, lower, ks, t, d.h0, d.l0, d.h1, d.l1); break;        \
  case 16 + N: block_half<N>(p.a1, lower, ks, t, d.h2, d.l2, d.h3, d.l3); break;
  switch (hp) {
    NSR_HP(0) NSR_HP(1) NSR_HP(2) NSR_HP(3) NSR_HP(4) NSR_HP(5) NSR_HP(6) NSR_HP(7)
    NSR_HP(8) NSR_HP(9) NSR_HP(10) NSR_HP(11) NSR_HP(12) NSR_HP(13) NSR_HP(14) NSR_HP(15)
    default: break;
  }
#undef NSR_HP
}
// the 32 half-steps of a pending pair over the k-steps gs = 0.. of the following pair, PER per k-step: with PER = 3
// the pair is complete after k-step 10, i.e. before operand k-steps 12..15 (the last pair of the previous layer)
// are first read
template <int PER>
__device__ __forceinline__ void pending_step(int gs, const Acc2& p, float lower, float ks, PairTmp& t, const Dst4& d) {
#pragma unroll
  for (int i = 0; i < PER; ++i) pending_half(PER * gs + i, p, lower, ks, t, d);
}
// operand k-steps k .. k+3 of a register set as the destination of a pending pair
#define NSR_DST4(H, L, K) Dst4{H[(K)], L[(K)], H[(K) + 1], L[(K) + 1], H[(K) + 2], L[(K) + 2], H[(K) + 3], L[(K) + 3]}

// colour head: register pair Q (0..15; block Q >> 3) of a finished dir_encoding pair (relu) dotted with the three
// rgb rows (pre-multiplied by 2^-k at pack time)
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

// head of the NEXT chunk (its first fragment groups, and its bias fragment if it has one), prefetched in the last
// kPF k-steps of the current one
__device__ __forceinline__ void prefetch_next_chunk(Pre2& nxt, int k, const Loader& ld, int bias_piece) {
  prefetch_group(nxt, k, ld.slot_next + ld.lane_off);
  if (k == 0 && bias_piece >= 0) nxt.bias = lds_vec(ld.slot_next + (unsigned)bias_piece * 1024u + ld.lane_off)[0];
}

// v -> (hi, lo) fp16 pairs, round to nearest (prologue only)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
  const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
  hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}

struct LayerCtx {
  int L;
  float lower, ks;
  const u32x4* stash;
  const u32x4 &bc0, &bc1;
  const ChunkRef &after0, &after1;
};
// chunk C (= 2 * pair + half, compile time) of a trunk layer; `cur` = the pair's accumulators
template <int C>
__device__ __forceinline__ void trunk_chunk(const LayerCtx& cx, u32x4 (&bh)[16], u32x4 (&bl)[16], u32x4 (&oh)[16],
                                            u32x4 (&ol)[16], Loader& ld, Acc2& cur, PairTmp& ptmp, const Acc2& pend, Pre2& pre) {
  constexpr int pb = C >> 1, hf = C & 1;
  const int L = cx.L;
  // chunks j+1 (prefetched from at the end of this one) and j+2 (DMA'd during this one)
  const ChunkRef c2 = (C + 2 < 8) ? layer_ref(L, (C + 2) & 7, ld.wave) : (C + 2 == 8 ? cx.after0 : cx.after1);
  const int next_pieces = (C + 1 < 8) ? (4 * trunk_steps(L) + (hf ? 1 : 0)) : cx.after0.pieces;
  const int next_bias = (hf == 1) ? next_pieces - 1 : -1;     // first halves carry the pair's bias fragment
  const unsigned a_addr = ld.slot_cur + ld.lane_off;
  Pre2 nxt;
  unsigned act_addr = a_addr;
  if (L == 4) {
    // skip layer: two of the four k-steps that contract over the encoded position (parked in LDS by the
    // prologue) lead each half; they hand the first fragment groups of the activation part over through `mid`
    u32x4 pe4[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pe4[i] = cx.stash[(2 * hf + i) * 64];
      pe4[2 + i] = cx.stash[(4 + 2 * hf + i) * 64];
    }
    Pre2 mid;
    pair_mma<2, -1>(
        cur, pre, a_addr, ld, c2, [&](int s, int part, int) -> u32x4 { return pe4[2 * part + s]; }, [&](int) {},
        [&](int k) { prefetch_group(mid, k, a_addr + 8 * 1024); });
#pragma unroll
    for (int k = 0; k < kPF; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) pre.f[k][i] = mid.f[k][i];
    act_addr += 8 * 1024;
  }
  pair_mma<8, kBar>(
      cur, pre, act_addr, ld, c2, [&](int s, int part, int) -> u32x4 { return part ? bl[8 * hf + s] : bh[8 * hf + s]; },
      [&](int s) {
        const int gs = 8 * hf + s;
        if (pb == 0)   // blocks 6, 7 of the previous layer (always relu'd: L1..L7) -> k-steps 12..15 of THIS layer's input
          pending_step<3>(gs, pend, 0.0f, cx.ks, ptmp, NSR_DST4(bh, bl, 12));
        else
          pending_step<3>(gs, pend, cx.lower, cx.ks, ptmp, NSR_DST4(oh, ol, (4 * pb - 4) & 15));
      },
      [&](int k) { prefetch_next_chunk(nxt, k, ld, next_bias); });
  pre = nxt;
  loader_advance(ld);
}

// One 256 -> 256 trunk layer L (1..8; 8 = xyz_encoding_final): in (bh, bl) -> out (oh, ol), four block pairs of
// two half-K chunks each.  `pend` is the pair that finished last (blocks 6, 7 of the previous layer on entry, of
// this layer on exit); it is activated and re-split beside the MFMAs of the FOLLOWING pair.  `pre` carries the
// prefetched head of the next chunk across chunk (and layer) boundaries.
__device__ __forceinline__ void trunk_layer(int L, u32x4 (&bh)[16], u32x4 (&bl)[16], u32x4 (&oh)[16], u32x4 (&ol)[16],
                                            const u32x4* stash, Loader& ld, const u32x4& bc0, const u32x4& bc1, float ks,
                                            Acc2& pend, Pre2& pre, const ChunkRef& after0, const ChunkRef& after1) {
  // relu on L2..L8, none on xyz_encoding_final
  const LayerCtx cx{L, (L < 8) ? 0.0f : -__builtin_inff(), ks, stash, bc0, bc1, after0, after1};
  PairTmp ptmp;
#define NSR_PAIR(PB)                                                      \
  {                                                                       \
    Acc2 cur;                                                             \
    init_acc(cur, pre.bias, bc0, bc1);                                    \
    trunk_chunk<2 * PB>(cx, bh, bl, oh, ol, ld, cur, ptmp, pend, pre);     \
    trunk_chunk<2 * PB + 1>(cx, bh, bl, oh, ol, ld, cur, ptmp, pend, pre); \
    pend = cur;                                                           \
  }
  NSR_PAIR(0) NSR_PAIR(1) NSR_PAIR(2) NSR_PAIR(3)
#undef NSR_PAIR
}

template <int MODE, bool SIGMA_ONLY>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_f16x3_kernel(const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ zv,
                 int64_t P, int N, int stride, float* __restrict__ out) {
  // 3 x 41 KiB weight ring + per-wave stash of the encoded position (8 fragments x 64 lanes x 16 B = 8 KiB
  // per wave) + the colour-head block (rgb weights and bias, 448 floats): 160,512 B of the CU's 160 KiB
  constexpr int kStash0 = 3 * kSlotFloats, kAux0 = kStash0 + 4 * 8 * 256;
  __shared__ __attribute__((aligned(16))) float ring[kAux0 + kAuxFloats];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* aux = ring + kAux0;          // LDS copy, visible after the first barrier
  for (int i = threadIdx.x; i < kAuxFloats; i += 256) ring[kAux0 + i] = packed[kPiecesTotal * 256 + i];

  Loader ld;
  ld.stream = packed;
  ld.wave = wave;
  ld.lane_off = (unsigned)lane * 16u;
  ld.slot_cur = lds_addr(ring);
  ld.slot_next = ld.slot_cur + kSlotBytes;
  ld.slot_free = ld.slot_cur + 2 * kSlotBytes;
  // chunks 0 and 1 (L1) stream in behind the encoding prologue
  loader_prepare_dma(ld, mkref(0, kL1ChunkPieces, wave), ld.slot_cur);
#pragma unroll
  for (int i = 0; i < 9; ++i) issue(ld, i);
  loader_prepare_dma(ld, mkref(kL1ChunkPieces, kL1ChunkPieces, wave), ld.slot_next);
#pragma unroll
  for (int i = 0; i < 9; ++i) issue(ld, i);

  const int64_t p = (int64_t)blockIdx.x * 128 + wave * 32 + m;
  const int64_t pc = p < P ? p : P - 1;
  float pe[32], de[16];
  encode_point<MODE>(x, zv, pc, N, stride, h, pe, de);
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
  // park the split position encoding in LDS: the skip layer re-reads it, which frees 32 registers in the loop
  u32x4* stash = reinterpret_cast<u32x4*>(ring + kStash0) + wave * 8 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    stash[s * 64] = peh[s];
    stash[(4 + s) * 64] = pel[s];
  }

  u32x4 bh[16], bl[16], oh[16], ol[16];
  u32x4 bc0, bc1;
  bias_operands(h, bc0, bc1);
  const float ks = kInvScale;
  Acc2 pend;
  Pre2 pre;

  // ---- L1: two chunks of two block pairs, 4 k-steps each; pair g is re-split during pair g + 1.
  // Publish point at the chunk start (chunks 0 / 1 were issued above; chunk j+2 is fetched here).
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    loader_publish(ld, layer_ref(1, c, wave));    // chunk j+2 = first / second chunk of L2
    const unsigned a_chunk = ld.slot_cur + ld.lane_off;
    Pre2 nxt;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int pg = 2 * c + g;                   // pair number: blocks 2pg, 2pg + 1
      const unsigned a_addr = a_chunk + g * 16 * 1024;
      Pre2 mine;
#pragma unroll
      for (int k = 0; k < kPF; ++k) prefetch_group(mine, k, a_addr);
      mine.bias = lds_vec(ld.slot_cur + (32 + g) * 1024 + ld.lane_off)[0];
      Acc2 cur;
      init_acc(cur, mine.bias, bc0, bc1);
      PairTmp ptmp;
      pair_mma<4, -1>(
          cur, mine, a_addr, ld, end_ref(wave),
          [&](int s, int part, int) -> u32x4 { return part ? pel[s] : peh[s]; },
          [&](int s) {
            const int i = 4 * g + s;        // DMA of chunk j+2: two pieces per k-step over the chunk's 8 k-steps
            issue(ld, 2 * i);
            issue(ld, 2 * i + 1);
            if (pg > 0) pending_step<8>(s, pend, 0.0f, ks, ptmp, NSR_DST4(bh, bl, 4 * pg - 4));
          },
          [&](int k) {
            if (c == 1 && g == 1) prefetch_next_chunk(nxt, k, ld, 32);   // head of the first trunk chunk (bias = piece 32)
          });
      pend = cur;
    }
    if (c == 1) pre = nxt;
    loader_advance(ld);
  }

  // ---- L2..L8 (+ xyz_encoding_final), two layers per trip so the register sets swap roles
  constexpr int kPairs = SIGMA_ONLY ? 3 : 4;
#pragma unroll 1
  for (int pair = 0; pair < kPairs; ++pair) {
    const int L = 1 + 2 * pair;
    trunk_layer(L, bh, bl, oh, ol, stash, ld, bc0, bc1, ks, pend, pre, layer_ref(L + 1, 0, wave), layer_ref(L + 1, 1, wave));
    // what follows layer L+1: the next trunk layer, or (after xyz_encoding_final) the density head + dir_encoding
    const bool last = !SIGMA_ONLY && pair == kPairs - 1;
    const ChunkRef a0 = last ? sigma_ref(wave) : layer_ref(L + 2, 0, wave);
    const ChunkRef a1 = last ? dir_ref(0, wave) : layer_ref(L + 2, 1, wave);
    trunk_layer(L + 1, oh, ol, bh, bl, stash, ld, bc0, bc1, ks, pend, pre, a0, a1);
  }
  if (SIGMA_ONLY)   // xyz_encoding_final is not evaluated: L8 is followed by the density head, then nothing
    trunk_layer(7, bh, bl, oh, ol, stash, ld, bc0, bc1, ks, pend, pre, sigma_ref(wave), end_ref(wave));

  // ---- density head: sigma.weight as row 0 of one 32-row block over h8 (= oh/ol: the input of
  // xyz_encoding_final, still intact), its K split in two halves that play the two "blocks" of the chunk (two
  // independent accumulators, added at the end).  The pending pair is xyz_encoding_final's last one (-> bh[12..15],
  // no activation), or L8's last one in a sigma_only launch (-> oh[12..15], relu), which the second K half reads
  // from its k-step 4 on: six half-steps per k-step finish it in time.
  float sigma;
  {
    Acc2 cur;
    init_acc(cur, pre.bias, bc0, bc1);
    PairTmp ptmp;
    Pre2 nxt;
    pair_mma<8, kBar>(
        cur, pre, ld.slot_cur + ld.lane_off, ld, SIGMA_ONLY ? end_ref(wave) : dir_ref(1, wave),
        [&](int s, int part, int blk) -> u32x4 { return part ? ol[8 * blk + s] : oh[8 * blk + s]; },
        [&](int s) {
          if (SIGMA_ONLY) pending_step<6>(s, pend, 0.0f, ks, ptmp, NSR_DST4(oh, ol, 12));
          else pending_step<4>(s, pend, -__builtin_inff(), ks, ptmp, NSR_DST4(bh, bl, 12));
        },
        [&](int k) {
          if (!SIGMA_ONLY) prefetch_next_chunk(nxt, k, ld, 36);     // dir chunk 0: 36 weight pieces, then the bias
        });
    sigma = (cur.a0[0] + cur.a1[0]) * kInvScale;   // row 0 of the block lives in register 0 of the h == 0 lanes
    pre = nxt;
    loader_advance(ld);
  }
  if (SIGMA_ONLY) {
    if (h == 0 && p < P) out[p] = sigma;
    dma_drain();   // no LDS-DMA may be in flight when the workgroup's LDS is released
    return;
  }

  // ---- dir_encoding (cat([g, de]) -> 128, relu), two block pairs of two half chunks (9 k-steps each), fused with
  // the rgb head (128 -> 3, sigmoid)
  float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    Acc2 cur;
    init_acc(cur, pre.bias, bc0, bc1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int c = 2 * pb + hf;
      const ChunkRef c2 = (c + 2 < 4) ? dir_ref(c + 2, wave) : end_ref(wave);
      const int next_bias = (c == 1) ? 36 : -1;
      Pre2 nxt;
      pair_mma<9, kBar>(
          cur, pre, ld.slot_cur + ld.lane_off, ld, c2,
          [&](int s, int part, int) -> u32x4 {
            const int gs = 9 * hf + s;
            return (gs < 16) ? (part ? bl[gs & 15] : bh[gs & 15]) : (part ? del[gs & 1] : deh[gs & 1]);
          },
          [&](int s) {
            if (pb > 0 && 9 * hf + s < 16) rgb_q(9 * hf + s, pend, aux + kAuxRgbW, h, rgb);   // blocks 0, 1
          },
          [&](int k) {
            if (c < 3) prefetch_next_chunk(nxt, k, ld, next_bias);
          });
      pre = nxt;
      loader_advance(ld);
    }
    pend = cur;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) rgb_q(q, pend, aux + kAuxRgbW + 64, h, rgb);   // blocks 2, 3
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float s = rgb[k];
    s += __shfl_xor(s, 32, 64);
    s += aux[kAuxRgbB + k];
    rgb[k] = 1.0f / (1.0f + expf(-s));
  }
  if (h == 0 && p < P) reinterpret_cast<float4*>(out)[p] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
  dma_drain();     // no LDS-DMA may be in flight when the workgroup's LDS is released
}

template <int MODE, bool SIGMA_ONLY>
static int launch(const void* packed, const float* x, const float* z, int64_t P, int N, int stride, float* out,
                  hipStream_t st) {
  const dim3 grid((unsigned)((P + 127) / 128)), block(256);
  hipLaunchKernelGGL((mlp_f16x3_kernel<MODE, SIGMA_ONLY>), grid, block, 0, st, static_cast<const float*>(packed), x, z, P,
                     N, stride, out);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

}  // namespace fp

extern "C" NSR_INTERNAL size_t nsr_f16x3p_packed_bytes(void) { return 4 * (size_t)(fp::kPiecesTotal * 256 + fp::kAuxFloats); }

extern "C" NSR_INTERNAL int nsr_f16x3p_pack(const float* const* w, void* packed_dev, void* stream) {
  fp::PackPtrs pp;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w[i]) return NSR_ERR_INVALID_ARG;
    pp.p[i] = w[i];
  }
  const int total = fp::kPiecesTotal * 256 + fp::kAuxFloats;
  hipLaunchKernelGGL(fp::pack_kernel, dim3((total + 255) / 256), dim3(256), 0, nsr_stream(stream), pp,
                     static_cast<unsigned*>(packed_dev));
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" NSR_INTERNAL int nsr_f16x3p_mlp_forward(const void* packed, const float* x, int64_t P, int sigma_only, float* out,
                                                  void* stream) {
  return sigma_only ? fp::launch<0, true>(packed, x, nullptr, P, 1, 8, out, nsr_stream(stream))
                    : fp::launch<0, false>(packed, x, nullptr, P, 1, 8, out, nsr_stream(stream));
}

extern "C" NSR_INTERNAL int nsr_f16x3p_render_rays(const void* packed, const float* rays, int ray_stride, const float* z,
                                                  int64_t R, int N, float* out, void* stream) {
  return fp::launch<1, false>(packed, rays, z, R * N, N, ray_stride, out, nsr_stream(stream));
}
