// Backward CHAIN of the training step on the split-fp16 matrix pipe (the reference's counterpart is
// autograd through models/networks.py:130-213, driven by nerf_downX_model.py:398-408).
//
// The input gradients of the network -- d(rgb_pre), d(sigma) from the compositing backward, down to the first trunk
// layer -- are a per-point chain exactly like the forward pass: dz_{l-1} = (W_l^T dz_l) * [z_{l-1} > 0].  This kernel
// runs it with the machinery of the inference kernel (nsr_f16x3_core.h): W^T streams through the LDS ring as split-fp16
// A fragments, a wave owns 32 points, the gradient of a finished 32-feature block is masked, written out and re-split
// into the next layer's B operand in the MFMA shadow of the following block.  It replaces the ten input-gradient GEMMs
// of the step (each of which read and wrote a (P, 256) fp32 panel through a store-bound epilogue).
//
//   prologue (VALU)  dzc = (Wrgb^T d_rgb_pre) * [zcc > 0]                  128 features, K = 3
//   layer 0          dg  = Wdir[:, :256]^T dzc                             K = 128 (zero-padded to 16 k-steps)
//   layer 1          dz8 = (Wfinal^T dg + wsigma d_sigma) * [z8 > 0]       K = 256; the density head's rank-1 term is added
//                                                                          to the finished block on the VALU
//   layers 2..8      dz_{l-1} = (W_l^T dz_l) * [z_{l-1} > 0], l = 8..2     (l = 5: the h4 columns of the skip layer)
//
// Inputs: the sign panels of the forward pass (one bit per pre-activation, nsr_f16x3_core.h) for the ReLU masks;
// outputs: gradient panels in the forward panels' layout (round 5: the fp16 `hi` operand registers this kernel makes for
// its own next layer, two 16-byte unit stores per block; stored value x the point's power-of-two `pscale` = the true
// gradient rounded to 11 bits) -- the A operands of the weight-gradient kernel (nsr_wgrad_f16.hip), which also sums them
// into the bias gradients -- and, per panel, the largest TRUE magnitude written (that kernel's fp16 pre-scale).
// Vector-memory choreography per 16-k-step block: weight DMA of the chunk after next in k-steps 8..13, then the 2
// stores of the pending block and the sign-word load of the next one in k-steps 14, 15 -- behind the DMA, so the next
// publish point (s_waitcnt vmcnt(3)) does not have to wait for them.
//
// Range: gradients sit far below fp16's range and differ by orders of magnitude from point to point, so every point
// (lane pair) carries its own power-of-two scale: the prologue scales the point's inputs to max 2^1..2^2, and each
// layer's outputs are re-normalised by a factor chosen from the measured maximum of the layer before (gains of real
// layers stay far inside the 2^7 of headroom this leaves below fp16's overflow).  All factors are powers of two, and
// the stored gradients are multiplied back to true scale in fp32, so the scaling itself is exact.
#include "nsr_f16x3_core.h"
#include "nsr_train_chain.h"
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------------------
// stream: 9 layers x 8 chunks (one 32-feature output block each) of 32 pieces, no bias pieces
// ---------------------------------------------------------------------------------------------------------
constexpr int kBwdLayers = 9;
constexpr int kBwdChunkPieces = 32;            // every chunk: 16 k-steps x (hi, lo)
__device__ __host__ __forceinline__ int bwd_piece0(int lam, int nb) { return 256 * lam + 32 * nb; }
constexpr int kBwdPieces = 256 * kBwdLayers;   // 2304 pieces of 1 KiB
constexpr int kBwdAuxFloats = 768;             // rgb.weight as [feature][4], then sigma.weight [256] (fp32, true scale)

struct BwdPackPtrs {
  const float* p[NSR_N_STATE_TENSORS];
};

// one thread per 32-bit word.  Piece (layer lam, block nb, k-step s, part): lane (i, h), halves j = 0..7 hold
// 64 W[k = act_feature(8 s + j, h)][column 32 nb + i] of the layer's nn.Linear weight (out, in) -- its transpose as the
// MFMA A operand.
// stop_grad (--stop_grad, models/networks.py:218-219: dir_encoding's input is detached): the g columns of dir_encoding's weight
// are packed as zeros, so the gradient that leaves the colour branch for xyz_encoding_final is an exact 0 at every point --
// xyz_encoding_final's weights and bias get zero gradients, the trunk sees the density head's gradient alone.
// hi_only (round 6, the one-term chain): the stream carries the hi pieces alone -- 16 pieces per chunk, 128 per layer
__device__ __forceinline__ void pack_bwd_body(const BwdPackPtrs& w, unsigned* __restrict__ out, int stop_grad, int mode);
__global__ void __launch_bounds__(256) pack_bwd_kernel(BwdPackPtrs w, unsigned* __restrict__ out, int stop_grad, int mode) {
  pack_bwd_body(w, out, stop_grad, mode);
}
// both networks of a training step in one launch (round 6; blockIdx.y = network) + the step's two start-of-step clears,
// which used to be memset launches of their own: zero[0 .. n_zero) doubles (the loss carries) and one option word
__global__ void __launch_bounds__(256) pack_bwd2_kernel(BwdPackPtrs w0, unsigned* __restrict__ out0, BwdPackPtrs w1, unsigned* __restrict__ out1,
                                                        int stop_grad, int mode, double* __restrict__ zero, int n_zero,
                                                        unsigned* __restrict__ word, unsigned value) {
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    if ((int)threadIdx.x < n_zero) zero[threadIdx.x] = 0.0;
    if (threadIdx.x == 64 && word) *word = value;
  }
  if (blockIdx.y) pack_bwd_body(w1, out1, stop_grad, mode);
  else pack_bwd_body(w0, out0, stop_grad, mode);
}
__device__ __forceinline__ void pack_bwd_body(const BwdPackPtrs& w, unsigned* __restrict__ out, int stop_grad, int mode) {
  // mode: 2 / 3 = (hi, lo) pieces for every layer; 1 = hi pieces only; 12 = mixed (layers 0..5 hi + lo, 6..8 hi only)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_pieces = mode == 12 ? 1536 + 3 * 128 : (mode == 1 ? kBwdPieces / 2 : kBwdPieces);
  const int stream_words = n_pieces * 256;
  if (idx >= stream_words + kBwdAuxFloats) return;
  unsigned v = 0u;
  if (idx < stream_words) {
    const int piece = idx >> 8, word = idx & 255;
    int lam, local;
    bool hi_only;
    if (mode == 12) {
      hi_only = piece >= 1536;
      lam = hi_only ? 6 + ((piece - 1536) >> 7) : piece >> 8;
      local = hi_only ? (piece - 1536) & 127 : piece & 255;
    } else {
      hi_only = mode == 1;
      lam = piece / (hi_only ? 128 : 256);
      local = piece % (hi_only ? 128 : 256);
    }
    const int per_chunk = hi_only ? kBwdChunkPieces / 2 : kBwdChunkPieces;
    const int nb = local / per_chunk, rem = local % per_chunk;
    const int s = hi_only ? rem : rem >> 1, part = hi_only ? 0 : rem & 1;
    const int lane = word >> 2, jj = word & 3;
    const int n = 32 * nb + (lane & 31), h = lane >> 5;
    float f[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = 2 * jj + e;
      float x = 0.0f;
      if (lam == 0) {                                   // dir_encoding.weight (128, 283): columns 0..255 = g
        const int k = act_feature(8 * s + j, h);
        if (k < 128 && !stop_grad) x = w.p[18][k * 283 + n];
      } else if (lam == 1) {                            // xyz_encoding_final.weight (256, 256)
        x = w.p[16][act_feature(8 * s + j, h) * 256 + n];
      } else {                                          // trunk layer l = 10 - lam (8..2), tensor 2 (l - 1)
        const int l = 10 - lam;
        const int k = act_feature(8 * s + j, h);
        x = (l == 5) ? w.p[8][k * 319 + 63 + n] : w.p[2 * (l - 1)][k * 256 + n];
      }
      f[e] = kWScale * x;
    }
    v = pack_hl(f[0], f[1], part);
  } else {
    const int a = idx - stream_words, feat = a >> 2, c = a & 3;
    v = __float_as_uint(a >= 512 ? w.p[20][a - 512] : (c < 3 ? w.p[22][c * 128 + feat] : 0.0f));
  }
  out[idx] = v;
}

__device__ __forceinline__ ChunkRef bwd_ref(int lam, int nb, int wave) {
  return make_ref(bwd_piece0(lam, nb), kBwdChunkPieces, wave);
}
// chunk number q = 8 lam + nb of the stream; past the end chunk 0 is re-fetched into the idle slot (as in the forward)
__device__ __forceinline__ ChunkRef bwd_seq(int q, int wave) {
  return q < 8 * kBwdLayers ? bwd_ref(q >> 3, q & 7, wave) : make_ref(0, 32, wave);
}

// ---------------------------------------------------------------------------------------------------------
// per-point scale bookkeeping (all powers of two, identical in the two lanes of a point)
// ---------------------------------------------------------------------------------------------------------
struct Scale {
  float cinv;      // 1 / (accumulator scale of the stage): accumulator x cinv = true gradient
  float phi;       // accumulator -> next operand factor
  unsigned phi2;   // the same as a packed fp16 pair
  float mx;        // running max |accumulator| (masked) of the stage
};
__device__ __forceinline__ float pow2f(int e) {          // 2^e, e clamped to the normal range
  e = e < -120 ? -120 : (e > 120 ? 120 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}
__device__ __forceinline__ int floor_log2(float x) {     // x > 0; denormals count as 2^-127
  return (int)((__float_as_uint(x) >> 23) & 255u) - 127;
}
__device__ __forceinline__ unsigned pack_h2(float x) {
  const _Float16 hx_ = (_Float16)x;
  const unsigned b = __builtin_bit_cast(unsigned short, hx_);
  return b | (b << 16);
}
// the stage that consumes operands of magnitude `o_max` (already combined over the point's two lanes): factors of its
// outputs, given the scale 1 / cinv_in ... of its accumulators
__device__ __forceinline__ void stage_factors(Scale& cur, const Scale& prev) {
  cur.cinv = prev.cinv * (1.0f / 64.0f) / prev.phi;      // accumulators of this stage = 64 x (operand scale) x truth
  float o = prev.mx * prev.phi;
  o = fmaxf(o, __shfl_xor(o, 32, 64));
  // operands below 2^E: outputs of a gain-1 layer land in [2, 4) x 64 before phi
  const int E = (o > 0.0f) ? floor_log2(o) + 1 : 2;
  int e = -4 - E;
  e = e < -14 ? -14 : (e > 0 ? 0 : e);                   // phi must be a normal fp16
  cur.phi = pow2f(e);
  cur.phi2 = pack_h2(cur.phi);
  cur.mx = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------
// re-split of a finished gradient block (in place in its accumulator registers)
//   MASK:  x = acc * [z > 0]                         (v_cmp + v_cndmask per element)
//   CONV:  hi = RNE_f16(x) * phi (v_cvt_pk, v_pk_mul), lo = RNE_f16(x phi - hi) (v_fma_mix), max |x| tracked
//   hi is also what the caller stores to the gradient panel: hi = RNE_f16(the true gradient / cinv) * phi with the POINT's
//   powers of two cinv, phi of this layer; pscale = cinv / phi is kept once per point and panel and applied by the consumer
// 17 half-steps as in the forward kernel: half-step 2P = first part of pair P, 2P + 1 = second part of pair P and the
// lo of pair P - 1.
// ---------------------------------------------------------------------------------------------------------
struct BwdTmp {
  unsigned hi[2];
};
// mask: the sign word of the block's pair (nsr_f16x3_core.h: register r in bit HB + 15 - r, HB = 16 for the even block of the
// pair, 0 for the odd one; set = the ReLU zeroed it): v_bfe_i32 spreads the bit over a register, v_bfi_b32 keeps the
// accumulator where it is clear
template <int P, bool MASK, bool CONV, int HB>
__device__ __forceinline__ void bsplit_a(Acc& p, unsigned mz, Scale& sc, BwdTmp& t) {
  unsigned& hi = t.hi[P & 1];
  unsigned t0, t1;
  if (MASK && CONV)
    asm volatile(
        "v_bfe_i32 %4, %6, %7, 1\n\t"
        "v_bfe_i32 %5, %6, %8, 1\n\t"
        "v_bfi_b32 %0, %4, 0, %0\n\t"
        "v_bfi_b32 %1, %5, 0, %1\n\t"
        "v_cvt_pk_f16_f32 %2, %0, %1\n\t"
        "v_max3_f32 %3, |%0|, |%1|, %3"
        : "+v"(p.m[2 * P]), "+v"(p.m[2 * P + 1]), "=&v"(hi), "+v"(sc.mx), "=&v"(t0), "=&v"(t1)
        : "v"(mz), "n"(HB + 15 - 2 * P), "n"(HB + 14 - 2 * P));
  else
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_max3_f32 %1, |%2|, |%3|, %1"
        : "=&v"(hi), "+v"(sc.mx)
        : "v"(p.m[2 * P]), "v"(p.m[2 * P + 1]));
}
template <int P>
__device__ __forceinline__ void bput(unsigned v, u32x4& d0, u32x4& d1) {
  if (P < 4) d0[P & 3] = v; else d1[P & 3] = v;
}
template <int P, bool CONV>   // P = 0..8
__device__ __forceinline__ void bsplit_b(Acc& p, const Scale& sc, BwdTmp& t, u32x4& h0, u32x4& l0, u32x4& h1, u32x4& l1) {
  if (!CONV) return;
  unsigned& cur = t.hi[P & 1];
  const unsigned prev = t.hi[(P & 1) ^ 1];
  unsigned lo = 0;
  constexpr int Q = P > 0 ? P - 1 : 0;   // the pair whose lo is made here
  if (P == 0) {
    asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(cur) : "v"(sc.phi2));
  } else if (P < 8) {
    asm volatile(
        "v_fma_mixlo_f16 %1, %2, %4, -%5 op_sel_hi:[0,0,1]\n\t"
        "v_pk_mul_f16 %0, %0, %6\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "+v"(cur), "=&v"(lo)
        : "v"(p.m[2 * Q]), "v"(p.m[2 * Q + 1]), "v"(sc.phi), "v"(prev), "v"(sc.phi2));
  } else {
    asm volatile(
        "v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo)
        : "v"(p.m[2 * Q]), "v"(p.m[2 * Q + 1]), "v"(sc.phi), "v"(prev));
  }
  if (P < 8) bput<(P < 8 ? P : 0)>(cur, h0, h1);
  if (P > 0) bput<Q>(lo, l0, l1);
}
template <bool MASK, bool CONV, int HB>
__device__ __forceinline__ void bwd_half(int hs, Acc& p, unsigned mz, Scale& sc, BwdTmp& t, u32x4& h0, u32x4& l0,
                                         u32x4& h1, u32x4& l1) {
  switch (hs) {
#define NSR_HS(P)                                                         \
    case 2 * P: bsplit_a<P, MASK, CONV, HB>(p, mz, sc, t); break;         \
    case 2 * P + 1: bsplit_b<P, CONV>(p, sc, t, h0, l0, h1, l1); break;
    NSR_HS(0) NSR_HS(1) NSR_HS(2) NSR_HS(3) NSR_HS(4) NSR_HS(5) NSR_HS(6) NSR_HS(7)
#undef NSR_HS
    case 16: bsplit_b<8, CONV>(p, sc, t, h0, l0, h1, l1); break;
    default: break;
  }
}
// Schedule over the gaps of a 16-step chunk (block_mma3, round 5; rounds 2-4 ran this kernel on the k-step-granular
// block_mma, whose fillers all queue behind the k-step's first MFMA): half-step i (0..15) in gap 1 + (i & 1) of k-step
// i >> 1, the last one (two instructions) in gap 0 of k-step 8 -- the forward kernel's pending_gap.  Everything is done
// before the publish point's DMA gaps (k-steps 8..13) and long before the operands of k-steps 14, 15 are used.
template <bool MASK, bool CONV, int HB>
__device__ __forceinline__ void bwd_gap(int s, int g, Acc& p, unsigned mz, Scale& sc, BwdTmp& t, u32x4& h0, u32x4& l0,
                                        u32x4& h1, u32x4& l1) {
  if (g == 0) {
    if (s == 8) bwd_half<MASK, CONV, HB>(16, p, mz, sc, t, h0, l0, h1, l1);
    return;
  }
  const int i = 2 * s + g - 1;
  if (i < 16) bwd_half<MASK, CONV, HB>(i, p, mz, sc, t, h0, l0, h1, l1);
}
// Stores of the pending block: its two units in k-steps 14 and 15 (the hi registers are final after half-step 15 =
// k-step 12), i.e. BEHIND the chunk's last DMA piece (k-step 13).  Vector-memory operations complete in issue order, so
// the next publish point -- which must see that DMA landed -- can leave these stores in flight (block_mma's YOUNGER
// count); they only have to be done one block later.  (With the stores spread over k-steps 8..13 the publish point's
// wait covered them too and the wave stalled on HBM write latency every block.)
// These are PLAIN (compiler-visible, non-temporal) stores, unlike the forward kernel's asm ones: the kernel also has
// compiler-visible loads (the mask words), and the compiler's vmcnt for a load only counts the younger operations it
// can see -- with invisible stores behind the load its wait would cover them as well.
template <int U>
__device__ __forceinline__ void bwd_unit_store(const u32x4& v, const char* blk, unsigned voff) {
#ifdef NSR_ABL_BWD_NO_STORE   // ablation (scripts/): how much of the kernel is its panel writes
  if (voff != 0xffffffffu) return;
#endif
  u32x4* dst = reinterpret_cast<u32x4*>(const_cast<char*>(blk) + U * 1024 + voff);
#ifdef NSR_ABL_BWD_STORE_L2   // ablation: the same store instructions, but every block of a wave lands on ONE 2 KiB run (L2-resident: no HBM writes)
  dst = reinterpret_cast<u32x4*>(reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(blk) & ~(uintptr_t)0x3FFF)) + U * 1024 + voff);
#endif
#ifdef NSR_ABL_BWD_DEFAULT_STORE   // A/B: the default cache policy instead of non-temporal
  *dst = v;
#else
  __builtin_nontemporal_store(v, dst);
#endif
}
__device__ __forceinline__ void bwd_store_step(int s, const u32x4& h0, const u32x4& h1, const char* blk, unsigned voff0,
                                               unsigned voff1) {
  if (s == 14) bwd_unit_store<0>(h0, blk, voff0);
  if (s == 15) bwd_unit_store<1>(h1, blk, voff1);
}

// Mask fetch: the sign word this lane needs to mask blocks X, X + 1 (X even: the two share a dword) travels by LDS-DMA into a
// 256-byte area of the wave's own (round 5), issued in k-step 6 of block X - 1 -- BEFORE that chunk's weight pieces (k-steps
// 8..13; M0 is written by every group's first piece, so the detour leaves no stale M0 behind), hence older than everything
// the next publish point (block X, k-step 8) waits for: by then it has landed.  The register copy is a plain ds_read in
// k-step 14 of block X, a whole block before its first use (the re-split of X runs in the shadow of block X + 1).  Two words
// are alive at a time (mz[pair & 1]).  (Rounds 2-4 used a plain global load here: hipcc's vmcnt for it can only count the
// operations it sees, i.e. not the asm DMA pieces issued behind the load, so its wait at the first use may also cover the
// chunk's youngest DMA pieces and stores.  Measured, interleaved on one box: 676.8 -> 671.7 us per pass -- the wait was rarely
// exposed; what the DMA form buys is that the publish points' vmcnt accounting has no compiler-visible load left in it.)
__device__ __forceinline__ void mask_dma(const unsigned* blk, unsigned lane4, unsigned lds_dst) {
  unsigned long long tmp;
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_mov_b64 %0, %2\n\t"
      "global_load_lds_dword %1, %0"
      : "=&s"(tmp)
      : "v"(lane4), "s"(blk), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ unsigned mask_read(unsigned lds_addr_lane) {
  return *(const __attribute__((address_space(3))) unsigned*)(size_t)lds_addr_lane;
}

struct BwdCtx {
  const unsigned* sgn;   // sign panels of the forward pass (masks)
  PanelRef dp;           // gradient panels
  unsigned voff0, voff1; // this lane's slot in unit 0 / 1 of a block (unit_voff)
  int lane;
  unsigned lane4;     // lane * 4
  unsigned lmask;     // LDS byte address of this wave's 256-byte mask area (mask_dma / mask_read)
  unsigned* lmax;     // LDS, 16 words: per gradient panel, float bits of the largest magnitude this workgroup wrote
  float* pscale;      // (10 panels, padded points): stored value x pscale = true gradient
  int64_t pidx;       // this lane's point slot (group * 32 + m); lanes of the upper half do not write
  bool writer;
};
__device__ __forceinline__ void store_pscale(const BwdCtx& cx, int panel, float v) {
  if (cx.writer) cx.pscale[(int64_t)panel * cx.dp.n_groups * 32 + cx.pidx] = v;
}
// largest true-scale magnitude written to gradient panel `panel` (the weight-gradient kernel scales by it): the wave's
// maximum by four DPP steps inside the rows of 16 lanes + four v_readlane, ONE single-lane LDS atomic per wave, one global
// atomicMax per workgroup and panel at the end of the kernel.  (Until the last session of round 5 every lane issued the LDS
// atomic: 64 read-modify-writes of ONE address serialise in the LDS pipe, and the A-fragment reads of the following k-steps
// queue behind them -- the chunk timeline (scripts/bwd_timeline.py) showed block 1 of every layer at 6,600 cycles against
// 2,000 for its neighbours, 37 k of the kernel's 201 k cycles per tile.)
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v, unsigned other) { return v > other ? v : other; }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  // non-negative floats order like their bit patterns
  v = dpp_max_u32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1, 0, 3, 2]
  v = dpp_max_u32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2, 3, 0, 1]
  v = dpp_max_u32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = dpp_max_u32(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ void publish_max(const BwdCtx& cx, int panel, float v) {
  const unsigned w = wave_max_u32(__float_as_uint(fabsf(v)));
  if (cx.lane == 0) atomicMax(cx.lmax + panel, w);
}

// One layer `lam` of the chain: operands (bh, bl) -> eight output blocks.
//   PREV_MASK / PREV_PANEL: the layer before (whose block 7 is still pending on entry): its mask flag and gradient panel
//   MASK: this layer's outputs are masked by forward panel `panel` (and written to gradient panel `panel`)
//   ADD: this layer's finished blocks get the density head's rank-1 term, acc += wsigma[feature] * sig (sig = d_sigma at
//        the accumulators' scale) -- 16 FMAs per block of layer 1 instead of streaming a (mostly zero) k-step for it
//   LAST: the last layer (nothing consumes its operands in this kernel, but they are what is stored)
//   NEXT_MASK / next_panel: the layer after this one, whose first block's masks are fetched during this layer's last block
//   FIRST: no layer before this one (layer 0): block 0 has nothing pending to convert or store
template <bool PREV_MASK, bool MASK, bool ADD, bool LAST, bool NEXT_MASK, bool FIRST = false>
__device__ __forceinline__ void bwd_layer(int lam, int prev_panel, int panel, int next_panel, u32x4 (&bh)[16], u32x4 (&bl)[16], u32x4 (&oh)[16],
                                          u32x4 (&ol)[16], const float* wsig_h, float d_sigma, Loader& ld, Acc& pend,
                                          Pre& pre, unsigned (&mz)[2], Scale& prev, const BwdCtx& cx) {
  Scale cur{};
  // d_sigma at the scale of this layer's accumulators (1 / cinv of this layer, known from the layer before: exact powers of two)
  const float sig = ADD ? d_sigma / (prev.cinv * (1.0f / 64.0f) / prev.phi) : 0.0f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    const int q = 8 * lam + nb;
    const ChunkRef c2 = bwd_seq(q + 2, ld.wave);
    Acc acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.m[r] = 0.0f;
    BwdTmp tmp;
    unsigned a_addr = ld.slot_cur + ld.lane_off;
    Pre nxt;
    if (nb == 1) {                             // the previous layer's last block was measured during block 0
      if (prev_panel >= 0) publish_max(cx, prev_panel, prev.mx * prev.cinv);
      stage_factors(cur, prev);
      store_pscale(cx, panel, cur.cinv * (1.0f / cur.phi));   // phi is a power of two >= 2^-14: exact
    }
    // mask word of the pair that starts with the block AFTER this one (loaded here if that block is even: nb odd), and of the
    // PENDING block's pair.  Pair q of a layer lives in mz[q & 1]; every layer has four pairs, so the previous layer's last
    // pair (block 7 pending during this layer's block 0) is mz[1] and never collides with this layer's pair 0
    const bool load_next = (nb & 1) && ((nb < 7) ? MASK : NEXT_MASK);
    const unsigned* next_blk =
        !load_next ? nullptr
                   : (nb < 7 ? sign_block(const_cast<unsigned*>(cx.sgn), cx.dp.group, panel, nb + 1)
                             : sign_block(const_cast<unsigned*>(cx.sgn), cx.dp.group, next_panel, 0));
    const int pb = (nb == 0) ? 7 : nb - 1;            // the pending block's index in its layer
    const unsigned mz_pend = mz[(pb >> 1) & 1];
    // younger than the DMA this chunk's publish point waits for: what the block before issued in its k-steps 14, 15,
    // behind its last DMA piece (k-step 13): the 2 stores of ITS pending block (never over-counted: a surplus would leave a
    // DMA piece in flight; the mask DMA goes out BEFORE a chunk's pieces)
    const int kYoung = (FIRST && nb <= 1) ? 0 : 2;
    auto mma = [&](auto young) {
    block_mma3<16, kBar, decltype(young)::value>(
        acc, pre, a_addr, ld, c2, [&](int s, int part) -> u32x4 { return part ? bl[s] : bh[s]; },
        [&](int s, int g) {
          if (nb == 0) {
            // block 7 of the layer before (the odd block of its pair) -> k-steps 14, 15 of THIS layer's input
            bwd_gap<PREV_MASK, true, 0>(s, g, pend, mz_pend, prev, tmp, bh[14], bl[14], bh[15], bl[15]);
            if (prev_panel >= 0 && g == 2) bwd_store_step(s, bh[14], bh[15], panel_block(cx.dp, prev_panel, 7), cx.voff0, cx.voff1);
          } else {
            if (pb & 1) bwd_gap<MASK, true, 0>(s, g, pend, mz_pend, cur, tmp, oh[2 * nb - 2], ol[2 * nb - 2], oh[2 * nb - 1], ol[2 * nb - 1]);
            else bwd_gap<MASK, true, 16>(s, g, pend, mz_pend, cur, tmp, oh[2 * nb - 2], ol[2 * nb - 2], oh[2 * nb - 1], ol[2 * nb - 1]);
            if (g == 2) bwd_store_step(s, oh[2 * nb - 2], oh[2 * nb - 1], panel_block(cx.dp, panel, nb - 1), cx.voff0, cx.voff1);
          }
          if (load_next && g == 1 && s == 6) mask_dma(next_blk, cx.lane4, cx.lmask);         // the pair that opens with block nb + 1
          if (MASK && !(nb & 1) && g == 1 && s == 14) mz[(nb >> 1) & 1] = mask_read(cx.lmask + cx.lane4);   // this block opens a pair
        },
        [&](int k, int g) {
          if (g == 0) prefetch_frag(nxt, k, ld.slot_next + ld.lane_off);
        });
    };
    if (kYoung == 2) mma(std::integral_constant<int, 2>{});
    else mma(std::integral_constant<int, 0>{});
    if (ADD) {   // plain C++ on purpose: the compiler inserts the MFMA -> VALU wait states
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.m[r] = fmaf(wsig_h[32 * nb + 8 * (r >> 2) + (r & 3)], sig, acc.m[r]);
    }
    pend = acc;
    pre = nxt;
    loader_advance(ld);
  }
  prev = cur;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
chain_bwd_kernel(const float* __restrict__ packed, const unsigned* __restrict__ sgn, char* __restrict__ dpan,
                 const float* __restrict__ d_rgb, int d_rgb_stride, const float* __restrict__ d_sigma, int d_sigma_stride,
                 int64_t P, unsigned* __restrict__ gmax, float* __restrict__ pscale) {
  constexpr int kAux0 = 3 * kSlotFloats;
  __shared__ __attribute__((aligned(16))) float ring[kAux0 + kBwdAuxFloats + 16 + 4 * 64];
  unsigned* lmax = reinterpret_cast<unsigned*>(ring + kAux0 + kBwdAuxFloats);
  if (threadIdx.x < 16) lmax[threadIdx.x] = 0u;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < kBwdAuxFloats; i += 256) ring[kAux0 + i] = packed[kBwdPieces * 256 + i];

  Loader ld;
  ld.stream = packed;
  ld.wave = wave;
  ld.lane_off = (unsigned)lane * 16u;
  ld.slot_cur = lds_addr(ring);
  ld.slot_next = ld.slot_cur + kSlotBytes;
  ld.slot_free = ld.slot_cur + 2 * kSlotBytes;
  // chunks 0 and 1 stream in behind the prologue
  loader_prepare_dma(ld, bwd_seq(0, wave), ld.slot_cur);
#pragma unroll
  for (int i = 0; i < 8; ++i) loader_issue(ld, i);
  loader_prepare_dma(ld, bwd_seq(1, wave), ld.slot_next);
#pragma unroll
  for (int i = 0; i < 8; ++i) loader_issue(ld, i);

  const int64_t p = (int64_t)blockIdx.x * 128 + wave * 32 + m;
  const int64_t pc = p < P ? p : P - 1;
  BwdCtx cx;
  cx.sgn = sgn;
  cx.dp.base = dpan;
  cx.dp.n_groups = (int64_t)gridDim.x * 4;
  cx.dp.group = (int64_t)blockIdx.x * 4 + wave;
  cx.dp.sgn = nullptr;
  cx.voff0 = unit_voff(m, h, 0);
  cx.voff1 = unit_voff(m, h, 1);
  cx.lane = lane;
  cx.lane4 = (unsigned)lane * 4u;
  cx.lmask = lds_addr(ring + kAux0 + kBwdAuxFloats + 16) + (unsigned)wave * 256u;
  cx.lmax = lmax;
  cx.pscale = pscale;
  cx.pidx = cx.dp.group * 32 + m;
  cx.writer = h == 0;

  // ---- prologue: the colour head's input gradient on the VALU (K = 3), masked by dir_encoding's ReLU
  const float g0 = d_rgb[pc * d_rgb_stride + 0], g1 = d_rgb[pc * d_rgb_stride + 1], g2 = d_rgb[pc * d_rgb_stride + 2];
  const float gs = d_sigma[pc * d_sigma_stride];
  unsigned zc[2];   // sign words of dir_encoding's four output blocks (two per dword)
#pragma unroll
  for (int b = 0; b < 2; ++b) zc[b] = sign_block(const_cast<unsigned*>(sgn), cx.dp.group, 9, 2 * b)[lane];
  __syncthreads();   // aux visible
  float dz[64];
  float mxin = fabsf(gs);
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    const int feat = act_feature(t, h);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(ring + kAux0 + 4 * feat);
    float v = __fmaf_rn(w4[2], g2, __fmaf_rn(w4[1], g1, __fmul_rn(w4[0], g0)));
    v = ((zc[t >> 5] >> ((((t >> 4) & 1) ? 0 : 16) + 15 - (t & 15))) & 1u) ? 0.0f : v;
    dz[t] = v;
    mxin = fmaxf(mxin, fabsf(v));
  }
  {
    float mdz = 0.0f;
#pragma unroll
    for (int t = 0; t < 64; ++t) mdz = fmaxf(mdz, fabsf(dz[t]));
    publish_max(cx, 9, mdz);
  }
  mxin = fmaxf(mxin, __shfl_xor(mxin, 32, 64));
  // the point's scale: its largest input lands in [2, 4)
  const float S = mxin > 0.0f ? pow2f(1 - floor_log2(mxin)) : 1.0f;
  store_pscale(cx, 9, 1.0f / S);   // the colour head's input gradient (prologue) is stored as hi of dz * S
  u32x4 bh[16], bl[16], oh[16], ol[16];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    unsigned a[4], b[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) split2(dz[8 * s + 2 * pr] * S, dz[8 * s + 2 * pr + 1] * S, a[pr], b[pr]);
    bh[s] = u32x4{a[0], a[1], a[2], a[3]};
    bl[s] = u32x4{b[0], b[1], b[2], b[3]};
  }
#pragma unroll
  for (int s = 8; s < 16; ++s) {
    bh[s] = u32x4{0u, 0u, 0u, 0u};
    bl[s] = u32x4{0u, 0u, 0u, 0u};
  }
  // gradient panel 9: the operand registers just made (block b = k-steps 2b, 2b + 1)
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const char* blk = panel_block(cx.dp, 9, b);
    unit_store<0>(bh[2 * b], blk, cx.voff0);
    unit_store<1>(bh[2 * b + 1], blk, cx.voff1);
  }
  const float* wsig_h = ring + kAux0 + 512 + 4 * h;   // sigma.weight, this lane half's features (+ 32 nb + 8 (r >> 2) + (r & 3))
  // "previous stage" of layer 0 = the prologue: operands at scale S with max in [2, 4)
  Scale prev;
  prev.cinv = 1.0f / S;      // so that stage_factors() gives layer 0 the accumulator scale 64 S
  prev.phi = 1.0f;
  prev.phi2 = 0u;
  prev.mx = 3.0f;            // E = 2 -> phi of layer 0 = 2^-6

  // chunk 0 published, first fragments in registers
  dma_drain();
  __syncthreads();
  Pre pre;
#pragma unroll
  for (int k = 0; k < kPF; ++k) prefetch_frag(pre, k, ld.slot_cur + ld.lane_off);
  loader_prepare_dma(ld, bwd_seq(2, wave), ld.slot_free);   // replaced at the first publish point; keeps the descriptor defined

  Acc pend;
#pragma unroll
  for (int r = 0; r < 16; ++r) pend.m[r] = 0.0f;
  unsigned mz[2] = {0u, 0u};

  // layer 0: dg (no mask) -> gradient panel 8.  Its block-0 hook sees a dummy pending block (zeros, converted into the
  // zero padding of its own input, stored nowhere)
  bwd_layer<false, false, false, false, true, true>(0, -1, 8, 7, bh, bl, oh, ol, wsig_h, gs, ld, pend, pre, mz, prev, cx);
  bwd_layer<false, true, true, false, true>(1, 8, 7, 6, oh, ol, bh, bl, wsig_h, gs, ld, pend, pre, mz, prev, cx);
  // layers 2..8: trunk layers 8..2; outputs dz7..dz1 -> panels 6..0
#pragma unroll 1
  for (int pair = 0; pair < 3; ++pair) {
    const int lam = 2 + 2 * pair;
    bwd_layer<true, true, false, false, true>(lam, 9 - lam, 8 - lam, 7 - lam, bh, bl, oh, ol, wsig_h, gs, ld, pend, pre, mz, prev,
                                              cx);
    bwd_layer<true, true, false, false, true>(lam + 1, 8 - lam, 7 - lam, 6 - lam, oh, ol, bh, bl, wsig_h, gs, ld, pend, pre, mz,
                                              prev, cx);
  }
  bwd_layer<true, true, false, true, false>(8, 1, 0, -1, bh, bl, oh, ol, wsig_h, gs, ld, pend, pre, mz, prev, cx);
  // the last block of dz1: mask, convert with the last layer's factors (`prev` now), store
  {
    BwdTmp tmp;
    // the accumulators were written by the MFMA just issued, and the hazard recognizer does not look inside inline asm:
    // give the matrix pipe its write-back latency (18 wait states for a 16-pass MFMA) before the asm below reads them
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int hs = 0; hs < 17; ++hs) bwd_half<true, true, 0>(hs, pend, mz[1], prev, tmp, bh[0], bl[0], bh[1], bl[1]);
    publish_max(cx, 0, prev.mx * prev.cinv);
    const char* blk = panel_block(cx.dp, 0, 7);
    unit_store<0>(bh[0], blk, cx.voff0);
    unit_store<1>(bh[1], blk, cx.voff1);
  }
  dma_drain();   // no LDS-DMA may be in flight when the workgroup's LDS is released
  __syncthreads();
  if (threadIdx.x < 10) atomicMax(gmax + threadIdx.x, lmax[threadIdx.x]);
}


// =========================================================================================================
// Reduced-term variants of the chain (round 6): NT = 2 or 1 MFMAs per product instead of three.
//   NT = 2:  W_hi g_hi + W_lo g_hi   -- the weights keep their 22 bits, the incoming gradient is its fp16 `hi` alone
//   NT = 1:  W_hi g_hi               -- both operands rounded to 11 bits; the stream carries the hi pieces only (16 KiB chunks)
// The gradient's `hi` registers are what the chain has stored for the weight-gradient kernel since round 5, so nothing about
// the panels, the sign words, pscale or gmax changes; what goes away is the `lo` operand set (64 registers), its two
// v_fma_mix per register pair, and one (NT = 2) or two (NT = 1) of the three MFMAs.  Error of a gradient tensor against
// the fp64 oracle, predicted on the CPU (scripts/study_bwd_terms.py -> profiles/r6_bwd_terms_study.txt) and measured on
// the device (tests/test_gpu_train.py, tolerances unchanged: 2e-3 of the norm, 5e-4 on the heads): see DESIGN 7.1.
// A workgroup is still 4 waves x 32 points; with <= 256 registers and a 48 KiB ring (NT = 1) two workgroups share a CU,
// so one wave's re-split VALU, LDS reads and DMA issue run under the other's MFMAs.
// =========================================================================================================
template <int NT, int W>
struct HCfg {
  static_assert(NT == 1 || NT == 2, "reduced-term chain: one or two MFMAs per product");
  static_assert(W == 4 || W == 8, "waves per workgroup: one or two per SIMD");
  static constexpr int kChunkPieces = NT == 1 ? 16 : 32;       // 16 k-steps x hi (x lo)
  static constexpr int kSlotB = kChunkPieces * 1024;           // one ring slot
  static constexpr int kLayerPieces = 8 * kChunkPieces;
  static constexpr int kPieces = kLayerPieces * kBwdLayers;
  static constexpr int kMine = kChunkPieces / W;               // pieces of every chunk one wave moves
  // ring depth: the chunk of sequence number q lives in slot q % kDepth and is fetched kDepth - 1 chunks ahead, during chunk
  // q - kDepth + 1, into the slot chunk q - kDepth has just left.
  static constexpr int kDepth = 3;
  // Deeper rings were built and measured (NT = 1: 8 slots, NT = 2: 4): 239.7 / 358.9 us per pass against 243.9 / 359.7 with
  // three -- the ring is not what these kernels wait for -- AND they raced: the mask words travel by LDS-DMA one block ahead
  // of their use (mask_dma), which is older than everything a publish point waits for only while that wait leaves just the
  // 2 youngest operations in flight.  A deeper ring must fetch the masks kDepth - 1 blocks ahead into a ring of their own.
  static_assert(kDepth == 3, "mask_dma's landing is only covered by the publish points of a three-slot ring");
  // vector-memory operations that are provably YOUNGER than the pieces of chunk b + 1 at the publish point of block b in
  // the steady state (operations complete in issue order; per block: [mask fetch, 0 or 1][kMine pieces][2 stores]; the
  // mask fetches are not counted, so the wait may cover the oldest of these stores as well -- never a piece too few):
  // the 2 stores of the block that issued chunk b + 1, then kDepth - 3 whole blocks
  static constexpr int kYoungSteady = 2 + (kDepth - 3) * (kMine + 2);
  static_assert(kYoungSteady <= 60, "vmcnt is a 6-bit counter");
};
// the same count for block `nb` of the FIRST layer (global block b = nb): block 0 stores nothing (no pending block), and the
// chunks 1 .. kDepth - 2 were fetched -- and waited for -- by the prologue (63 = no wait)
template <int NT, int W>
constexpr int hyoung_first(int nb) {
  using C = HCfg<NT, W>;
  const int lo = nb - C::kDepth + 2;           // the block that issued chunk nb + 1
  if (lo < 0) return 63;
  int n = lo == 0 ? 0 : 2;
  for (int i = lo + 1; i < nb; ++i) n += C::kMine + (i == 0 ? 0 : 2);
  return n;
}
template <int NT, int W>
struct HOcc {
  // waves per SIMD: W = 8 is one workgroup of two waves per SIMD sharing the ring (256 points per tile: half the L2 -> LDS
  // weight traffic per point); W = 4 is one wave per SIMD
  static constexpr int kWavesPerEu = W == 8 ? 2 : 1;
};
// the contiguous share of a chunk that wave `wave` of W moves
template <int W>
__device__ __forceinline__ ChunkRef make_ref_w(int piece0, int pieces, int wave) {
  ChunkRef c;
  c.piece0 = piece0;
  c.pieces = pieces;
  c.first = (wave * pieces) / W;
  c.count = ((wave + 1) * pieces) / W - c.first;
  return c;
}
// MODE = MFMA terms per product over the nine layers of the chain: 1 or 2 = that many everywhere; 12 = MIXED (round 6): two
// terms on the six layers nearest the output (dir_encoding, xyz_encoding_final, trunk 8 .. 5), one on the three below (trunk
// 4 .. 2).  A weight-rounding error injected near the output reaches every tensor below it, one injected near the input reaches
// few: profiles/r6_bwd_terms_study.txt section 3 -- 6 + 3 is the first mix that keeps the whole gradient inside 2e-4 with
// margin.  Per layer the stream, the ring geometry and the k-steps are that layer's NT's; a layer's last two blocks fetch, and
// its last block prefetches, chunks of the NEXT layer's geometry.
template <int MODE>
__device__ __host__ constexpr int nt_of(int lam) { return MODE == 12 ? (lam < 6 ? 2 : 1) : MODE; }
template <int MODE>
__device__ __host__ constexpr int max_nt() { return MODE == 12 ? 2 : MODE; }
template <int MODE>
__device__ __host__ constexpr int layer_piece0(int lam) {      // first 1 KiB piece of layer lam in the stream
  return MODE == 12 ? (lam < 6 ? 256 * lam : 1536 + 128 * (lam - 6)) : (MODE == 1 ? 128 : 256) * lam;
}
template <int MODE>
__device__ __host__ constexpr int stream_pieces() { return layer_piece0<MODE>(kBwdLayers); }
template <int MODE, int W>
__device__ __forceinline__ ChunkRef hseq(int q, int wave) {   // chunk q = 8 lam + nb; past the end chunk 0 again (idle slot)
  const int qq = q < 8 * kBwdLayers ? q : 0;
  const int lam = qq >> 3, nb = qq & 7;
  const int pieces = (MODE == 12 ? (lam < 6 ? 2 : 1) : MODE) == 1 ? 16 : 32;
  return make_ref_w<W>(layer_piece0<MODE>(lam) + nb * pieces, pieces, wave);
}
struct PreH {
  u32x4 ah[kPF], al[kPF];   // al: NT = 2 only
};
template <int NT>
__device__ __forceinline__ void prefetch_frag_h(PreH& pre, int k, unsigned seq_addr) {
  const u32x4* a = lds_vec(seq_addr);
  if (NT == 2) {
    pre.ah[k] = a[(2 * k) * 64];
    pre.al[k] = a[(2 * k + 1) * 64];
  } else {
    pre.ah[k] = a[k * 64];
  }
}
// the 16 k-steps of one output block; every MFMA opens a fenced gap (block_mma3's discipline): hook(s, g) runs in gap g of
// k-step s (g < NT), the fragment reads of k-step s + kPF (or next(k)) in gap 0, the chunk's DMA pieces one per gap from
// the publish point on (NT = 2: 8 pieces per wave in k-steps BAR .. BAR + 3; NT = 1: 4 pieces, one per k-step)
template <int NT, int PW, int BAR, int YOUNGER, class BOf, class Hook, class Next>
__device__ __forceinline__ void block_mma_h(Acc& acc, const PreH& pre, unsigned a_addr, Loader& ld, const ChunkRef& c2,
                                            BOf&& b_of, Hook&& hook, Next&& next) {
  constexpr int NSTEP = 16;
  static_assert(PW <= NT * (NSTEP - BAR - 2), "the DMA pieces must be out before the stores of k-steps 14, 15");
  const u32x4* a_pieces = lds_vec(a_addr);
  u32x4 ah[NSTEP], al[NT == 2 ? NSTEP : 1];
#pragma unroll
  for (int s = 0; s < kPF; ++s) {
    ah[s] = pre.ah[s];
    if (NT == 2) al[s] = pre.al[s];
  }
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    if (s == BAR - 3) loader_prepare_dma(ld, c2, ld.slot_free);
    if (s == BAR) loader_publish<YOUNGER, false>(ld, c2);
    const u32x4 bh = b_of(s);
    const int j0 = (s - BAR) * NT;     // DMA issue slot of this k-step's first gap (one piece per gap from the publish point on)
    if (NT == 2) {
      acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al[s]), as_h8(bh), acc.m, 0, 0, 0);
      if (s + kPF < NSTEP) {
        ah[s + kPF] = a_pieces[(2 * (s + kPF)) * 64];
        al[s + kPF] = a_pieces[(2 * (s + kPF) + 1) * 64];
      } else {
        next(s + kPF - NSTEP);
      }
      hook(s, 0);
      if (s >= BAR && j0 < PW) loader_issue(ld, j0);
      __builtin_amdgcn_sched_barrier(0);
      acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bh), acc.m, 0, 0, 0);
      hook(s, 1);
      if (s >= BAR && j0 + 1 < PW) loader_issue(ld, j0 + 1);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bh), acc.m, 0, 0, 0);
      if (s + kPF < NSTEP) ah[s + kPF] = a_pieces[(s + kPF) * 64];
      else next(s + kPF - NSTEP);
      hook(s, 0);
      if (s >= BAR && j0 < PW) loader_issue(ld, j0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// re-split of a finished gradient block without a lo part, per register pair P: X = mask in place (4 VALU, only where the
// layer has a ReLU), Y = hi = RNE_f16(x) * phi + the running maximum (3 VALU).  Sixteen parts.
template <int P, bool MASK, int HB>
__device__ __forceinline__ void hsplit_x(Acc& p, unsigned mz) {
  if (!MASK) return;
  unsigned t0, t1;
  asm volatile(
      "v_bfe_i32 %2, %4, %5, 1\n\t"
      "v_bfe_i32 %3, %4, %6, 1\n\t"
      "v_bfi_b32 %0, %2, 0, %0\n\t"
      "v_bfi_b32 %1, %3, 0, %1"
      : "+v"(p.m[2 * P]), "+v"(p.m[2 * P + 1]), "=&v"(t0), "=&v"(t1)
      : "v"(mz), "n"(HB + 15 - 2 * P), "n"(HB + 14 - 2 * P));
}
template <int P>
__device__ __forceinline__ void hsplit_y(const Acc& p, Scale& sc, u32x4& h0, u32x4& h1) {
  unsigned hi;
  asm volatile(
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_max3_f32 %1, |%2|, |%3|, %1\n\t"
      "v_pk_mul_f16 %0, %0, %4"
      : "=&v"(hi), "+v"(sc.mx)
      : "v"(p.m[2 * P]), "v"(p.m[2 * P + 1]), "v"(sc.phi2));
  bput<P>(hi, h0, h1);
}
template <bool MASK, int HB>
__device__ __forceinline__ void hsplit_part(int i, Acc& p, unsigned mz, Scale& sc, u32x4& h0, u32x4& h1) {
  switch (i) {
#define NSR_HP(P)                                            \
    case 2 * P: hsplit_x<P, MASK, HB>(p, mz); break;         \
    case 2 * P + 1: hsplit_y<P>(p, sc, h0, h1); break;
    NSR_HP(0) NSR_HP(1) NSR_HP(2) NSR_HP(3) NSR_HP(4) NSR_HP(5) NSR_HP(6) NSR_HP(7)
#undef NSR_HP
    default: break;
  }
}
// Where the parts go.  The accumulators of the pending block were written by the MFMAs of the block before and are read
// here by inline asm, which the hazard recognizer does not look into: nothing touches them before TWO MFMAs of the new
// block have issued (the three-term kernel starts in gap 1 of k-step 0 as well).
//   NT = 2: part i in gap number i + 1 (k-step (i + 1) >> 1, gap (i + 1) & 1): done in k-step 8, gap 0
//   NT = 1: register pair P (parts 2P, 2P + 1) in k-step P + 1: done in k-step 8
// i.e. before the operands of k-steps 14, 15 are read (block 0 of a layer converts INTO its own input) and before the
// stores of k-steps 14, 15.
template <int NT, bool MASK, int HB>
__device__ __forceinline__ void hsplit_gap(int s, int g, Acc& p, unsigned mz, Scale& sc, u32x4& h0, u32x4& h1) {
  if (NT == 2) {
    const int i = 2 * s + g - 1;
    if (i >= 0 && i < 16) hsplit_part<MASK, HB>(i, p, mz, sc, h0, h1);
  } else if (s >= 1 && s <= 8) {
    hsplit_part<MASK, HB>(2 * (s - 1), p, mz, sc, h0, h1);
    hsplit_part<MASK, HB>(2 * (s - 1) + 1, p, mz, sc, h0, h1);
  }
}

// bwd_layer without the lo operand set (see there for the flags)
template <int MODE, int NT, int NTN, int W, bool PREV_MASK, bool MASK, bool ADD, bool LAST, bool NEXT_MASK, bool FIRST = false>
__device__ __forceinline__ void bwd_layer_h(int lam, int prev_panel, int panel, int next_panel, u32x4 (&bh)[16], u32x4 (&oh)[16],
                                            const float* wsig_h, float d_sigma, Loader& ld, unsigned ring0, Acc& pend, PreH& pre,
                                            unsigned (&mz)[2], Scale& prev, const BwdCtx& cx) {
  constexpr int LG = NT - 1;   // the last gap of a k-step
  // the slot addresses below are constants: made opaque once per layer, or LICM hoists the sixteen lane addresses of the
  // two-layer loop body out of the loop and the allocator spills (the persistent inference build's lesson)
  asm volatile("" : "+s"(ring0));
  Scale cur{};
  const float sig = ADD ? d_sigma / (prev.cinv * (1.0f / 64.0f) / prev.phi) : 0.0f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    const int q = 8 * lam + nb;
    constexpr int D = HCfg<NT, W>::kDepth;
    constexpr unsigned kSlot = (unsigned)HCfg<max_nt<MODE>(), W>::kSlotB;      // slots are sized for the chain's widest chunk
    const ChunkRef c2 = hseq<MODE, W>(q + D - 1, ld.wave);     // the chunk fetched during this block
    // slots of chunks q, q + 1 and of the chunk fetched now (= the slot chunk q - 1 has left)
    ld.slot_cur = ring0 + ((unsigned)q % D) * kSlot;
    ld.slot_next = ring0 + ((unsigned)(q + 1) % D) * kSlot;
    ld.slot_free = ring0 + ((unsigned)(q + D - 1) % D) * kSlot;
    Acc acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.m[r] = 0.0f;
    unsigned a_addr = ld.slot_cur + ld.lane_off;
    PreH nxt;
    if (nb == 1) {
      if (prev_panel >= 0) publish_max(cx, prev_panel, prev.mx * prev.cinv);
      stage_factors(cur, prev);
      store_pscale(cx, panel, cur.cinv * (1.0f / cur.phi));
    }
    const bool load_next = (nb & 1) && ((nb < 7) ? MASK : NEXT_MASK);
    const unsigned* next_blk =
        !load_next ? nullptr
                   : (nb < 7 ? sign_block(const_cast<unsigned*>(cx.sgn), cx.dp.group, panel, nb + 1)
                             : sign_block(const_cast<unsigned*>(cx.sgn), cx.dp.group, next_panel, 0));
    const int pb = (nb == 0) ? 7 : nb - 1;
    const unsigned mz_pend = mz[(pb >> 1) & 1];
    // the chunk fetched during this block (q + 2) and the one whose first fragments are prefetched at its end (q + 1) belong to
    // the next layer from blocks 6 / 7 on
    auto mma = [&](auto young, auto fetch_nt, auto pre_nt) {
      block_mma_h<NT, HCfg<decltype(fetch_nt)::value, W>::kMine, kBar, decltype(young)::value>(
          acc, pre, a_addr, ld, c2, [&](int s) -> u32x4 { return bh[s]; },
          [&](int s, int g) {
            if (nb == 0) {
              hsplit_gap<NT, PREV_MASK, 0>(s, g, pend, mz_pend, prev, bh[14], bh[15]);
              if (prev_panel >= 0 && g == LG) bwd_store_step(s, bh[14], bh[15], panel_block(cx.dp, prev_panel, 7), cx.voff0, cx.voff1);
            } else {
              if (pb & 1) hsplit_gap<NT, MASK, 0>(s, g, pend, mz_pend, cur, oh[2 * nb - 2], oh[2 * nb - 1]);
              else hsplit_gap<NT, MASK, 16>(s, g, pend, mz_pend, cur, oh[2 * nb - 2], oh[2 * nb - 1]);
              if (g == LG) bwd_store_step(s, oh[2 * nb - 2], oh[2 * nb - 1], panel_block(cx.dp, panel, nb - 1), cx.voff0, cx.voff1);
            }
            if (load_next && g == LG && s == 6) mask_dma(next_blk, cx.lane4, cx.lmask);
            if (MASK && !(nb & 1) && g == LG && s == 14) mz[(nb >> 1) & 1] = mask_read(cx.lmask + cx.lane4);
          },
          [&](int k) { prefetch_frag_h<decltype(pre_nt)::value>(nxt, k, ld.slot_next + ld.lane_off); });
    };
    auto mma_nb = [&](auto young) {
      if (nb < 6) mma(young, std::integral_constant<int, NT>{}, std::integral_constant<int, NT>{});
      else if (nb == 6) mma(young, std::integral_constant<int, NTN>{}, std::integral_constant<int, NT>{});
      else mma(young, std::integral_constant<int, NTN>{}, std::integral_constant<int, NTN>{});
    };
    if (FIRST) {
      switch (nb) {   // nb is a constant after unrolling
#define NSR_YF(B) case B: mma_nb(std::integral_constant<int, hyoung_first<NT, W>(B)>{}); break;
        NSR_YF(0) NSR_YF(1) NSR_YF(2) NSR_YF(3) NSR_YF(4) NSR_YF(5) NSR_YF(6) NSR_YF(7)
#undef NSR_YF
        default: break;
      }
    } else {
      mma_nb(std::integral_constant<int, HCfg<NT, W>::kYoungSteady>{});
    }
    if (ADD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.m[r] = fmaf(wsig_h[32 * nb + 8 * (r >> 2) + (r & 3)], sig, acc.m[r]);
    }
    pend = acc;
    pre = nxt;
  }
  prev = cur;
}

template <int MODE, int W>
__global__ void __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(HOcc<max_nt<MODE>(), W>::kWavesPerEu, HOcc<max_nt<MODE>(), W>::kWavesPerEu)))
chain_bwd_h_kernel(const float* __restrict__ packed, const unsigned* __restrict__ sgn, char* __restrict__ dpan,
                   const float* __restrict__ d_rgb, int d_rgb_stride, const float* __restrict__ d_sigma, int d_sigma_stride,
                   int64_t P, unsigned* __restrict__ gmax, float* __restrict__ pscale) {
  constexpr int NT = max_nt<MODE>();         // ring geometry: the widest chunk
  constexpr int NT0 = nt_of<MODE>(0);
  using C = HCfg<NT, W>;
  constexpr int kAux0 = C::kDepth * C::kSlotB / 4;
  __shared__ __attribute__((aligned(16))) float ring[kAux0 + kBwdAuxFloats + 16 + W * 64];
  unsigned* lmax = reinterpret_cast<unsigned*>(ring + kAux0 + kBwdAuxFloats);
  if (threadIdx.x < 16) lmax[threadIdx.x] = 0u;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < kBwdAuxFloats; i += 64 * W) ring[kAux0 + i] = packed[stream_pieces<MODE>() * 256 + i];

  Loader ld;
  ld.stream = packed;
  ld.wave = wave;
  ld.lane_off = (unsigned)lane * 16u;
  const unsigned ring0 = lds_addr(ring);
  ld.slot_cur = ring0;
  ld.slot_next = ring0 + C::kSlotB;
  ld.slot_free = ring0 + (C::kDepth - 1) * C::kSlotB;
  // chunks 0 .. kDepth - 2 stream in behind the prologue
#pragma unroll
  for (int c = 0; c < C::kDepth - 1; ++c) {
    loader_prepare_dma(ld, hseq<MODE, W>(c, wave), ring0 + (unsigned)c * (unsigned)C::kSlotB);
#pragma unroll
    for (int i = 0; i < HCfg<NT0, W>::kMine; ++i) loader_issue(ld, i);       // chunks 0, 1: layer 0's geometry
  }

  // panels are laid out in point groups of 32 (one wave), four per 128-point tile of the FORWARD kernel: n_groups follows
  // from P alone.  With W = 8 the last workgroup may hold up to four waves past the end: they repeat the last real group
  // (same inputs, same results, same stores: a benign duplicate) so that every wave's DMA / store counts stay uniform.
  const int64_t n_groups = ((P + 127) / 128) * 4;
  int64_t group = (int64_t)blockIdx.x * W + wave;
  group = group < n_groups ? group : n_groups - 1;
  const int64_t p = group * 32 + m;
  const int64_t pc = p < P ? p : P - 1;
  BwdCtx cx;
  cx.sgn = sgn;
  cx.dp.base = dpan;
  cx.dp.n_groups = n_groups;
  cx.dp.group = group;
  cx.dp.sgn = nullptr;
  cx.voff0 = unit_voff(m, h, 0);
  cx.voff1 = unit_voff(m, h, 1);
  cx.lane = lane;
  cx.lane4 = (unsigned)lane * 4u;
  cx.lmask = lds_addr(ring + kAux0 + kBwdAuxFloats + 16) + (unsigned)wave * 256u;
  cx.lmax = lmax;
  cx.pscale = pscale;
  cx.pidx = cx.dp.group * 32 + m;
  cx.writer = h == 0;

  // ---- prologue: as in chain_bwd_kernel (the colour head's input gradient on the VALU, K = 3), hi operands only
  const float g0 = d_rgb[pc * d_rgb_stride + 0], g1 = d_rgb[pc * d_rgb_stride + 1], g2 = d_rgb[pc * d_rgb_stride + 2];
  const float gs = d_sigma[pc * d_sigma_stride];
  unsigned zc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) zc[b] = sign_block(const_cast<unsigned*>(sgn), cx.dp.group, 9, 2 * b)[lane];
  __syncthreads();   // aux visible
  float dz[64];
  float mxin = fabsf(gs);
#pragma unroll
  for (int t = 0; t < 64; ++t) {
    const int feat = act_feature(t, h);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(ring + kAux0 + 4 * feat);
    float v = __fmaf_rn(w4[2], g2, __fmaf_rn(w4[1], g1, __fmul_rn(w4[0], g0)));
    v = ((zc[t >> 5] >> ((((t >> 4) & 1) ? 0 : 16) + 15 - (t & 15))) & 1u) ? 0.0f : v;
    dz[t] = v;
    mxin = fmaxf(mxin, fabsf(v));
  }
  {
    float mdz = 0.0f;
#pragma unroll
    for (int t = 0; t < 64; ++t) mdz = fmaxf(mdz, fabsf(dz[t]));
    publish_max(cx, 9, mdz);
  }
  mxin = fmaxf(mxin, __shfl_xor(mxin, 32, 64));
  const float S = mxin > 0.0f ? pow2f(1 - floor_log2(mxin)) : 1.0f;
  store_pscale(cx, 9, 1.0f / S);
  u32x4 bh[16], oh[16];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    unsigned a[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) a[pr] = pack_hl(dz[8 * s + 2 * pr] * S, dz[8 * s + 2 * pr + 1] * S, 0);
    bh[s] = u32x4{a[0], a[1], a[2], a[3]};
  }
#pragma unroll
  for (int s = 8; s < 16; ++s) bh[s] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const char* blk = panel_block(cx.dp, 9, b);
    unit_store<0>(bh[2 * b], blk, cx.voff0);
    unit_store<1>(bh[2 * b + 1], blk, cx.voff1);
  }
  const float* wsig_h = ring + kAux0 + 512 + 4 * h;
  Scale prev;
  prev.cinv = 1.0f / S;
  prev.phi = 1.0f;
  prev.phi2 = 0u;
  prev.mx = 3.0f;

  dma_drain();
  __syncthreads();
  PreH pre;
#pragma unroll
  for (int k = 0; k < kPF; ++k) prefetch_frag_h<NT0>(pre, k, ld.slot_cur + ld.lane_off);
  loader_prepare_dma(ld, hseq<MODE, W>(C::kDepth - 1, wave), ld.slot_free);   // replaced at the first publish point; keeps the descriptor defined

  Acc pend;
#pragma unroll
  for (int r = 0; r < 16; ++r) pend.m[r] = 0.0f;
  unsigned mz[2] = {0u, 0u};

  if constexpr (MODE == 12) {
    // nine layers written out: each with its own and its successor's geometry (the register sets swap roles layer by layer)
#define NSR_L(LAM, PM, M, ADDF, LASTF, NM, FIRSTF, PP, P_, NP, IN, OUT)                                                                   \
    bwd_layer_h<MODE, nt_of<MODE>(LAM), nt_of<MODE>(LAM < 8 ? LAM + 1 : 0), W, PM, M, ADDF, LASTF, NM, FIRSTF>(LAM, PP, P_, NP, IN, OUT, wsig_h, \
                                                                                                               gs, ld, ring0, pend, pre, mz, prev, cx)
    NSR_L(0, false, false, false, false, true, true, -1, 8, 7, bh, oh);
    NSR_L(1, false, true, true, false, true, false, 8, 7, 6, oh, bh);
    NSR_L(2, true, true, false, false, true, false, 7, 6, 5, bh, oh);
    NSR_L(3, true, true, false, false, true, false, 6, 5, 4, oh, bh);
    NSR_L(4, true, true, false, false, true, false, 5, 4, 3, bh, oh);
    NSR_L(5, true, true, false, false, true, false, 4, 3, 2, oh, bh);
    NSR_L(6, true, true, false, false, true, false, 3, 2, 1, bh, oh);
    NSR_L(7, true, true, false, false, true, false, 2, 1, 0, oh, bh);
    NSR_L(8, true, true, false, true, false, false, 1, 0, -1, bh, oh);
#undef NSR_L
  } else {
  bwd_layer_h<MODE, NT, NT, W, false, false, false, false, true, true>(0, -1, 8, 7, bh, oh, wsig_h, gs, ld, ring0, pend, pre, mz, prev, cx);
  bwd_layer_h<MODE, NT, NT, W, false, true, true, false, true>(1, 8, 7, 6, oh, bh, wsig_h, gs, ld, ring0, pend, pre, mz, prev, cx);
#pragma unroll 1
  for (int pair = 0; pair < 3; ++pair) {
    const int lam = 2 + 2 * pair;
    bwd_layer_h<MODE, NT, NT, W, true, true, false, false, true>(lam, 9 - lam, 8 - lam, 7 - lam, bh, oh, wsig_h, gs, ld, ring0, pend, pre, mz, prev, cx);
    bwd_layer_h<MODE, NT, NT, W, true, true, false, false, true>(lam + 1, 8 - lam, 7 - lam, 6 - lam, oh, bh, wsig_h, gs, ld, ring0, pend, pre, mz, prev, cx);
  }
  bwd_layer_h<MODE, NT, NT, W, true, true, false, true, false>(8, 1, 0, -1, bh, oh, wsig_h, gs, ld, ring0, pend, pre, mz, prev, cx);
  }
  {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA write-back before inline asm reads the accumulators
#pragma unroll
    for (int i = 0; i < 16; ++i) hsplit_part<true, 0>(i, pend, mz[1], prev, bh[0], bh[1]);
    publish_max(cx, 0, prev.mx * prev.cinv);
    const char* blk = panel_block(cx.dp, 0, 7);
    unit_store<0>(bh[0], blk, cx.voff0);
    unit_store<1>(bh[1], blk, cx.voff1);
  }
  dma_drain();
  __syncthreads();
  if (threadIdx.x < 10) atomicMax(gmax + threadIdx.x, lmax[threadIdx.x]);
}

}  // namespace

// terms: 1, 2, 3 MFMA terms per product on every layer, or 12 = mixed (two on layers 0..5, one on 6..8)
static bool bwd_terms_ok(int terms) { return terms == 1 || terms == 2 || terms == 3 || terms == 12; }
static int bwd_stream_pieces(int terms) { return terms == 12 ? 1536 + 3 * 128 : (terms == 1 ? kBwdPieces / 2 : kBwdPieces); }

extern "C" NSR_INTERNAL size_t nsr_chain_bwd_packed_bytes(void) { return 4 * (size_t)(kBwdPieces * 256 + kBwdAuxFloats); }

extern "C" NSR_INTERNAL int nsr_chain_bwd_pack(const float* const* w, void* packed_dev, int stop_grad, int terms, void* stream) {
  if (!bwd_terms_ok(terms)) return NSR_ERR_INVALID_ARG;
  BwdPackPtrs pp;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w[i]) return NSR_ERR_INVALID_ARG;
    pp.p[i] = w[i];
  }
  const int total = bwd_stream_pieces(terms) * 256 + kBwdAuxFloats;
  hipLaunchKernelGGL(pack_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, nsr_stream(stream), pp,
                     static_cast<unsigned*>(packed_dev), stop_grad, terms);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" NSR_INTERNAL int nsr_chain_bwd_pack2(const float* const* w0, void* packed0, const float* const* w1, void* packed1, int stop_grad,
                                                int terms, double* zero, int n_zero, unsigned* word, unsigned value, void* stream) {
  if (!bwd_terms_ok(terms) || n_zero < 0 || n_zero > 64) return NSR_ERR_INVALID_ARG;
  BwdPackPtrs a, b;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w0[i] || !w1[i]) return NSR_ERR_INVALID_ARG;
    a.p[i] = w0[i];
    b.p[i] = w1[i];
  }
  const int total = bwd_stream_pieces(terms) * 256 + kBwdAuxFloats;
  hipLaunchKernelGGL(pack_bwd2_kernel, dim3((total + 255) / 256, 2), dim3(256), 0, nsr_stream(stream), a, static_cast<unsigned*>(packed0), b,
                     static_cast<unsigned*>(packed1), stop_grad, terms, zero, n_zero, word, value);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" NSR_INTERNAL int nsr_chain_bwd(const void* packed, const unsigned* sgn, void* dpan, const float* d_rgb, int d_rgb_stride,
                                          const float* d_sigma, int d_sigma_stride, int64_t P, unsigned* gmax,
                                          float* pscale, int terms, int gmax_is_zero, void* stream) {
  if (!bwd_terms_ok(terms)) return NSR_ERR_INVALID_ARG;
  if (P <= 0) return NSR_OK;
#ifdef NSR_BWD_WAVES   // A/B builds: 4 = one wave per SIMD for the two-term chain / two 4-wave workgroups per CU for the one-term chain
  const int waves = NSR_BWD_WAVES;
#else
  const int waves = 8;
#endif
  // gmax_is_zero: the caller's previous kernel has cleared the ten words (the training step: composite_bwd_kernel)
  if (!gmax_is_zero && hipMemsetAsync(gmax, 0, 10 * sizeof(unsigned), nsr_stream(stream)) != hipSuccess) return NSR_ERR_LAUNCH;
  const dim3 grid((unsigned)((P + 127) / 128)), block(256);
  const dim3 grid8((unsigned)((P + 255) / 256)), block8(512);
  const float* pk = static_cast<const float*>(packed);
  char* dp = static_cast<char*>(dpan);
  hipStream_t st = nsr_stream(stream);
  if (terms == 12)
    hipLaunchKernelGGL((chain_bwd_h_kernel<12, 8>), grid8, block8, 0, st, pk, sgn, dp, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, P, gmax, pscale);
  else if (terms == 3)
    hipLaunchKernelGGL(chain_bwd_kernel, grid, block, 0, st, pk, sgn, dp, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, P, gmax, pscale);
  else if (terms == 2 && waves == 8)
    hipLaunchKernelGGL((chain_bwd_h_kernel<2, 8>), grid8, block8, 0, st, pk, sgn, dp, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, P, gmax, pscale);
  else if (terms == 2)
    hipLaunchKernelGGL((chain_bwd_h_kernel<2, 4>), grid, block, 0, st, pk, sgn, dp, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, P, gmax, pscale);
  else if (waves == 8)
    hipLaunchKernelGGL((chain_bwd_h_kernel<1, 8>), grid8, block8, 0, st, pk, sgn, dp, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, P, gmax, pscale);
  else
    hipLaunchKernelGGL((chain_bwd_h_kernel<1, 4>), grid, block, 0, st, pk, sgn, dp, d_rgb, d_rgb_stride, d_sigma, d_sigma_stride, P, gmax, pscale);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
