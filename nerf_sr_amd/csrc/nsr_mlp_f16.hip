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
#include <utility>

using namespace nsr;
using namespace nsr::hx;

#ifdef NSR_ABL_NO_BARRIER
#define NSR_SYNC() ((void)0)
#else
#define NSR_SYNC() do { dma_drain(); __syncthreads(); } while (0)
#endif

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
      const int n_rows = (c.tensor == 20) ? 1 : 32 * c.nnb;     // sigma head: one real output row
      if (word < n_rows) v = __float_as_uint(w.p[c.tensor + 1][32 * c.nb0 + word]);
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
        f[e] = (col == kPad || !real_row) ? 0.0f : w.p[c.tensor][n * ld + col];
      }
      v = pack_hl(f[0], f[1], part);
    }
  } else {
    const int a = idx - stream_words;
    float f = 0.0f;
    if (a < hx::kAuxRgbB) f = w.p[22][a];
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

// LDS byte address (32-bit) of a __shared__ object
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p);
}

// Weight DMA (global -> LDS, 64 lanes x 16 B) issued from inline asm: wave-uniform 64-bit base in SGPRs
// + one 32-bit lane offset.  Why not __builtin_amdgcn_global_load_lds: hipcc's waitcnt insertion treats
// an in-flight LDS-DMA as a FLAT-like event and from then on waits lgkmcnt(0) — i.e. for the fragment
// loads issued two instructions earlier — before every MFMA group.  The asm form is invisible to that
// pass; its completion is waited for by hand (dma_drain) right before the barrier that publishes the
// chunk.  OFF (0..3072) is added to BOTH the global and the LDS address, so one (base, M0) pair serves
// four consecutive 1 KiB pieces.  M0 is written in the same statement that consumes it (hipcc reserves
// M0 and uses it nowhere else in this kernel).
template <int OFF>
__device__ __forceinline__ void glds16_asm(const char* base_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:%3"
      :
      : "v"(lane_off), "s"(base_uniform), "s"(lds_dst_uniform), "i"(OFF)
      : "memory");
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct Loader {
  const float* stream;   // packed blob viewed as 32-bit words
  int q;                 // chunk being CONSUMED
  int q_end;
  bool skip_final;       // sigma_only launches jump from L8 straight to the density-head chunk
  int wave, lane;
  // descriptor of chunk q+1 (what the current chunk's body is prefetching).  Each wave moves a contiguous
  // quarter of the chunk (8..11 pieces): wave-uniform byte address of its first piece (scalar registers),
  // the matching LDS byte address, its piece count, + the lane's 16-byte offset (one 32-bit VGPR)
  const char* next_base;
  unsigned next_lds;
  int next_count;
  unsigned lane_off;
};

__device__ __forceinline__ void loader_prepare_next(Loader& ld, const float* next_slot) {
  int qn = ld.q + 1;
  if (ld.skip_final && qn == kChunkFinal0) qn = kChunkSigma;
  // nothing follows the last chunk: re-fetch chunk 0 into the idle slot so that the first eight DMA issues
  // of every chunk stay unconditional (branch-free k-steps; 32 KiB of dead traffic per tile)
  int pieces = 32, piece0 = 0;
  if (qn < ld.q_end) {
    const Chunk c = chunk_info(qn);
    pieces = chunk_pieces(c.steps, c.nnb);
    piece0 = c.piece0;
  }
  const int first = (ld.wave * pieces) >> 2;
  ld.next_count = (((ld.wave + 1) * pieces) >> 2) - first;
  ld.next_base = reinterpret_cast<const char*>(ld.stream) + (size_t)(piece0 + first) * 1024;
  ld.next_lds = lds_addr(next_slot) + (unsigned)first * 1024u;
}

// issue this wave's DMA piece number i of the next chunk (no-op past its end)
__device__ __forceinline__ void loader_issue(const Loader& ld, int i) {
#ifdef NSR_ABL_NO_DMA
  return;
#endif
  // every wave owns at least 8 pieces of every chunk: the first eight issues need no bounds test
  if (i < 8 || i < ld.next_count) {
    const char* base = ld.next_base + (i >> 2) * 4096;
    const unsigned dst = ld.next_lds + (unsigned)(i >> 2) * 4096u;
    switch (i & 3) {
      case 0: glds16_asm<0>(base, ld.lane_off, dst); break;
      case 1: glds16_asm<1024>(base, ld.lane_off, dst); break;
      case 2: glds16_asm<2048>(base, ld.lane_off, dst); break;
      default: glds16_asm<3072>(base, ld.lane_off, dst); break;
    }
  }
}

__device__ __forceinline__ h8 as_h8(const u32x4& v) { return __builtin_bit_cast(h8, v); }

struct Acc {
  f32x16 m;   // bias + sum (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi), fp32.  (A separate accumulator for the
              // two correction terms was measured: it removes the dependent MFMA issue (-5 % on a bare
              // MFMA+LDS loop) but its 16 extra adds per block cost more in the full kernel.)
};

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>)
template <int... S, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, S...>, F&& f) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// A-fragment load the compiler does not track: hipcc's own waitcnt insertion drains lgkmcnt(0) every
// PF steps (stalling on loads issued two instructions earlier); these are waited for by COUNT below.
template <int OFF>
__device__ __forceinline__ void lds_read16_async(u32x4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// wait until at most N younger LDS operations are outstanding; names the registers so that every
// consumer is ordered behind the wait (cdna_hip_programming.md §5.7, form ii)
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N));
}

// NSTEP k-steps of one output block.  A fragments are software-pipelined PF steps ahead of the
// MFMAs that consume them; sched_barrier pins the order.  a_addr is the lane's LDS byte address of
// the block's first piece; b_of(s, part) yields the step's activation operands, hook(s) runs in the
// MFMA shadow of step s.
template <int NSTEP, class BOf, class Hook>
__device__ __forceinline__ void block_mma(Acc& acc, const float* a_ptr, BOf&& b_of, Hook&& hook) {
  constexpr int PF = 3;
  u32x4 ah[NSTEP], al[NSTEP];
  const u32x4* a_pieces = reinterpret_cast<const u32x4*>(a_ptr);
#ifdef NSR_ABL_NO_LDSREAD
  u32x4 fake = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  asm volatile("" : "+v"(fake));
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) { ah[s] = fake; al[s] = fake; }
  (void)a_pieces;
#else
#pragma unroll
  for (int s = 0; s < PF && s < NSTEP; ++s) {
    ah[s] = a_pieces[(2 * s) * 64];
    al[s] = a_pieces[(2 * s + 1) * 64];
  }
#endif
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
#ifndef NSR_ABL_NO_LDSREAD
    if (s + PF < NSTEP) {
      ah[s + PF] = a_pieces[(2 * (s + PF)) * 64];
      al[s + PF] = a_pieces[(2 * (s + PF) + 1) * 64];
    }
#endif
    const u32x4 bh = b_of(s, 0), bl = b_of(s, 1);
#ifndef NSR_ABL_NO_MFMA
    // a_lo first: it is the younger of the step's two fragment loads, so ONE lgkmcnt wait serves all three
    acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al[s]), as_h8(bh), acc.m, 0, 0, 0);
    acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bl), acc.m, 0, 0, 0);
    acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bh), acc.m, 0, 0, 0);
#else
    acc.m[0] += __builtin_bit_cast(float, ah[s][0] ^ bh[0] ^ al[s][1] ^ bl[1]);
#endif
    hook(s);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// accumulator init from the chunk's bias piece (fp32, D-fragment order)
__device__ __forceinline__ void init_acc(Acc& a, const float* bias32, int h) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias32 + 8 * qd + 4 * h);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.m[4 * qd + i] = b4[i];
  }
}

// v -> (hi, lo) fp16 pairs packed two per 32-bit register (round toward zero for hi; lo takes the rest)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const auto ph = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  const auto pl = __builtin_amdgcn_cvt_pkrtz(x0 - (float)ph[0], x1 - (float)ph[1]);
  hi = __builtin_bit_cast(unsigned, ph);
  lo = __builtin_bit_cast(unsigned, pl);
}

// Pair P (accumulator registers 2P, 2P+1) of the pending block -> operand registers of the consuming
// layer: block nb becomes k-steps 2nb (P < 4 -> h0/l0) and 2nb+1 (P >= 4 -> h1/l1), element P & 3.
// Done in two halves so that no k-step carries more than ~5 extra VALU issues (a wave has ~5 free
// issue slots per MFMA): A = activation + hi, B = lo.
struct PairTmp {
  float x0, x1;
  unsigned hi;
};
template <int P>
__device__ __forceinline__ void pair_half_a(const Acc& p, float lower, PairTmp& t, u32x4& h0, u32x4& h1) {
  t.x0 = fmaxf(p.m[2 * P], lower);
  t.x1 = fmaxf(p.m[2 * P + 1], lower);
#ifdef NSR_ABL_NO_CONVERT
  t.hi = __float_as_uint(t.x0);
#else
  t.hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t.x0, t.x1));
#endif
  // pin the work HERE (MFMA shadow of the current k-step); LLVM would otherwise sink it to its first use,
  // i.e. serialise all eight blocks' conversions at the layer end
  asm volatile("" : "+v"(t.hi), "+v"(t.x0), "+v"(t.x1));
  if (P < 4) h0[P & 3] = t.hi; else h1[P & 3] = t.hi;
}
template <int P>
__device__ __forceinline__ void pair_half_b(const PairTmp& t, u32x4& l0, u32x4& l1) {
#ifdef NSR_ABL_NO_CONVERT
  unsigned lo = __float_as_uint(t.x1);
#else
  const auto ph = __builtin_bit_cast(decltype(__builtin_amdgcn_cvt_pkrtz(0.f, 0.f)), t.hi);
  unsigned lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t.x0 - (float)ph[0], t.x1 - (float)ph[1]));
#endif
  asm volatile("" : "+v"(lo));
  if (P < 4) l0[P & 3] = lo; else l1[P & 3] = lo;
}
// half-pair number hp (0..15) of the pending block
__device__ __forceinline__ void pending_half(int hp, const Acc& p, float lower, PairTmp& t, u32x4& h0, u32x4& l0,
                                             u32x4& h1, u32x4& l1) {
  switch (hp) {
    case 0: pair_half_a<0>(p, lower, t, h0, h1); break;
    case 1: pair_half_b<0>(t, l0, l1); break;
    case 2: pair_half_a<1>(p, lower, t, h0, h1); break;
    case 3: pair_half_b<1>(t, l0, l1); break;
    case 4: pair_half_a<2>(p, lower, t, h0, h1); break;
    case 5: pair_half_b<2>(t, l0, l1); break;
    case 6: pair_half_a<3>(p, lower, t, h0, h1); break;
    case 7: pair_half_b<3>(t, l0, l1); break;
    case 8: pair_half_a<4>(p, lower, t, h0, h1); break;
    case 9: pair_half_b<4>(t, l0, l1); break;
    case 10: pair_half_a<5>(p, lower, t, h0, h1); break;
    case 11: pair_half_b<5>(t, l0, l1); break;
    case 12: pair_half_a<6>(p, lower, t, h0, h1); break;
    case 13: pair_half_b<6>(t, l0, l1); break;
    case 14: pair_half_a<7>(p, lower, t, h0, h1); break;
    case 15: pair_half_b<7>(t, l0, l1); break;
    default: break;
  }
}
// Schedule over the k-steps of a 16-step chunk: 16 halves in k-steps 0..13 (k-steps 0 and 7 take two), so
// that even the operands of k-steps 14, 15 (block 7 of the previous layer) are ready before they are used.
__device__ __forceinline__ void pending_step(int s, const Acc& p, float lower, PairTmp& t, u32x4& h0, u32x4& l0,
                                             u32x4& h1, u32x4& l1) {
  if (s == 0) { pending_half(0, p, lower, t, h0, l0, h1, l1); pending_half(1, p, lower, t, h0, l0, h1, l1); }
  else if (s < 7) pending_half(s + 1, p, lower, t, h0, l0, h1, l1);
  else if (s == 7) { pending_half(8, p, lower, t, h0, l0, h1, l1); pending_half(9, p, lower, t, h0, l0, h1, l1); }
  else if (s < 14) pending_half(s + 2, p, lower, t, h0, l0, h1, l1);
}
// colour head: pair P of a finished dir_encoding block (relu) dotted with the three rgb rows
template <int P>
__device__ __forceinline__ void pair_rgb(const Acc& p, const float* w32, int h, float (&rgb)[3]) {
  const float x0 = fmaxf(p.m[2 * P], 0.0f);
  const float x1 = fmaxf(p.m[2 * P + 1], 0.0f);
  constexpr int r = 2 * P;                       // registers r, r+1 <-> features 8*(r>>2) + 4h + (r&3), +1
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float2 w2 = *reinterpret_cast<const float2*>(w32 + 128 * k + 8 * (r >> 2) + 4 * h + (r & 3));
    rgb[k] = fmaf(x1, w2.y, fmaf(x0, w2.x, rgb[k]));
  }
}
template <int S0>
__device__ __forceinline__ void rgb_step(int s, const Acc& p, const float* w32, int h, float (&rgb)[3]) {
  if (s == S0 + 0) pair_rgb<0>(p, w32, h, rgb);
  if (s == S0 + 1) pair_rgb<1>(p, w32, h, rgb);
  if (s == S0 + 2) pair_rgb<2>(p, w32, h, rgb);
  if (s == S0 + 3) pair_rgb<3>(p, w32, h, rgb);
  if (s == S0 + 4) pair_rgb<4>(p, w32, h, rgb);
  if (s == S0 + 5) pair_rgb<5>(p, w32, h, rgb);
  if (s == S0 + 6) pair_rgb<6>(p, w32, h, rgb);
  if (s == S0 + 7) pair_rgb<7>(p, w32, h, rgb);
}

// weight DMA of the next chunk: one piece per k-step over the first 11 k-steps of the chunk
__device__ __forceinline__ void dma_step(const Loader& ld, int s) {
  if (s < 11) loader_issue(ld, s);
}

constexpr int kConvStep0 = 6;   // pending block is converted in k-steps 6..13 (one register pair each)

// One 256 -> 256 trunk layer L (1..8; 8 = xyz_encoding_final): in (bh, bl) -> out (oh, ol).
// L == 4 prepends the 4 positional-encoding k-steps (skip connection).  `pend` is the block that
// finished last (block 7 of the previous layer on entry; block 7 of this layer on exit): it is
// summed, activated and re-split in the MFMA shadow of the FOLLOWING block's k-steps 1..4.
__device__ __forceinline__ void trunk_layer(int L, u32x4 (&bh)[16], u32x4 (&bl)[16], u32x4 (&oh)[16], u32x4 (&ol)[16],
                                            const u32x4* stash, Loader& ld, float* ring, int h, Acc& pend) {
  const float lower = (L < 8) ? 0.0f : -__builtin_inff();   // relu on L1..L8, none on xyz_encoding_final
  const int skip = (L == 4) ? 8 : 0;                        // pieces taken by the pe k-steps
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    float* slot = ring + (nb & 1) * kSlotFloats;
    float* next_slot = ring + ((nb + 1) & 1) * kSlotFloats;
    NSR_SYNC();                            // chunk ld.q landed (vmcnt(0)) and the other slot is free
    loader_prepare_next(ld, next_slot);
    const float* a0 = slot + ld.lane * 4;
    Acc cur;
    PairTmp ptmp;
    init_acc(cur, slot + (32 + skip) * 256, h);   // bias piece follows the 2*steps weight pieces
    if (L == 4) {
      // skip connection: the encoded position was parked in LDS by the prologue (8 fragments per lane)
      u32x4 pe8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pe8[i] = stash[i * 64];
      block_mma<4>(cur, a0, [&](int s, int part) -> u32x4 { return pe8[4 * part + s]; }, [&](int) {});
    }
    block_mma<16>(
        cur, a0 + skip * 256, [&](int s, int part) -> u32x4 { return part ? bl[s] : bh[s]; },
        [&](int s) {
          dma_step(ld, s);
          if (nb == 0)
            // block 7 of the previous layer (always relu'd: the previous layer is L1..L7) -> k-steps 14, 15
            // of THIS layer's input, needed only at the end of this chunk
            pending_step(s, pend, 0.0f, ptmp, bh[14], bl[14], bh[15], bl[15]);
          else
            pending_step(s, pend, lower, ptmp, oh[2 * nb - 2], ol[2 * nb - 2], oh[2 * nb - 1], ol[2 * nb - 1]);
        });
    pend = cur;
    ld.q += 1;
  }
}

template <int MODE, bool SIGMA_ONLY, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_f16x3_kernel(const float* __restrict__ packed, const float* __restrict__ x, const float* __restrict__ zv,
                 int64_t P, int N, float* __restrict__ out) {
  // 2 x 41 KiB weight ring + per-wave stash of the encoded inputs (12 fragments x 64 lanes x 16 B = 12 KiB)
  // + the colour-head block (rgb weights and bias, 448 floats), so the loop issues no global loads
  __shared__ __attribute__((aligned(16))) float ring[2 * kSlotFloats + 4 * 12 * 256 + hx::kAuxFloats];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const float* aux = ring + 2 * kSlotFloats + 4 * 12 * 256;   // LDS copy, visible after the first barrier
  for (int i = threadIdx.x; i < hx::kAuxFloats; i += 256)
    ring[2 * kSlotFloats + 4 * 12 * 256 + i] = packed[kPiecesTotal * 256 + i];

  Loader ld;
  ld.stream = packed;
  ld.q = -1;
  ld.q_end = SIGMA_ONLY ? kChunkSigma + 1 : kChunks;
  ld.skip_final = SIGMA_ONLY;
  ld.wave = wave;
  ld.lane = lane;
  ld.lane_off = (unsigned)lane * 16u;
  loader_prepare_next(ld, ring);           // chunk 0
#pragma unroll
  for (int i = 0; i < 11; ++i) loader_issue(ld, i);
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
    const int64_t ray = pc / ((NS > 0) ? NS : N);
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

  // park the split encodings in LDS: L5 (skip) and dir_encoding re-read them, which frees 48 registers
  u32x4* stash = reinterpret_cast<u32x4*>(ring + 2 * kSlotFloats) + wave * 12 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    stash[s * 64] = peh[s];
    stash[(4 + s) * 64] = pel[s];
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    stash[(8 + s) * 64] = deh[s];
    stash[(10 + s) * 64] = del[s];
  }

  u32x4 bh[16], bl[16], oh[16], ol[16];
  Acc pend;

  // ---- L1: two chunks of four output blocks, 4 k-steps each; block b is re-split during block b+1
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float* slot = ring + (c & 1) * kSlotFloats;
    float* next_slot = ring + ((c + 1) & 1) * kSlotFloats;
    NSR_SYNC();
    loader_prepare_next(ld, next_slot);
    const float* a0 = slot + lane * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb = 4 * c + g;
      Acc cur;
      PairTmp ptmp;
      init_acc(cur, slot + 32 * 256 + 32 * g, h);
      block_mma<4>(
          cur, a0 + g * 8 * 256, [&](int s, int part) -> u32x4 { return part ? pel[s] : peh[s]; },
          [&](int s) {
            dma_step(ld, 4 * g + s);
            if (nb > 0) {   // four halves per k-step: the block has only four k-steps
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4)
                pending_half(4 * s + q4, pend, 0.0f, ptmp, bh[2 * nb - 2], bl[2 * nb - 2], bh[2 * nb - 1], bl[2 * nb - 1]);
            }
          });
      pend = cur;
    }
    ld.q += 1;
  }

  // ---- L2..L8 (+ xyz_encoding_final), two layers per trip so the register sets swap roles
  constexpr int kPairs = SIGMA_ONLY ? 3 : 4;
#pragma unroll 1
  for (int pair = 0; pair < kPairs; ++pair) {
    const int L = 1 + 2 * pair;
    trunk_layer(L, bh, bl, oh, ol, stash, ld, ring, h, pend);
    trunk_layer(L + 1, oh, ol, bh, bl, stash, ld, ring, h, pend);
  }
  if (SIGMA_ONLY) {
    trunk_layer(7, bh, bl, oh, ol, stash, ld, ring, h, pend);
    ld.q = kChunkSigma;      // xyz_encoding_final is not evaluated
  }

  // ---- density head: sigma.weight as row 0 of one more 32-row block over h8 (= oh/ol: the input of
  // xyz_encoding_final, still intact).  The pending block is xyz_encoding_final's last one (-> bh, no
  // activation), or L8's last one in a sigma_only launch (-> oh, relu).
  float sigma;
  {
    float* slot = ring;                      // 58 / 66 chunks so far: slot parity 0
    float* next_slot = ring + kSlotFloats;
    NSR_SYNC();
    loader_prepare_next(ld, next_slot);
    const float* a0 = slot + lane * 4;
    Acc cur;
    PairTmp ptmp;
    init_acc(cur, slot + 32 * 256, h);
    block_mma<16>(
        cur, a0, [&](int s, int part) -> u32x4 { return part ? ol[s] : oh[s]; },
        [&](int s) {
          dma_step(ld, s);
          if (SIGMA_ONLY)
            pending_step(s, pend, 0.0f, ptmp, oh[14], ol[14], oh[15], ol[15]);
          else
            pending_step(s, pend, -__builtin_inff(), ptmp, bh[14], bl[14], bh[15], bl[15]);
        });
    sigma = cur.m[0];                        // row 0 of the block lives in register 0 of the h == 0 lanes
    ld.q += 1;
  }
  if (SIGMA_ONLY) {
    if (h == 0 && p < P) out[p] = sigma;
    return;
  }

  // ---- dir_encoding (cat([g, de]) -> 128, relu) fused with the rgb head (128 -> 3, sigmoid)
  float rgb[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    float* slot = ring + ((nb + 1) & 1) * kSlotFloats;     // 67 chunks precede dir_encoding
    float* next_slot = ring + (nb & 1) * kSlotFloats;
    NSR_SYNC();
    loader_prepare_next(ld, next_slot);
    const float* a0 = slot + lane * 4;
    Acc cur;
    init_acc(cur, slot + 36 * 256, h);
    u32x4 de4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) de4[i] = stash[(8 + i) * 64];
    block_mma<18>(
        cur, a0,
        [&](int s, int part) -> u32x4 { return (s < 16) ? (part ? bl[s & 15] : bh[s & 15]) : de4[2 * part + (s & 1)]; },
        [&](int s) {
          dma_step(ld, s);
          if (nb > 0) rgb_step<kConvStep0>(s, pend, aux + hx::kAuxRgbW + 32 * (nb - 1), h, rgb);
        });
    pend = cur;
    ld.q += 1;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) rgb_step<0>(s, pend, aux + hx::kAuxRgbW + 32 * 3, h, rgb);
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
