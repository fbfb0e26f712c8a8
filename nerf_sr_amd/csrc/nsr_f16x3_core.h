// Building blocks shared by the split-fp16 ("f16x3") chain kernels: the inference kernel (nsr_mlp_f16.hip) and the
// training chains (nsr_train_chain.hip).  Weight stream by LDS-DMA through a 3-slot ring, one 32-feature output block
// per chunk, fragments software-pipelined ahead of the MFMAs.  See nsr_mlp_f16.hip for the arithmetic.
#pragma once
#include "nsr_common.h"
#include "nsr_mlp_layout.h"
#include "nsr_panels.h"

using namespace nsr;
using namespace nsr::hx;

// NSR_ABL_*: ablation switches for the measurement ladder in profiles/r1_f16x3_pmc.txt (scripts/ablate.sh builds
// variant libraries with them); never defined in the product build.
#ifdef NSR_ABL_NO_BARRIER
#define NSR_SYNC() ((void)0)
#else
#define NSR_SYNC() do { dma_drain(); __syncthreads(); } while (0)
#endif

// weights and biases enter the stream multiplied by 2^6 (see the header); |w| < 1023 stays inside fp16
constexpr float kWScale = 64.0f, kWInvScale = 1.0f / 64.0f;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hl(float a, float b, int part) {
  _Float16 ha = (_Float16)a, hb = (_Float16)b;               // round to nearest even
  if (part) {
    ha = (_Float16)(a - (float)ha);
    hb = (_Float16)(b - (float)hb);
  }
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
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
// The base is copied into a scratch SGPR pair INSIDE the statement: when hipcc has spilled the descriptor to VGPR lanes
// it restores it with v_readlane right in front of the statement, and a VMEM instruction that reads a VALU-written SGPR
// needs five wait states which hipcc's hazard pass cannot see through inline asm (round 4: a persistent-loop build read
// a stale pair and faulted).  An SALU read is interlocked, and the copy doubles as the wait state after the M0 write.
// M0 is a reserved register for hipcc (it never allocates it: naming it in a clobber list is a warning), so the pieces
// 4q+1 .. 4q+3 of a 4 KiB group reuse the M0 piece 4q wrote (glds16_keep_asm).
template <int OFF>
__device__ __forceinline__ void glds16_asm(const char* base_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
  unsigned long long tmp;
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_mov_b64 %0, %2\n\t"
      "global_load_lds_dwordx4 %1, %0 offset:%4"
      : "=&s"(tmp)
      : "v"(lane_off), "s"(base_uniform), "s"(lds_dst_uniform), "i"(OFF)
      : "memory");
}
// a further piece of the 4 KiB group whose first piece set M0 (same base, same M0)
template <int OFF>
__device__ __forceinline__ void glds16_keep_asm(const char* base_uniform, unsigned lane_off) {
  unsigned long long tmp;
  asm volatile(
      "s_mov_b64 %0, %2\n\t"
      "global_load_lds_dwordx4 %1, %0 offset:%3"
      : "=&s"(tmp)
      : "v"(lane_off), "s"(base_uniform), "i"(OFF)
      : "memory");
}
// two consecutive pieces (OFF, OFF + 1024) with ONE M0 write
template <int OFF>
__device__ __forceinline__ void glds16x2_asm(const char* base_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
  unsigned long long tmp;
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_mov_b64 %0, %2\n\t"
      "global_load_lds_dwordx4 %1, %0 offset:%4\n\t"
      "global_load_lds_dwordx4 %1, %0 offset:%5"
      : "=&s"(tmp)
      : "v"(lane_off), "s"(base_uniform), "s"(lds_dst_uniform), "i"(OFF), "i"(OFF + 1024)
      : "memory");
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// vector memory operations complete in issue order: with N younger operations (loads / stores issued AFTER the last DMA
// piece that must have landed) allowed to stay in flight
template <int N>
__device__ __forceinline__ void dma_drain_but() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// ---------------------------------------------------------------------------
// weight-stream loader: 3-slot LDS ring, the chunk at sequence position j lives in slot j % 3.
//
// Protocol (per wave, chunk j being consumed):
//   * entering chunk j: its data is PUBLISHED (every wave's DMA landed and a barrier passed), chunk
//     j+1's DMA has been issued by every wave, and the first k-steps' fragments + bias of chunk j are
//     already in registers (prefetched at the end of chunk j-1) — no bubble at the chunk boundary;
//   * at k-step kBar of chunk j: wait for the own DMA of chunk j+1 (issued half a chunk ago), barrier
//     => chunk j+1 is published and every wave has left chunk j-1, whose slot is therefore free;
//   * k-steps kBar..kBar+5: issue the DMA of chunk j+2 into that slot (two pieces per k-step);
//   * last three k-steps: prefetch chunk j+1's first fragments and bias.
// ---------------------------------------------------------------------------
constexpr int kSlotBytes = kSlotFloats * 4;
constexpr int kBar = 8;

// a chunk of the stream as this wave sees it: where it starts, how many 1 KiB pieces it has, and the
// contiguous quarter of it this wave moves (all wave-uniform, SGPRs)
struct ChunkRef {
  int piece0, pieces;
  int first, count;
};
__device__ __forceinline__ ChunkRef make_ref(int piece0, int pieces, int wave) {
  ChunkRef c;
  c.piece0 = piece0;
  c.pieces = pieces;
  c.first = (wave * pieces) >> 2;
  c.count = (((wave + 1) * pieces) >> 2) - c.first;
  return c;
}

struct Loader {
  const float* stream;   // packed blob viewed as 32-bit words
  int wave;
  unsigned lane_off;     // lane * 16
  unsigned slot_cur, slot_next, slot_free;   // LDS byte addresses of the slots of chunks j, j+1, j+2
  // DMA descriptor of the chunk being fetched: wave-uniform byte address of this wave's first piece, the
  // matching LDS byte address, its piece count
  const char* dma_base;
  unsigned dma_lds;
  int dma_count;
#ifdef NSR_ABL_TIMELINE
  unsigned long long* tk = nullptr;   // k-step stamps of ONE designated chunk (null elsewhere; compile-time after inlining)
#endif
};
#ifdef NSR_ABL_TIMELINE
__device__ __forceinline__ unsigned long long tl_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#endif

__device__ __forceinline__ void loader_prepare_dma(Loader& ld, const ChunkRef& c, unsigned slot_lds) {
  ld.dma_count = c.count;
  ld.dma_base = reinterpret_cast<const char*>(ld.stream) + ((unsigned)(c.piece0 + c.first) << 10);   // the stream is < 4 MiB
  ld.dma_lds = slot_lds + (unsigned)c.first * 1024u;
}

// issue this wave's DMA piece number i of the chunk being fetched (no-op past its end)
__device__ __forceinline__ void loader_issue(const Loader& ld, int i) {
#ifdef NSR_ABL_NO_DMA
  return;
#endif
  if (i < 8 || i < ld.dma_count) {   // every wave owns at least 8 pieces of every chunk
    const char* base = ld.dma_base + (i >> 2) * 4096;
    const unsigned dst = ld.dma_lds + (unsigned)(i >> 2) * 4096u;
    // pieces are issued in order, piece 4q first: it writes M0 for its group
    switch (i & 3) {
      case 0: glds16_asm<0>(base, ld.lane_off, dst); break;
      case 1: glds16_keep_asm<1024>(base, ld.lane_off); break;
      case 2: glds16_keep_asm<2048>(base, ld.lane_off); break;
      default: glds16_keep_asm<3072>(base, ld.lane_off); break;
    }
  }
}

// pieces 2j, 2j + 1 (j = 0..3: always present) with one M0 write
__device__ __forceinline__ void loader_issue2(const Loader& ld, int j) {
#ifdef NSR_ABL_NO_DMA
  return;
#endif
  const char* base = ld.dma_base + (j >> 1) * 4096;
  const unsigned dst = ld.dma_lds + (unsigned)(j >> 1) * 4096u;
  if (j & 1) glds16x2_asm<2048>(base, ld.lane_off, dst);
  else glds16x2_asm<0>(base, ld.lane_off, dst);
}

__device__ __forceinline__ void loader_advance(Loader& ld) {
  const unsigned t = ld.slot_cur;
  ld.slot_cur = ld.slot_next;
  ld.slot_next = ld.slot_free;
  ld.slot_free = t;
}

// the publish point of chunk j+1 (see protocol above), then start fetching chunk j+2
// YOUNGER: vector-memory operations known to have been issued AFTER the last DMA piece this publish point waits for
// (they may stay in flight); strict: ... unless this call site cannot vouch for them (runtime, wave-uniform)
template <int YOUNGER = 0, bool PREPARE = true>
__device__ __forceinline__ void loader_publish(Loader& ld, const ChunkRef& c2, bool strict = false) {
#ifndef NSR_ABL_NO_DRAIN
  if (YOUNGER > 0 && strict) dma_drain();
  else dma_drain_but<YOUNGER>();
#endif
#ifndef NSR_ABL_NO_BARRIER
#ifndef NSR_ABL_FENCED_BARRIER
  // s_barrier alone: what the publish point orders is (a) every wave's DMA pieces of chunk j+1 (each wave has just waited
  // for its own: vmcnt) and (b) every wave's READS of chunk j-1, which were consumed by MFMAs a whole chunk ago.  The
  // LDS reads in flight here are fragment prefetches of the CURRENT chunk, which nobody overwrites: __syncthreads()'s
  // lgkmcnt(0) would only stall the wave on them.
  asm volatile("s_barrier" ::: "memory");
#else
  __syncthreads();
#endif
#endif
  if (PREPARE) loader_prepare_dma(ld, c2, ld.slot_free);
}

__device__ __forceinline__ h8 as_h8(const u32x4& v) { return __builtin_bit_cast(h8, v); }
__device__ __forceinline__ const u32x4* lds_vec(unsigned byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const u32x4*)(const __attribute__((address_space(3))) u32x4*)(size_t)byte_addr;
#else
  (void)byte_addr;
  return nullptr;   // host pass of the single-source compile; never executed
#endif
}

struct Acc {
  f32x16 m;   // bias + sum (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi), fp32.  (A separate accumulator for the
              // two correction terms was measured: it removes the dependent MFMA issue (-5 % on a bare
              // MFMA+LDS loop) but its 16 extra adds per block cost more in the full kernel.)
};

// what a k-step sequence needs before its first MFMA: the A fragments of its first kPF k-steps (and, for
// a new chunk, the accumulator init = bias).  Filled during the last k-steps of the preceding sequence.
#ifndef NSR_F16X3_KPF
#define NSR_F16X3_KPF 3
#endif
constexpr int kPF = NSR_F16X3_KPF;
struct Pre {
  u32x4 ah[kPF], al[kPF];
  f32x16 bias;
};

// fragments of k-step k (< kPF) of the sequence whose pieces start at LDS byte address `seq_addr` (+ lane*16)
__device__ __forceinline__ void prefetch_frag(Pre& pre, int k, unsigned seq_addr) {
  const u32x4* a = lds_vec(seq_addr);
  pre.ah[k] = a[(2 * k) * 64];
  pre.al[k] = a[(2 * k + 1) * 64];
}
// accumulator init of a chunk from its bias piece (fp32, D-fragment order; broadcast reads)
__device__ __forceinline__ void prefetch_bias(Pre& pre, unsigned bias_addr, int h) {
  const f32x4* b = reinterpret_cast<const f32x4*>(lds_vec(bias_addr + 16u * (unsigned)h));
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const f32x4 b4 = b[2 * qd];                    // floats 8*qd + 4*h .. +3
#pragma unroll
    for (int i = 0; i < 4; ++i) pre.bias[4 * qd + i] = b4[i];
  }
}

// NSTEP k-steps of one output block, fragments software-pipelined kPF k-steps ahead (the first kPF come
// in through `pre`), order pinned by sched_barrier.  a_addr: LDS byte address (+ lane*16) of the
// sequence's first piece.  b_of(s, part) yields the k-step's activation operands; hook(s) runs in the MFMA
// shadow of k-step s; BAR >= 0 places the chunk's publish point + DMA issue (k-steps BAR..BAR+5);
// next(k), k = 0..2, runs in the last three k-steps and prefetches the following sequence into `nxt`.
template <int NSTEP, int BAR, int YOUNGER = 0, class BOf, class Hook, class Next>
__device__ __forceinline__ void block_mma(Acc& acc, const Pre& pre, unsigned a_addr, Loader& ld, const ChunkRef& c2,
                                          BOf&& b_of, Hook&& hook, Next&& next, bool strict = false) {
  static_assert(NSTEP >= kPF, "sequence shorter than the prefetch depth");
  const u32x4* a_pieces = lds_vec(a_addr);
  u32x4 ah[NSTEP], al[NSTEP];
#pragma unroll
  for (int s = 0; s < kPF; ++s) {
    ah[s] = pre.ah[s];
    al[s] = pre.al[s];
  }
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
#ifdef NSR_ABL_TIMELINE
    if (ld.tk && NSTEP == 16 && (s == 0 || s == 4 || s == 8 || s == 11 || s == 14)) ld.tk[s == 0 ? 0 : (s == 4 ? 1 : (s == 8 ? 2 : (s == 11 ? 3 : 4)))] = tl_now();
#endif
    if (s == BAR) loader_publish<YOUNGER>(ld, c2, strict);
    if (s + kPF < NSTEP) {
      ah[s + kPF] = a_pieces[(2 * (s + kPF)) * 64];
      al[s + kPF] = a_pieces[(2 * (s + kPF) + 1) * 64];
    } else {
      next(s + kPF - NSTEP);
    }
    const u32x4 bh = b_of(s, 0), bl = b_of(s, 1);
    // a_lo first: it is the younger of the step's two fragment loads, so ONE lgkmcnt wait serves all three
    acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al[s]), as_h8(bh), acc.m, 0, 0, 0);
    acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bl), acc.m, 0, 0, 0);
    acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bh), acc.m, 0, 0, 0);
    hook(s);
    // DMA of chunk j+2: two pieces per k-step over six k-steps.  (Measured alternatives: bursts of four pieces
    // sharing one M0 write -7 %: back-to-back LDS-DMA issues hold the wave longer than one MFMA; one piece per
    // k-step from an earlier publish point (k-step 5) -3 %.)
    if (BAR >= 0 && s >= BAR && s < BAR + 6) {
      if (s - BAR < 4) {
        loader_issue2(ld, s - BAR);   // pieces 0..7: every wave owns them, one M0 write per pair
      } else {
        loader_issue(ld, 2 * (s - BAR));
        loader_issue(ld, 2 * (s - BAR) + 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Gap-aware form of block_mma (round 4; the inference kernel uses it everywhere).  A wave alone on its SIMD hides at most
// ~5 single-issue instructions behind one 32-cycle MFMA, and only if they sit in THAT MFMA's gap: the k-step-granular form
// above lets hipcc put all of a k-step's fillers (2 fragment reads + 7 re-split VALU + 2 DMA pieces and their SALU) behind
// the k-step's FIRST MFMA, and the measured price was ~5 cycles for every filler beyond the fifth (profiles/
// r4_timeline_before_128.json: quiet k-steps 103 cycles, the double-half-step k-steps 122, the DMA k-steps 140-155, against
// 96 of matrix time).  Here every k-step is three fenced (MFMA + fillers) segments:
//   gap 0: a_lo*b_hi + the fragment reads of k-step s + kPF (or next(k, 0)) + hook(s, 0)
//   gap 1: a_hi*b_lo + hook(s, 1)
//   gap 2: a_hi*b_hi + hook(s, 2) + next(k, 2)
// and the chunk's DMA pieces go one per gap (gaps 1 and 2 of k-steps BAR..BAR+5, each with its own M0 write).
// DMA = false: the publish point stays, but no chunk is fetched behind it (the last two chunks of a one-tile workgroup:
// there is no next tile to fetch L1 for).
// MMA = false (ablation builds only, -DNSR_ABL_NO_DENSITY_MMA): the block runs without its MFMAs -- what they cost.
template <int NSTEP, int BAR, int YOUNGER = 0, bool DMA = true, bool MMA = true, class BOf, class Hook, class Next>
__device__ __forceinline__ void block_mma3(Acc& acc, const Pre& pre, unsigned a_addr, Loader& ld, const ChunkRef& c2,
                                           BOf&& b_of, Hook&& hook, Next&& next, bool strict = false) {
  static_assert(NSTEP >= kPF, "sequence shorter than the prefetch depth");
  const u32x4* a_pieces = lds_vec(a_addr);
  u32x4 ah[NSTEP], al[NSTEP];
#pragma unroll
  for (int s = 0; s < kPF; ++s) {
    ah[s] = pre.ah[s];
    al[s] = pre.al[s];
  }
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
#ifdef NSR_ABL_TIMELINE
    if (ld.tk && NSTEP == 16 && (s == 0 || s == 4 || s == 8 || s == 11 || s == 14)) ld.tk[s == 0 ? 0 : (s == 4 ? 1 : (s == 8 ? 2 : (s == 11 ? 3 : 4)))] = tl_now();
#endif
    // the DMA descriptor of chunk j+2 (six SALU) is formed three k-steps ahead of the publish point, in a quiet gap: the
    // previous chunk's last piece went out at its k-step BAR+5, and slot_free does not change inside a chunk
    if (DMA && BAR >= 3 && s == BAR - 3) loader_prepare_dma(ld, c2, ld.slot_free);
    if (s == BAR) loader_publish<YOUNGER, (DMA && BAR < 3)>(ld, c2, strict);
    const u32x4 bh = b_of(s, 0), bl = b_of(s, 1);
    // ---- gap 0.  a_lo first: it is the younger of the step's two fragment loads, so ONE lgkmcnt wait serves all three
    if (MMA) acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(al[s]), as_h8(bh), acc.m, 0, 0, 0);
    if (s + kPF < NSTEP) {
      ah[s + kPF] = a_pieces[(2 * (s + kPF)) * 64];
      al[s + kPF] = a_pieces[(2 * (s + kPF) + 1) * 64];
    } else {
      next(s + kPF - NSTEP, 0);
    }
    hook(s, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- gap 1
    if (MMA) acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bl), acc.m, 0, 0, 0);
    hook(s, 1);
    if (DMA && BAR >= 0 && s >= BAR && s < BAR + 6) loader_issue(ld, 2 * (s - BAR));
    __builtin_amdgcn_sched_barrier(0);
    // ---- gap 2
    if (MMA) acc.m = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(ah[s]), as_h8(bh), acc.m, 0, 0, 0);
    hook(s, 2);
    if (s + kPF >= NSTEP) next(s + kPF - NSTEP, 2);
    if (DMA && BAR >= 0 && s >= BAR && s < BAR + 6) loader_issue(ld, 2 * (s - BAR) + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// v -> (hi, lo) fp16 pairs packed two per 32-bit register, both rounded to nearest (prologue only)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
  const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
  hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}

// ---------------------------------------------------------------------------
// Training panels (round 5: 2 bytes per value).  The chain kernels of the training step (the TRAIN instantiation of the
// inference kernel and nsr_train_chain.hip) keep, per layer, exactly what the weight-gradient contraction
//     dW_l = sum over the points of  dz_l (x) a_{l-1}
// needs of them: the fp16 `hi` operand registers they have already made for their own next layer -- the activation
// a = relu(z) (forward) and the masked input gradient x * phi (backward; phi = the point's power-of-two operand scale).
// One 32-feature output block of a wave's 32 points is two "units" of 1 KiB (64 lanes x 16 B): unit u = the packed pairs
// 4u .. 4u+3 of the block = the u32x4 that is k-step 2 nb + u of the consuming layer's B operand; lane (m, h) holds, for
// point m, the block's features 16u + 4h + {0..3} (first 8 bytes) and 16u + 8 + 4h + {0..3} (second 8 bytes).
// A unit leaves as ONE global_store_dwordx4 per lane (2 stores per block instead of the 16 dword stores of the fp32
// panels of rounds 2-4: the chain kernels were store-ISSUE bound), into slot ((2m + h) ^ 8u) of the unit -- a full,
// contiguous KiB per wave instruction, permuted so that the weight-gradient kernel's LDS image of it (a plain LDS-DMA copy)
// is read conflict-free by ds_read_b64_tr_b16 (nsr_wgrad_f16.hip: the transposing read turns [point][4 features] into the
// MFMA operand's [feature][4 points]; feature 16u + l of the block arrives in lane l of a 16-lane group, natural order).
// A panel set is twelve panels back to back, panel-major: 0..7 = trunk layers 1..8, 8 = xyz_encoding_final (no relu),
// 9 = dir_encoding (128 rows); forward sets only: 10 = the encoded position (64 rows), 11 = the encoded direction (64 rows,
// 32 of them zero); row order of the two encodings: enc_row() below.  Per sample point 2,560 rows x 2 B = 5 KiB forward,
// 2,432 x 2 B backward (rounds 2-4: 11.3 + 10.8 KB).
// ---------------------------------------------------------------------------
struct PanelRef {
  char* base;         // panel set
  int64_t n_groups;   // ceil(P / 128) * 4
  int64_t group;      // this wave's point group (wave-uniform)
  unsigned* sgn;      // sign panels (see below)
};
// first byte of output block `blk` of `panel` for this wave's points (its two units follow each other)
__device__ __forceinline__ const char* panel_block(const PanelRef& t, int panel, int blk) {
  return t.base + panel_offset_bytes(t.n_groups, panel) + (t.group * panel_rows(panel) + 32 * blk) * kPanelRowBytes;
}
// this lane's byte offset inside unit U (0 / 1) of a block: slot (2m + h) ^ 8U
__device__ __forceinline__ unsigned unit_voff(int m, int h, int u) { return 16u * (unsigned)((2 * m + h) ^ (8 * u)); }
// cache policy of the panel stores: write-once streams far larger than the caches, so non-temporal (round 2, same box,
// 2,048-ray training step: default policy 7.51-7.54 ms, "sc0 sc1" 7.38 ms, "nt" 6.50 ms)
#ifndef NSR_PANEL_STORE_POLICY
#define NSR_PANEL_STORE_POLICY " nt"
#endif
// unit U of the block at `blk`: v = the four packed pairs, voff = unit_voff(m, h, U)
template <int U>
__device__ __forceinline__ void unit_store(const u32x4& v, const char* blk, unsigned voff) {
#ifdef NSR_ABL_FWD_NO_STORE   // ablation (scripts/): how much of the TRAIN forward kernel is its panel writes
  if (voff != 0xffffffffu) return;
#endif
  // base through an in-statement SALU copy: see glds16_asm (VALU-restored SGPR -> VMEM hazard behind inline asm)
  unsigned long long tmp;
  asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0 offset:%4" NSR_PANEL_STORE_POLICY : "=&s"(tmp) : "v"(voff), "v"(v), "s"(blk), "n"(U * 1024) : "memory");
}

// Sign panels: one bit per pre-activation ([z < 0], i.e. "the ReLU zeroes it"; +0.0 counts as active) of the layers whose
// ReLU the backward chain masks with (panels 0..7 and 9; xyz_encoding_final has none) -- the only thing the backward chain
// needs from the forward pass.  Two consecutive blocks (2q, 2q + 1) of a panel share one dword per lane: lane (m, h) keeps
// the 16 sign bits of its 16 accumulator registers of block 2q in bits 31..16 (register r in bit 31 - r) and of block
// 2q + 1 in bits 15..0 (bit 15 - r) -- what 32 sign_push in block order leave behind.  Per point group 34 pair blocks (4 per
// 256-row panel, 2 for panel 9) of 64 dwords: 272 bytes per sample point (rounds 2-4: one half-used dword per block, 608).
constexpr int kSignPairs = 34;
__device__ __host__ __forceinline__ int64_t sign_panel_words(int64_t n_groups) { return n_groups * kSignPairs * 64; }
// the dword block that holds the signs of output block `blk` of `panel` (shared with block blk ^ 1)
__device__ __forceinline__ unsigned* sign_block(unsigned* base, int64_t group, int panel, int blk) {
  return base + (group * kSignPairs + (panel == 9 ? 32 : 4 * panel) + (blk >> 1)) * 64;
}
// bits = (bits << 1) | sign(v)
__device__ __forceinline__ void sign_push(unsigned& bits, float v) {
  asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(bits) : "v"(v));
}
__device__ __forceinline__ void sign_store(unsigned bits, const unsigned* blk, unsigned lane4) {
  unsigned long long tmp;
  asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0" : "=&s"(tmp) : "v"(lane4), "v"(bits), "s"(blk) : "memory");
}
