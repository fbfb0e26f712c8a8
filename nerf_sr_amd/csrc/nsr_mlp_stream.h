// Weight-stream machinery shared by the 16-bit-operand MLP kernels: LDS-DMA from inline asm, the 3-slot
// ring loader and its publish protocol (see nsr_mlp_f16.hip for the measurements behind every choice).
#pragma once
#include "nsr_common.h"

namespace nsr {
namespace stream {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS byte address (32-bit) of a __shared__ object
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)((const __attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ const u32x4* lds_vec(unsigned byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const u32x4*)(const __attribute__((address_space(3))) u32x4*)(size_t)byte_addr;
#else
  (void)byte_addr;
  return nullptr;   // host pass of the single-source compile; never executed
#endif
}

// One wave-instruction: 64 lanes x 16 B, global -> LDS (dst = M0 + OFF + lane*16), wave-uniform 64-bit base in
// SGPRs + one 32-bit lane offset.  Inline asm on purpose: with the builtin in flight hipcc's waitcnt pass waits
// lgkmcnt(0) before every MFMA group; the asm form is invisible to it and is drained by hand (dma_drain).
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

// a chunk of the stream as this wave sees it: where it starts, how many 1 KiB pieces it has, and the
// contiguous 1/NW of it this wave moves (NW = waves per workgroup; all wave-uniform, SGPRs)
struct ChunkRef {
  int piece0, pieces;
  int first, count;
};
template <int NW>
__device__ __forceinline__ ChunkRef make_ref(int piece0, int pieces, int wave) {
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  constexpr int sh = NW == 4 ? 2 : 3;
  ChunkRef c;
  c.piece0 = piece0;
  c.pieces = pieces;
  c.first = (wave * pieces) >> sh;
  c.count = (((wave + 1) * pieces) >> sh) - c.first;
  return c;
}

// 3-slot LDS ring, chunk j in slot j % 3.  Protocol per wave while chunk j is consumed: at the publish point
// wait for the own DMA of chunk j+1, barrier (chunk j+1 published, chunk j-1's slot free), then fetch chunk
// j+2 into that slot; the last k-steps prefetch chunk j+1's first fragments, so nothing stalls at a boundary.
struct Loader {
  const float* stream;   // packed blob viewed as 32-bit words
  int wave;
  unsigned lane_off;     // lane * 16
  unsigned slot_cur, slot_next, slot_free;
  const char* dma_base;
  unsigned dma_lds;
  int dma_count;
};
__device__ __forceinline__ void loader_prepare_dma(Loader& ld, const ChunkRef& c, unsigned slot_lds) {
  ld.dma_count = c.count;
  ld.dma_base = reinterpret_cast<const char*>(ld.stream) + (size_t)(c.piece0 + c.first) * 1024;
  ld.dma_lds = slot_lds + (unsigned)c.first * 1024u;
}
// issue this wave's DMA piece number i of the chunk being fetched (no-op past its end); every wave owns at
// least MINP pieces of every chunk (32-piece minimum chunk / waves), so the first MINP issues are branch-free
template <int MINP>
__device__ __forceinline__ void loader_issue(const Loader& ld, int i) {
  if (i < MINP || i < ld.dma_count) {
    const char* base = ld.dma_base + (i >> 2) * 4096;
    const unsigned dst = ld.dma_lds + (unsigned)(i >> 2) * 4096u;
    switch (i & 3) {
      case 0: glds16_asm<0>(base, ld.lane_off, dst); break;
      case 1: glds16_asm<1024>(base, ld.lane_off, dst); break;
      case 2: glds16_asm<2048>(base, ld.lane_off, dst); break;
      default: glds16_asm<3072>(base, ld.lane_off, dst); break;
    }
  }
}
__device__ __forceinline__ void loader_advance(Loader& ld) {
  const unsigned t = ld.slot_cur;
  ld.slot_cur = ld.slot_next;
  ld.slot_next = ld.slot_free;
  ld.slot_free = t;
}
__device__ __forceinline__ void loader_publish(Loader& ld, const ChunkRef& c2) {
  dma_drain();
  __syncthreads();
  loader_prepare_dma(ld, c2, ld.slot_free);
}

}  // namespace stream
}  // namespace nsr
