// Weight gradient of a linear layer from two training panels on the fp16 matrix pipe (nsr_gemm.h: WgradArgs).
//
//   partial[z] (M x N) = sum over slice z of the points p of  dz[p][0..M) a[p][0..N)^T
//
// Round 5: both operands are the 2-byte panels the chain kernels write (nsr_f16x3_core.h): A = a gradient panel -- the
// backward chain's own fp16 `hi` operand, i.e. the input gradient rounded to 11 bits at the POINT's power-of-two scale
// (stored x pscale[p] = true) --, B = a forward panel -- the activation relu(z) (or g, for xyz_encoding_final) rounded to
// 11 bits.  ONE v_mfma_f32_32x32x16_f16 per product, fp32 accumulation: the contraction runs over hundreds of thousands of
// points whose rounding errors are independent, so a weight-gradient tensor moves by 1e-5 .. 8e-5 of its norm -- less than
// running the reference's own arithmetic in fp32 instead of fp64 does (profiles/r4_train_fp16_wgrad_study.txt).  Rounds
// 2-4 read fp32 panels (twice the bytes) and issued three MFMAs per product after splitting every value on the VALU.
//
// HBM-bound by design: every byte of both panels is read exactly once by exactly one workgroup (tile = all M x all N).
// Data path: a point group's rows of either panel are one contiguous run of 1 KiB units; LDS-DMA copies them, unit for
// unit, into a 4-stage LDS ring (three point groups in flight per CU), no register staging, no VALU on the way.  The units
// hold [point][4 features] (what a chain kernel's lane owns); the MFMA wants, per lane, one feature x 8 consecutive points:
// ds_read_b64_tr_b16 transposes on the way out of LDS -- a 16-lane group reads sixteen 8-byte pieces (4 points x 4 feature
// runs) and lane l receives feature l of 4 points.  The slot permutation the chain kernels store with, (2m + h) ^ 8u, makes
// the 32 pieces of a half-wave fall on 32 distinct 8-byte bank pairs.
// Range: the per-point factor pscale[p] * S (S = the power of two that puts the panel's largest true magnitude, kept by the
// backward chain, at 2^13 .. 2^14) is applied to the A fragments as two exact v_pk_mul_f16 (2^ceil(e/2) * 2^floor(e/2):
// neither the factors nor the intermediate can leave fp16's range while the result is inside it); elements far below the
// panel's maximum meet fp16's subnormal floor at 2^-38 of that maximum -- nothing a sum over the points can see.  All
// factors are powers of two: multiplying the loss by 2^k multiplies every gradient by exactly 2^k.
// The bias gradient (row sums of A at true scale) rides along: v_dot2_f32_f16 of the scaled fragments with ones.
#include "nsr_gemm.h"
#include "nsr_f16x3_core.h"

namespace nsr {
namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef NSR_SLICE_INLINE
#define NSR_SLICE_INLINE __forceinline__
#endif
constexpr int kNT = 512;                       // 8 waves, each a (32 BM) x (32 BN) tile of the TM x TN product
constexpr int kStages = 4;                     // LDS ring: the group being multiplied + three in flight
constexpr int kStageData = 32 * 1024;          // A rows then B rows of one point group (256 + 256 rows x 64 B at most)
constexpr int kStageBytes = kStageData + 256;  // + the group's pscale (one 256-byte DMA piece: 32 floats, twice)
constexpr int kFac0 = kStages * kStageBytes;   // per wave: F1[32], F2[32] fp16 factors of the current group
constexpr int kLdsBytes = kFac0 + 8 * 128;

// 64 lanes x 4 B, lane-linear, global -> LDS (the dword form of glds16_asm, nsr_f16x3_core.h)
__device__ __forceinline__ void glds4_asm(const char* base_uniform, unsigned lane_off, unsigned lds_dst_uniform) {
  unsigned long long tmp;
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_mov_b64 %0, %2\n\t"
      "global_load_lds_dword %1, %0"
      : "=&s"(tmp)
      : "v"(lane_off), "s"(base_uniform), "s"(lds_dst_uniform)
      : "memory");
}
// the transposing LDS read: lane l of a 16-lane group receives, of the sixteen 8-byte pieces the group's lanes address,
// half (l & 3) of the pieces of lanes (l >> 2), 4 + (l >> 2), 8 + (l >> 2), 12 + (l >> 2)
__device__ __forceinline__ h4 tr16(unsigned byte_addr) {
  return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4*)(size_t)byte_addr));
}
__device__ __forceinline__ u32x2 lds_u2(unsigned byte_addr) {
  return *(const __attribute__((address_space(3))) u32x2*)(size_t)byte_addr;
}

// TM x TN (BM, BN) = 256 x 256 (4, 2): trunk layers; 128 x 256 (2, 2): dir_encoding over g; 256 x 64 (1, 2): a trunk
// layer over the encoded position; 128 x 64 (1, 1): dir_encoding over the encoded direction
// One slice of one product: the point groups [g_begin, g_end) of w's panels -> partial slot `slot` (+ row sums).
// Ends drained and behind a workgroup barrier (the caller may re-use the LDS at once).
template <int TM, int kTN, int BM, int BN>
__device__ NSR_SLICE_INLINE void wgrad_slice(const WgradArgs& w, int64_t g_begin, int64_t g_end, int64_t slot, unsigned lds0) {
  constexpr int WM = TM / (32 * BM), WN = kTN / (32 * BN);   // wave grid
  static_assert(WM * WN == 8, "tile does not split over 8 waves");
  constexpr int kPiecesA = TM / 16, kPieces = (TM + kTN) / 16;   // 1 KiB units of a point group
  constexpr int PW = (kPieces + 7) / 8;                           // DMA pieces a wave issues per stage (the surplus re-fetches
                                                                  // the last piece: same bytes to the same place)
  static_assert((TM + kTN) * 64 <= kStageData, "stage too small");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM, li = lane & 31, h = lane >> 5;

  // power-of-two pre-scale of A from the panel's largest true magnitude
  const float amax = __uint_as_float(*w.a_max_bits);
  int eS = 13 - ((int)((__float_as_uint(amax) >> 23) & 255u) - 127);
  eS = eS < -100 ? -100 : (eS > 100 ? 100 : eS);
  if (!(amax > 0.0f)) eS = 0;
  const float inv_S = __uint_as_float((unsigned)(127 - eS) << 23);

  f32x16 acc[BM][BN];
#pragma unroll
  for (int bi = 0; bi < BM; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.0f;
  float rs[BM];
#pragma unroll
  for (int bi = 0; bi < BM; ++bi) rs[bi] = 0.0f;

  if (g_begin < g_end) {
    // ---- DMA of point group g into ring stage st (all operands wave-uniform but the lane offset); the panel descriptors
    // are copied out of the argument block once: the asm statements clobber memory, and the compiler would re-load them
    // from the kernel arguments (s_load + lgkmcnt(0), which also waits for the LDS reads) in every trip
    const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)(lane & 31) * 4u;
    const char* src0[PW];
    int64_t gstep[PW];
    unsigned dst0[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      int q = wave + 8 * j;
      if (q >= kPieces) q = kPieces - 1;
      const bool is_a = q < kPiecesA;
      src0[j] = is_a ? static_cast<const char*>(w.A) + q * 1024 : static_cast<const char*>(w.B) + (q - kPiecesA) * 1024;
      gstep[j] = is_a ? w.a_gbytes : w.b_gbytes;
      dst0[j] = (unsigned)q * 1024u;
    }
    const char* ps0 = reinterpret_cast<const char*>(w.a_pscale);
    auto issue = [&](int64_t g, int st) {
      if (g >= g_end) g = g_end - 1;     // tail: keep the per-stage operation count uniform (the vmcnt below relies on it)
      const unsigned dst = lds0 + (unsigned)st * kStageBytes;
#pragma unroll
      for (int j = 0; j < PW; ++j) glds16_asm<0>(src0[j] + g * gstep[j], lane16, dst + dst0[j]);
      glds4_asm(ps0 + g * 128, lane4, dst + kStageData);   // every wave: the same 128 B twice
    };
    // this lane's piece addresses inside a 2 KiB block for the transposing reads (see the header): 16-lane group g4 =
    // (unit u, k half hh); source lane i = lane & 15 addresses point 16 s + 8 hh + 4 t + (i >> 2), feature run i & 3
    const int g4 = lane >> 4, u = g4 & 1, hh = g4 >> 1, i16 = lane & 15;
    const unsigned c0 = (unsigned)(u * 1024 + 256 * hh + 32 * (i16 >> 2) + 16 * (i16 & 1) + 8 * ((i16 >> 1) & 1));
    const unsigned rd_t0 = c0 + 128u * (unsigned)u, rd_t1 = c0 + 128u * (unsigned)(1 - u);
    const unsigned fac = lds0 + kFac0 + (unsigned)wave * 128u;
    const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};

#pragma unroll
    for (int i = 0; i < kStages - 1; ++i) issue(g_begin + i, i);
    int st = 0;
    for (int64_t g = g_begin; g < g_end; ++g) {
      // this wave's pieces of group g have landed (two younger stages may stay in flight); then everybody's have, and
      // everybody has left the stage that is refilled next
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" : : "n"(2 * (PW + 1)) : "memory");
      issue(g + kStages - 1, (st + kStages - 1) & (kStages - 1));
      const unsigned sb = lds0 + (unsigned)st * kStageBytes;
      // ---- per-point factors of this group, made once per wave: e = log2(pscale * S) as 2^(e >> 1) * 2^(e - (e >> 1))
      if (lane < 32) {
        const unsigned pb = *(const __attribute__((address_space(3))) unsigned*)(size_t)(sb + kStageData + 4u * (unsigned)lane);
        int e = (int)((pb >> 23) & 255u) - 127 + eS;
        e = e < -28 ? -28 : (e > 28 ? 28 : e);
        const int e1 = e >> 1, e2 = e - e1;
        *(__attribute__((address_space(3))) unsigned short*)(size_t)(fac + 2u * (unsigned)lane) = (unsigned short)((15 + e1) << 10);
        *(__attribute__((address_space(3))) unsigned short*)(size_t)(fac + 64u + 2u * (unsigned)lane) = (unsigned short)((15 + e2) << 10);
      }
      asm volatile("" ::: "memory");   // the wave's own LDS writes are seen by its later reads (DS operations stay in order)
      h2 f1[2][2][2], f2[2][2][2];     // [k-step][t][pair]
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned o = 2u * (unsigned)(16 * s + 8 * hh + 4 * t);
          const u32x2 a = lds_u2(fac + o), b = lds_u2(fac + 64u + o);
          // through scalars: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index
          // (hipcc 7.2; found in the ISA: one ds_read_b32 per pair of points)
          const unsigned a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
          f1[s][t][0] = __builtin_bit_cast(h2, a0);
          f1[s][t][1] = __builtin_bit_cast(h2, a1);
          f2[s][t][0] = __builtin_bit_cast(h2, b0);
          f2[s][t][1] = __builtin_bit_cast(h2, b1);
        }
      const unsigned a_base = sb + (unsigned)(BM * wm) * 2048u, b_base = sb + (unsigned)(TM * 64) + (unsigned)(BN * wn) * 2048u;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        h8 bf[BN];
#pragma unroll
        for (int bj = 0; bj < BN; ++bj) {
          const h4 b0 = tr16(b_base + rd_t0 + (unsigned)(bj * 2048 + 512 * s));
          const h4 b1 = tr16(b_base + rd_t1 + (unsigned)(bj * 2048 + 512 * s));
          bf[bj] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int bi = 0; bi < BM; ++bi) {
          h4 a0 = tr16(a_base + rd_t0 + (unsigned)(bi * 2048 + 512 * s));
          h4 a1 = tr16(a_base + rd_t1 + (unsigned)(bi * 2048 + 512 * s));
          h2 p00 = {a0[0], a0[1]}, p01 = {a0[2], a0[3]}, p10 = {a1[0], a1[1]}, p11 = {a1[2], a1[3]};
          p00 = (p00 * f1[s][0][0]) * f2[s][0][0];
          p01 = (p01 * f1[s][0][1]) * f2[s][0][1];
          p10 = (p10 * f1[s][1][0]) * f2[s][1][0];
          p11 = (p11 * f1[s][1][1]) * f2[s][1][1];
          if (wn == 0) {   // wave-uniform: one wave column sums the rows (the bias gradient)
            rs[bi] = __builtin_amdgcn_fdot2(p00, ones, rs[bi], false);
            rs[bi] = __builtin_amdgcn_fdot2(p01, ones, rs[bi], false);
            rs[bi] = __builtin_amdgcn_fdot2(p10, ones, rs[bi], false);
            rs[bi] = __builtin_amdgcn_fdot2(p11, ones, rs[bi], false);
          }
          const h8 af = {p00[0], p00[1], p01[0], p01[1], p10[0], p10[1], p11[0], p11[1]};
#pragma unroll
          for (int bj = 0; bj < BN; ++bj) acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[bj], acc[bi][bj], 0, 0, 0);
        }
      }
      st = (st + 1) & (kStages - 1);
    }
    // the tail's surplus fetches must have landed before the ring is handed on (or the workgroup's LDS released)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // partial sums at true scale: D lane (column li, half h) of block (bi, bj) holds rows 8 (r >> 2) + 4 h + (r & 3)
  float* C = w.partial + slot * w.split_stride;
#pragma unroll
  for (int bi = 0; bi < BM; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj) {
      const int n = 32 * BN * wn + 32 * bj + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * BM * wm + 32 * bi + 8 * (r >> 2) + 4 * h + (r & 3);
        C[(int64_t)m * kTN + n] = acc[bi][bj][r] * inv_S;
      }
    }
  if (w.row_sums && wn == 0) {   // lane (li, h) summed row li of its blocks over the points 8 h .. 8 h + 7 (mod 16)
#pragma unroll
    for (int bi = 0; bi < BM; ++bi) {
      const float s2 = rs[bi] + __shfl_xor(rs[bi], 32, 64);
      if (h == 0) w.row_sums[slot * TM + 32 * BM * wm + 32 * bi + li] = s2 * inv_S;
    }
  }
}

// ALL weight-gradient products of one network pass in ONE launch (WgradJobs, nsr_gemm.h).  The products' point groups
// form one work list, product after product, a group of product p costing cost_p = M + N panel rows (the kernel is HBM
// bound: a group's time is the bytes it reads).  Workgroup w owns the cost range [w, w + 1) * per_wg of that list, i.e.
// for every product it overlaps the groups whose start cost falls inside -- so the ~256 resident workgroups (one per CU)
// carry the same number of bytes, each sweeps a few hundred point groups before it writes a partial tile, and a product
// owns about as many partial slots as its share of the bytes (21 for a 256 x 256 trunk product).  Deterministic: the
// mapping is static, no atomics.
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(2, 2))) wgrad_jobs_kernel(WgradJobs jobs) {
  __shared__ __attribute__((aligned(1024))) char lds[kLdsBytes];
  const unsigned lds0 = lds_addr(lds);
  const int64_t lo = (int64_t)blockIdx.x * jobs.per_wg;
  const int64_t hi = (lo + jobs.per_wg < jobs.total_cost) ? lo + jobs.per_wg : jobs.total_cost;
  for (int p = 0; p < jobs.n; ++p) {
    const WgradJob& q = jobs.j[p];
    const int64_t c0 = q.cost0, c1 = c0 + jobs.n_groups * q.cost;
    if (hi <= c0 || lo >= c1) continue;     // wave-uniform
    const int64_t a = (lo > c0 ? lo : c0) - c0, b = (hi < c1 ? hi : c1) - c0;
    const int64_t g_begin = (a + q.cost - 1) / q.cost;
    int64_t g_end = (b + q.cost - 1) / q.cost;
    if (g_end > jobs.n_groups) g_end = jobs.n_groups;
    const int64_t slot = (int64_t)blockIdx.x - q.w_first;
    const WgradArgs& w = q.w;
    if (w.M == 256 && w.N == 256) wgrad_slice<256, 256, 4, 2>(w, g_begin, g_end, slot, lds0);
    else if (w.M == 128 && w.N == 256) wgrad_slice<128, 256, 2, 2>(w, g_begin, g_end, slot, lds0);
    else if (w.M == 256 && w.N == 64) wgrad_slice<256, 64, 1, 2>(w, g_begin, g_end, slot, lds0);
    else wgrad_slice<128, 64, 1, 1>(w, g_begin, g_end, slot, lds0);
  }
}

}  // namespace

static bool wgrad_args_ok(const WgradArgs& w) {
  if (!w.A || !w.B || !w.partial || !w.a_max_bits || !w.a_pscale) return false;
  if (!((w.M == 256 || w.M == 128) && (w.N == 256 || w.N == 64))) return false;
  if (w.a_gbytes % 1024 || w.b_gbytes % 1024) return false;
  return !((reinterpret_cast<uintptr_t>(w.A) & 15) || (reinterpret_cast<uintptr_t>(w.B) & 15));
}

// fills cost0 / w_first / n_slots of every job and the work-list totals for `n_wg` workgroups; returns the number of
// workgroups actually needed (<= n_wg)
NSR_INTERNAL int wgrad_jobs_plan(WgradJobs& jobs, int64_t P, int n_wg) {
  jobs.n_groups = P / 32;
  int64_t c = 0;
  for (int p = 0; p < jobs.n; ++p) {
    jobs.j[p].cost = jobs.j[p].w.M + jobs.j[p].w.N;
    jobs.j[p].cost0 = c;
    c += jobs.n_groups * jobs.j[p].cost;
  }
  jobs.total_cost = c;
  if (n_wg < 1) n_wg = 1;
  jobs.per_wg = (c + n_wg - 1) / n_wg;
  if (jobs.per_wg < 1) jobs.per_wg = 1;
  for (int p = 0; p < jobs.n; ++p) {
    WgradJob& q = jobs.j[p];
    const int64_t c1 = q.cost0 + jobs.n_groups * q.cost;
    q.w_first = (int)(q.cost0 / jobs.per_wg);
    q.n_slots = jobs.n_groups > 0 ? (int)((c1 - 1) / jobs.per_wg) - q.w_first + 1 : 0;
  }
  return (int)((c + jobs.per_wg - 1) / jobs.per_wg);
}

NSR_INTERNAL int wgrad_jobs_f16(const WgradJobs& jobs, int n_wg, hipStream_t st) {
  if (jobs.n < 0 || jobs.n > kMaxWgradJobs || jobs.n_groups < 0) return NSR_ERR_INVALID_ARG;
  for (int p = 0; p < jobs.n; ++p)
    if (!wgrad_args_ok(jobs.j[p].w)) return NSR_ERR_INVALID_ARG;
  if (jobs.n == 0 || jobs.n_groups == 0 || n_wg < 1) return NSR_OK;
  hipLaunchKernelGGL(wgrad_jobs_kernel, dim3((unsigned)n_wg), dim3(kNT), 0, st, jobs);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

}  // namespace nsr
