// Weight gradient of a linear layer from two training panels on the split-fp16 matrix pipe (nsr_gemm.h: wgrad_f16x3).
//
//   partial[z] (M x N) = mult * sum over slice z of the points p of  A[p][0..M) B[p][0..N)^T
//
// A = a gradient panel (true-scale fp32, nsr_train_chain.hip), B = a forward panel (pre-activations x 2^6, read through
// max(., 0) when the consumer saw them behind a ReLU).  Both are "blocked transposed" ([group of 32 points][row][32],
// nsr_f16x3_core.h), so the K tile of one point group is ONE contiguous rows x 128 B run of each panel: the staging
// loads are perfectly coalesced float4 streams and every byte of both panels is read exactly once by exactly one
// workgroup (tile = all M x all N columns; PMC: 8.36 GB fetched per 2,048-ray step against 8.36 GB of panels,
// profiles/r2_train_traffic.json).  Each value is split into fp16 (hi, lo) on its way from registers to LDS
// and each product is a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3/16 of the
// fp32-MFMA cycles, which turns the product from matrix-pipe bound (68 % of the fp32 peak, 0.32 ms per fine-pass layer)
// into HBM bound (two panels of P KiB each).
// Range: gradients are tiny (1e-8 .. 1e-3), so A is multiplied by a power of two S that puts the panel's largest
// magnitude (kept by the backward chain: an LDS atomicMax per lane and layer, a global one per workgroup) at 2^13..2^14; elements far below the
// maximum lose relative precision against fp16's subnormal floor (2^-24 / S absolute), which is 2^-38 of the panel's
// maximum -- nothing a sum over the points can see.  S and B's 2^6 are removed from the accumulators (exactly) before
// the partial sums are written; the split-K reduction is the deterministic second pass of the fp32 path.
// While a thread stages its A values it also sums them per row in fp32: the bias gradient of the layer.
#include "nsr_gemm.h"

namespace nsr {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// the panels are read once: non-temporal staging loads (same box, 2,048-ray training step: 6.02-6.03 ms with plain
// loads, 5.94 ms with these)
#ifdef NSR_WGRAD_PLAIN_LOADS
#define NSR_WGRAD_LOAD(p) (*(p))
#else
#define NSR_WGRAD_LOAD(p) __builtin_nontemporal_load(p)
#endif

#ifndef NSR_SLICE_INLINE
#define NSR_SLICE_INLINE __forceinline__
#endif
constexpr int kTK = 32, kLd = kTK + 8;   // one point group per K tile; LDS row stride in halves (80 B: conflict-free b128 reads)
constexpr int kNT = 512;                 // 8 waves, each a (32 BM) x (32 BN) tile of the TM x TN product

// TM x TN (BM, BN) = 256 x 256 (4, 2): trunk layers; 128 x 256 (2, 2): dir_encoding over g; 256 x 64 (1, 2): a trunk
// layer over the encoded position; 128 x 64 (1, 1): dir_encoding over the encoded direction
// One slice of one product: the point groups [g_begin, g_end) of w's panels -> partial slot `slot` (+ row sums).  `lds`:
// 2 * TM * kLd + 2 * kTN * kLd halves.  Ends behind a workgroup barrier (the caller may re-use the LDS at once).
template <int TM, int kTN, int BM, int BN>
__device__ NSR_SLICE_INLINE void wgrad_slice(const WgradArgs& w, int64_t g_begin, int64_t g_end, int64_t slot, _Float16* lds) {
  constexpr int WM = TM / (32 * BM), WN = kTN / (32 * BN);   // wave grid
  static_assert(WM * WN == 8, "tile does not split over 8 waves");
  constexpr int kArrA = TM * kLd, kArrB = kTN * kLd;
  constexpr int NA = TM * kTK / 4 / kNT, NB = kTN * kTK / 4 / kNT;   // float4 per thread and K tile: 4 (2) and 4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM, li = lane & 31, h = lane >> 5;
  const int64_t z = slot;

  // power-of-two pre-scale of A from the panel's maximum magnitude
  const float amax = __uint_as_float(*w.a_max_bits);
  int e = 13 - ((int)((__float_as_uint(amax) >> 23) & 255u) - 127);
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  const float S = amax > 0.0f ? __uint_as_float((unsigned)(e + 127) << 23) : 1.0f;
  const float b_lower = w.b_relu ? 0.0f : -__builtin_inff();

  f32x16 acc[BM][BN];
#pragma unroll
  for (int bi = 0; bi < BM; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.0f;

  // staging: float4 number tid + 512 i of the group's contiguous (rows x 32) run = row (tid >> 3) + 64 i, k = 4 (tid & 7)
  f32x4 sa[NA], sb[NB];
  f32x4 ps = {1.0f, 1.0f, 1.0f, 1.0f};   // per-point scales of this thread's four points (the same for all its rows)
  float rs[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) rs[i] = 0.0f;
  auto load = [&](int64_t g) {
    const f32x4* ap = reinterpret_cast<const f32x4*>(w.A + g * w.a_gstride);
    const f32x4* bp = reinterpret_cast<const f32x4*>(w.B + g * w.b_gstride);
#pragma unroll
    for (int i = 0; i < NA; ++i) sa[i] = NSR_WGRAD_LOAD(ap + tid + kNT * i);
#pragma unroll
    for (int i = 0; i < NB; ++i) sb[i] = NSR_WGRAD_LOAD(bp + tid + kNT * i);
    if (w.a_pscale) ps = reinterpret_cast<const f32x4*>(w.a_pscale + g * 32)[tid & 7];
  };
  auto split_store = [&](const f32x4& v, _Float16* hi_arr, _Float16* lo_arr, int row) {
    h4 hi, lo;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const _Float16 x = (_Float16)v[e4];
      hi[e4] = x;
      lo[e4] = (_Float16)(v[e4] - (float)x);
    }
    const int off = row * kLd + 4 * (tid & 7);
    *reinterpret_cast<h4*>(hi_arr + off) = hi;
    *reinterpret_cast<h4*>(lo_arr + off) = lo;
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const f32x4 t = sa[i] * ps;           // true gradients
      rs[i] += (t[0] + t[1]) + (t[2] + t[3]);
      split_store(t * S, lds, lds + kArrA, (tid >> 3) + 64 * i);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      f32x4 v = sb[i];
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) v[e4] = fmaxf(v[e4], b_lower);
      split_store(v, lds + 2 * kArrA, lds + 2 * kArrA + kArrB, (tid >> 3) + 64 * i);
    }
  };

  if (g_begin < g_end) load(g_begin);
  for (int64_t g = g_begin; g < g_end; ++g) {
    store();
    __syncthreads();
    if (g + 1 < g_end) load(g + 1);
    const _Float16* ap = lds + (32 * BM * wm + li) * kLd + 8 * h;
    const _Float16* bp = lds + 2 * kArrA + (32 * BN * wn + li) * kLd + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h8 bh[BN], bl[BN];
#pragma unroll
      for (int bj = 0; bj < BN; ++bj) {
        bh[bj] = *reinterpret_cast<const h8*>(bp + 32 * bj * kLd + 16 * s);
        bl[bj] = *reinterpret_cast<const h8*>(bp + kArrB + 32 * bj * kLd + 16 * s);
      }
#pragma unroll
      for (int bi = 0; bi < BM; ++bi) {
        const h8 ah = *reinterpret_cast<const h8*>(ap + 32 * bi * kLd + 16 * s);
        const h8 al = *reinterpret_cast<const h8*>(ap + kArrA + 32 * bi * kLd + 16 * s);
#pragma unroll
        for (int bj = 0; bj < BN; ++bj) {
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[bj], acc[bi][bj], 0, 0, 0);   // small terms first
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[bj], acc[bi][bj], 0, 0, 0);
          acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[bj], acc[bi][bj], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // partial sums at true scale: D lane (column li, half h) of block (bi, bj) holds rows 8 (r >> 2) + 4 h + (r & 3)
  const float mult = w.out_scale / S;
  float* C = w.partial + (int64_t)z * w.split_stride;
#pragma unroll
  for (int bi = 0; bi < BM; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj) {
      const int n = 32 * BN * wn + 32 * bj + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * BM * wm + 32 * bi + 8 * (r >> 2) + 4 * h + (r & 3);
        C[(int64_t)m * kTN + n] = acc[bi][bj][r] * mult;
      }
    }
  if (w.row_sums) {   // the 8 threads that stage one row are neighbours
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float s = rs[i];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if ((tid & 7) == 0) w.row_sums[(int64_t)z * TM + (tid >> 3) + 64 * i] = s;
    }
  }
}

constexpr int kLdsHalves = 2 * 256 * kLd + 2 * 256 * kLd;   // the largest tile (256 x 256)

template <int TM, int kTN, int BM, int BN>
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_f16x3_kernel(WgradArgs w, int64_t groups_per_slice) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * TM * kLd + 2 * kTN * kLd];   // A hi | A lo | B hi | B lo
  const int64_t n_groups = w.P / 32;
  const int64_t g_begin = (int64_t)blockIdx.x * groups_per_slice;
  const int64_t g_end = (g_begin + groups_per_slice < n_groups) ? g_begin + groups_per_slice : n_groups;
  wgrad_slice<TM, kTN, BM, BN>(w, g_begin, g_end, blockIdx.x, lds);
}

// ALL weight-gradient products of one network pass in ONE launch (WgradJobs, nsr_gemm.h).  The products' point groups
// form one work list, product after product, a group of product p costing cost_p = M + N panel rows (the kernel is HBM
// bound: a group's time is the bytes it reads).  Workgroup w owns the cost range [w, w + 1) * per_wg of that list, i.e.
// for every product it overlaps the groups whose start cost falls inside -- so the ~256 resident workgroups (one per CU)
// carry the same number of bytes, each sweeps a few hundred point groups before it writes a partial tile instead of the
// 16-32 a per-product launch with 256 slices each allowed, and a product owns about as many partial slots as its share
// of the bytes (21 for a 256 x 256 trunk product instead of 256).  Deterministic: the mapping is static, no atomics.
__global__ void __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(2, 2))) wgrad_jobs_kernel(WgradJobs jobs) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[kLdsHalves];
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
    const WgradArgs& w = q.w;               // the slices never read w.P / w.splits
    if (w.M == 256 && w.N == 256) wgrad_slice<256, 256, 4, 2>(w, g_begin, g_end, slot, lds);
    else if (w.M == 128 && w.N == 256) wgrad_slice<128, 256, 2, 2>(w, g_begin, g_end, slot, lds);
    else if (w.M == 256 && w.N == 64) wgrad_slice<256, 64, 1, 2>(w, g_begin, g_end, slot, lds);
    else wgrad_slice<128, 64, 1, 1>(w, g_begin, g_end, slot, lds);
  }
}

}  // namespace

static int wgrad_shape(const WgradArgs& w) {
  return (w.N == 256 ? 0 : (w.N == 64 ? 2 : -8)) + (w.M == 256 ? 0 : (w.M == 128 ? 1 : -8));
}
static bool wgrad_args_ok(const WgradArgs& w) {
  if (!w.A || !w.B || !w.partial || !w.a_max_bits || wgrad_shape(w) < 0) return false;
  if (w.a_gstride % 4 || w.b_gstride % 4) return false;
  return !((reinterpret_cast<uintptr_t>(w.A) & 15) || (reinterpret_cast<uintptr_t>(w.B) & 15));
}

NSR_INTERNAL int wgrad_f16x3(const WgradArgs& w, hipStream_t st) {
  const int shape = wgrad_shape(w);
  if (!wgrad_args_ok(w) || w.splits < 1) return NSR_ERR_INVALID_ARG;
  if (w.P < 0 || w.P % 32 != 0) return NSR_ERR_INVALID_ARG;
  if (w.P == 0) return NSR_OK;
  const int64_t n_groups = w.P / 32;
  const int64_t per = (n_groups + w.splits - 1) / w.splits;
  const dim3 grid((unsigned)w.splits), block(kNT);
  if (shape == 0) hipLaunchKernelGGL((wgrad_f16x3_kernel<256, 256, 4, 2>), grid, block, 0, st, w, per);
  else if (shape == 1) hipLaunchKernelGGL((wgrad_f16x3_kernel<128, 256, 2, 2>), grid, block, 0, st, w, per);
  else if (shape == 2) hipLaunchKernelGGL((wgrad_f16x3_kernel<256, 64, 1, 2>), grid, block, 0, st, w, per);
  else hipLaunchKernelGGL((wgrad_f16x3_kernel<128, 64, 1, 1>), grid, block, 0, st, w, per);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
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

NSR_INTERNAL int wgrad_jobs_f16x3(const WgradJobs& jobs, int n_wg, hipStream_t st) {
  if (jobs.n < 0 || jobs.n > kMaxWgradJobs || jobs.n_groups < 0) return NSR_ERR_INVALID_ARG;
  for (int p = 0; p < jobs.n; ++p)
    if (!wgrad_args_ok(jobs.j[p].w)) return NSR_ERR_INVALID_ARG;
  if (jobs.n == 0 || jobs.n_groups == 0 || n_wg < 1) return NSR_OK;
  hipLaunchKernelGGL(wgrad_jobs_kernel, dim3((unsigned)n_wg), dim3(kNT), 0, st, jobs);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

}  // namespace nsr
