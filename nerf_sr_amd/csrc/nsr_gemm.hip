// fp32-MFMA GEMM of the training step, see nsr_gemm.h.
//
// v_mfma_f32_32x32x2_f32 (exact fp32 products, the arithmetic the reference trains in).  A wave owns a 64 x 64
// quadrant (2 x 2 MFMA blocks, 64 accumulator registers); a workgroup is 2 x WN waves: tile 128 x 128 with 4 waves
// and K tiles of 32, or 128 x 256 with 8 waves and K tiles of 16 when N >= 256 (an A row panel is read once; 61 KB
// of LDS and 128 VGPRs) -- in both cases TWO workgroups share a CU, so the prologue and the store-heavy epilogue of
// one hide behind the MFMAs of the other (with K = 256 a workgroup lives for only 16 K tiles).
// LDS is double buffered with register prefetch: one barrier per K tile.
// This MFMA takes ONE float per lane and operand, so either memory orientation of an operand is staged as it lies
// in memory (coalesced float4 loads along its contiguous axis) and only the LDS read differs:
//   K-contiguous operand: LDS [row][k], stride TK + 4 floats; lane (i, h) reads the float4 at k = 8q + 4h and feeds four
//                         consecutive MFMAs (k = 8q + 4h + t, t = 0..3); conflict free.
//   K-major operand:      LDS [k][row], stride rows + 4; lane (i, h) reads the scalar at (k = 8q + 4h + t, i): the 32
//                         lanes of a half read 32 consecutive floats, conflict free.
// Both use the same k <-> (q, h, t) mapping, so the K permutation cancels in the product.
#include "nsr_gemm.h"
#include "nsr_gemm_epilogue.h"

namespace nsr {

namespace {

constexpr int kTM = 128;
// K tile TK (16 or 32); K-contiguous LDS row stride TK + 4 floats (16 B aligned, conflict free for float4 reads)

template <int ROWS, int TK>   // floats of one staged operand tile (either orientation fits)
constexpr int tile_floats() { return (ROWS * (TK + 4) > TK * (ROWS + 4)) ? ROWS * (TK + 4) : TK * (ROWS + 4); }

// global -> registers: tile of ROWS x 32 of an operand, NT threads, VEC float4 per thread
template <int KMAJOR, int ROWS, int NT, int TK>
__device__ __forceinline__ void load_tile(f32x4 (&v)[ROWS * TK / 4 / NT], const float* __restrict__ p, int64_t ld,
                                          int64_t r0, int64_t extent, int64_t k0, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * TK / 4 / NT; ++i) {
    const int idx = tid + NT * i;
    if (KMAJOR) {
      const int k = idx / (ROWS / 4), r4 = idx % (ROWS / 4);
      int64_t r = r0 + 4 * r4;
      r = (r + 3 < extent) ? r : extent - 4;                 // past the edge: any valid data, dropped later
      v[i] = *reinterpret_cast<const f32x4*>(p + (k0 + k) * ld + r);
    } else {
      const int row = idx / (TK / 4), c4 = idx % (TK / 4);
      int64_t r = r0 + row;
      r = r < extent ? r : extent - 1;
      v[i] = *reinterpret_cast<const f32x4*>(p + r * ld + k0 + 4 * c4);
    }
  }
}
template <int KMAJOR, int ROWS, int NT, int TK>
__device__ __forceinline__ void store_tile(const f32x4 (&v)[ROWS * TK / 4 / NT], float* s, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * TK / 4 / NT; ++i) {
    const int idx = tid + NT * i;
    if (KMAJOR) {
      const int k = idx / (ROWS / 4), r4 = idx % (ROWS / 4);
      *reinterpret_cast<f32x4*>(s + k * (ROWS + 4) + 4 * r4) = v[i];
    } else {
      const int row = idx / (TK / 4), c4 = idx % (TK / 4);
      *reinterpret_cast<f32x4*>(s + row * (TK + 4) + 4 * c4) = v[i];
    }
  }
}
// the four operand values lane (i, h) feeds to the MFMAs t = 0..3 of group q, for the 32-row block at `row`
template <int KMAJOR, int ROWS, int TK>
__device__ __forceinline__ f32x4 frag(const float* s, int row, int q, int h) {
  if (KMAJOR) {
    const float* p = s + (8 * q + 4 * h) * (ROWS + 4) + row;
    return f32x4{p[0], p[ROWS + 4], p[2 * (ROWS + 4)], p[3 * (ROWS + 4)]};
  }
  return *reinterpret_cast<const f32x4*>(s + row * (TK + 4) + 8 * q + 4 * h);
}

template <int AK, int BK, int WN, int TK>
__global__ void __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(WN, WN)))   // two workgroups per CU
gemm_kernel(GemmArgs g, int n_col_tiles, int64_t k_chunk) {
  constexpr int NT = 128 * WN, TN = 64 * WN;
  constexpr int kAF = tile_floats<kTM, TK>(), kBF = tile_floats<TN, TK>();
  __shared__ __attribute__((aligned(16))) float lds[2 * (kAF + kBF)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, li = lane & 31, h = lane >> 5;
  // column tile fastest: the workgroups that share an A row panel are launched back to back
  const int64_t bid = blockIdx.x;
  const int64_t m0 = (bid / n_col_tiles) * kTM;
  const int n0 = (int)(bid % n_col_tiles) * TN;
  const int z = blockIdx.y;
  const int64_t k_begin = (int64_t)z * k_chunk;
  const int64_t k_end = (k_begin + k_chunk < g.K) ? k_begin + k_chunk : g.K;
  const int n_tiles = (int)((k_end - k_begin) / TK);

  f32x16 acc[2][2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.0f;
  // 32-column blocks of this wave that lie inside N (wave-uniform): the others are skipped on the matrix pipe
  bool col_on[2];
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) col_on[bj] = (n0 + 64 * wn + 32 * bj) < g.N;

  f32x4 sa[kTM * TK / 4 / NT], sb[TN * TK / 4 / NT];
  if (n_tiles > 0) {
    load_tile<AK, kTM, NT, TK>(sa, g.A, g.lda, m0, g.M, k_begin, tid);
    load_tile<BK, TN, NT, TK>(sb, g.B, g.ldb, n0, g.N, k_begin, tid);
    store_tile<AK, kTM, NT, TK>(sa, lds, tid);
    store_tile<BK, TN, NT, TK>(sb, lds + kAF, tid);
  }
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const float* As = lds + (t & 1) * (kAF + kBF);
    const float* Bs = As + kAF;
    if (t + 1 < n_tiles) {
      load_tile<AK, kTM, NT, TK>(sa, g.A, g.lda, m0, g.M, k_begin + (int64_t)(t + 1) * TK, tid);
      load_tile<BK, TN, NT, TK>(sb, g.B, g.ldb, n0, g.N, k_begin + (int64_t)(t + 1) * TK, tid);
    }
#pragma unroll
    for (int q = 0; q < TK / 8; ++q) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) a[bi] = frag<AK, kTM, TK>(As, 64 * wm + 32 * bi + li, q, h);
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) b[bj] = frag<BK, TN, TK>(Bs, 64 * wn + 32 * bj + li, q, h);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
          if (col_on[bj]) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
              acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bi][tt], b[bj][tt], acc[bi][bj], 0, 0, 0);
          }
    }
    if (t + 1 < n_tiles) {
      float* An = lds + ((t + 1) & 1) * (kAF + kBF);
      store_tile<AK, kTM, NT, TK>(sa, An, tid);
      store_tile<BK, TN, NT, TK>(sb, An + kAF, tid);
    }
    __syncthreads();
  }

  gemm_epilogue<TN>(g, acc, col_on, m0, n0, wm, wn, li, h, tid, z, bid / n_col_tiles, lds);
}

template <int AK, int BK>
int launch(const GemmArgs& a, int splits, hipStream_t st) {
  const bool wide = a.N >= 256;
  const int tn = wide ? 256 : 128;
  const int n_col_tiles = (a.N + tn - 1) / tn;
  const int64_t row_tiles = (a.M + kTM - 1) / kTM;
  const int64_t k_tiles = a.K / 32;
  const int64_t k_chunk = ((k_tiles + splits - 1) / splits) * 32;
  const dim3 grid((unsigned)(row_tiles * n_col_tiles), (unsigned)splits);
  // wide tile: K tiles of 16 keep the double-buffered LDS at 61 KB, so TWO 8-wave workgroups share a CU and one's
  // prologue / epilogue hides behind the other's MFMAs (with K = 256 a workgroup lives for only 8-16 K tiles)
  if (wide) hipLaunchKernelGGL((gemm_kernel<AK, BK, 4, 16>), grid, dim3(512), 0, st, a, n_col_tiles, k_chunk);
  else hipLaunchKernelGGL((gemm_kernel<AK, BK, 2, 32>), grid, dim3(256), 0, st, a, n_col_tiles, k_chunk);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

}  // namespace

NSR_INTERNAL int gemm(const GemmArgs& g, hipStream_t st) {
  if (g.M < 0 || g.N <= 0 || g.K < 0 || (g.K % 32) != 0 || g.n_valid > g.N) return NSR_ERR_INVALID_ARG;
  if (!g.A || !g.B || (!g.C && !g.Ct)) return NSR_ERR_INVALID_ARG;
  if ((g.lda % 4) || (g.ldb % 4) || (g.Ct && (g.ldct % 4))) return NSR_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.B) & 15) ||
      (g.Ct && (reinterpret_cast<uintptr_t>(g.Ct) & 15)))
    return NSR_ERR_INVALID_ARG;
  if (g.a_kmajor && (g.M % 4 != 0 || g.M < 4)) return NSR_ERR_INVALID_ARG;
  if (g.b_kmajor && (g.N % 4 != 0 || g.N < 4)) return NSR_ERR_INVALID_ARG;
  const int splits = g.splits > 1 ? g.splits : 1;
  if (splits > 1 && (g.bias || g.mask || g.act != kActNone || g.Ct || !g.C || g.col_sums)) return NSR_ERR_INVALID_ARG;
  if (g.M == 0) return NSR_OK;
  GemmArgs a = g;
  a.splits = splits;
  if (!g.a_kmajor && !g.b_kmajor) return launch<0, 0>(a, splits, st);
  if (!g.a_kmajor && g.b_kmajor) return launch<0, 1>(a, splits, st);
  if (g.a_kmajor && g.b_kmajor) return launch<1, 1>(a, splits, st);
  return NSR_ERR_UNSUPPORTED;
}

}  // namespace nsr
