// fp32-MFMA GEMM of the training step: C = epilogue(A · B^T), see nsr_gemm.h.
//
// v_mfma_f32_32x32x2_f32 (exact fp32 products, the arithmetic the reference trains in).  Workgroup = 4 waves,
// tile 128 x 128 x 32, each wave a 64 x 64 quadrant (2 x 2 MFMA blocks, 64 accumulator registers), two
// workgroups per CU.  Both operands are K-contiguous in memory, so the global -> LDS staging is a plain
// float4 copy (row stride 36 floats: the fragment reads below are bank-conflict free) and a lane's ds_read_b128
// feeds four consecutive MFMAs: lane (i, h) holds A[i][8q + 4h + t], t = 0..3, the B operand uses the same k
// mapping, so the K permutation cancels.  LDS is double buffered with register prefetch: one barrier per tile.
#include "nsr_gemm.h"

namespace nsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kTM = 128, kTN = 128, kTK = 32, kLd = kTK + 4;   // LDS row stride (floats), 16 B aligned

struct Stage {
  f32x4 a[4], b[4];
};

__device__ __forceinline__ void load_tile(Stage& s, const GemmArgs& g, int64_t m0, int n0, int64_t k0, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 3, c4 = idx & 7;
    int64_t m = m0 + row;
    m = m < g.M ? m : g.M - 1;
    int n = n0 + row;
    n = n < g.N ? n : g.N - 1;
    s.a[i] = *reinterpret_cast<const f32x4*>(g.A + m * g.lda + k0 + 4 * c4);
    s.b[i] = *reinterpret_cast<const f32x4*>(g.B + (int64_t)n * g.ldb + k0 + 4 * c4);
  }
}
__device__ __forceinline__ void store_tile(const Stage& s, float* As, float* Bs, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 3, c4 = idx & 7;
    *reinterpret_cast<f32x4*>(As + row * kLd + 4 * c4) = s.a[i];
    *reinterpret_cast<f32x4*>(Bs + row * kLd + 4 * c4) = s.b[i];
  }
}

__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmArgs g, int n_col_tiles, int64_t k_chunk) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (kTM + kTN) * kLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, li = lane & 31, h = lane >> 5;
  // column tile fastest: the workgroups that share an A row panel are launched back to back
  const int64_t bid = blockIdx.x;
  const int64_t m0 = (bid / n_col_tiles) * kTM;
  const int n0 = (int)(bid % n_col_tiles) * kTN;
  const int z = blockIdx.y;
  const int64_t k_begin = (int64_t)z * k_chunk;
  const int64_t k_end = (k_begin + k_chunk < g.K) ? k_begin + k_chunk : g.K;
  const int n_tiles = (int)((k_end - k_begin) / kTK);

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

  Stage st;
  if (n_tiles > 0) {
    load_tile(st, g, m0, n0, k_begin, tid);
    store_tile(st, lds, lds + kTM * kLd, tid);
  }
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    float* As = lds + (t & 1) * (kTM + kTN) * kLd;
    float* Bs = As + kTM * kLd;
    if (t + 1 < n_tiles) load_tile(st, g, m0, n0, k_begin + (int64_t)(t + 1) * kTK, tid);
    const float* ap = As + (64 * wm + li) * kLd + 4 * h;
    const float* bp = Bs + (64 * wn + li) * kLd + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) a[bi] = *reinterpret_cast<const f32x4*>(ap + 32 * bi * kLd + 8 * q);
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) b[bj] = *reinterpret_cast<const f32x4*>(bp + 32 * bj * kLd + 8 * q);
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
      float* An = lds + ((t + 1) & 1) * (kTM + kTN) * kLd;
      store_tile(st, An, An + kTM * kLd, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane (column li, half h) holds rows 8 (r >> 2) + 4 h + (r & 3) of each 32 x 32 block
  float* C = g.C ? g.C + (g.splits > 1 ? (int64_t)z * g.split_stride : 0) : nullptr;
#pragma unroll
  for (int bj = 0; bj < 2; ++bj) {
    const int n = n0 + 64 * wn + 32 * bj + li;
    if (!col_on[bj] || n >= g.n_valid) continue;
    const float bias = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int64_t mb = m0 + 64 * wm + 32 * bi + 4 * h;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t m = mb + 8 * rq + e;
          float x = acc[bi][bj][4 * rq + e] + bias;
          if (g.act == kActRelu) x = fmaxf(x, 0.0f);
          else if (g.act == kActSigmoid) x = 1.0f / (1.0f + expf(-x));
          if (g.mask && m < g.M) x = g.mask[m * g.ldm + n] > 0.0f ? x : 0.0f;
          v[e] = x;
          if (C && m < g.M) C[m * g.ldc + n] = x;
        }
        if (g.Ct) {
          const int64_t m = mb + 8 * rq;
          float* dst = g.Ct + (int64_t)n * g.ldct + m;
          if (m + 3 < g.M) *reinterpret_cast<f32x4*>(dst) = v;
          else
            for (int e = 0; e < 4; ++e)
              if (m + e < g.M) dst[e] = v[e];
        }
      }
    }
  }
}

}  // namespace

NSR_INTERNAL int gemm_nt(const GemmArgs& g, hipStream_t st) {
  if (g.M < 0 || g.N <= 0 || g.K < 0 || (g.K % kTK) != 0 || g.n_valid > g.N) return NSR_ERR_INVALID_ARG;
  if (!g.A || !g.B || (!g.C && !g.Ct)) return NSR_ERR_INVALID_ARG;
  if ((g.lda % 4) || (g.ldb % 4) || (g.Ct && (g.ldct % 4))) return NSR_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.B) & 15) ||
      (g.Ct && (reinterpret_cast<uintptr_t>(g.Ct) & 15)))
    return NSR_ERR_INVALID_ARG;
  const int splits = g.splits > 1 ? g.splits : 1;
  if (splits > 1 && (g.bias || g.mask || g.act != kActNone || g.Ct || !g.C)) return NSR_ERR_INVALID_ARG;
  if (g.M == 0) return NSR_OK;
  const int n_col_tiles = (g.N + kTN - 1) / kTN;
  const int64_t row_tiles = (g.M + kTM - 1) / kTM;
  const int64_t k_tiles = g.K / kTK;
  const int64_t k_chunk = ((k_tiles + splits - 1) / splits) * kTK;
  GemmArgs a = g;
  a.splits = splits;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)(row_tiles * n_col_tiles), (unsigned)splits), dim3(256), 0, st, a,
                     n_col_tiles, k_chunk);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

}  // namespace nsr
