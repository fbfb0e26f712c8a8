// Split-fp16 GEMM of the forward products (nsr_gemm.h: gemm_f16x3), v_mfma_f32_32x32x16_f16 x 3 per product.
//
// Workgroup = 2 x WN waves, each wave a 64 x 64 quadrant (2 x 2 accumulator blocks): 4 waves on a 128 x 128 tile, or
// 8 waves on 128 x 256 when N >= 256 (the A panel -- activations, HBM traffic -- is then read once); K tiles of 32 =
// two MFMA k-steps.  LDS holds the tile as four fp16 arrays (A hi, A lo, B hi, B lo; row stride 40 halves = 80 B,
// which makes the 16-byte fragment reads of eight consecutive rows hit eight different 16-byte bank groups; the staging
// stores need their own lane -> row assignment to stay conflict-free at that pitch, see stage_row);
// single buffered with register prefetch (40 / 60 KB per workgroup; two waves per SIMD: two 4-wave workgroups or
// one 8-wave workgroup per CU, so the MFMAs of one wave hide the staging and barriers of the other).  A is fp32 in memory and is split into (hi, lo) on its way from
// registers to LDS; B arrives pre-split.  Optional implicit im2col: the A rows are gathered from an NHWC activation
// (a K tile of 32 channels never straddles a tap because cin % 32 == 0), so a 3 x 3 convolution needs no col matrix.
#include <stdlib.h>
#include "nsr_gemm.h"
#include "nsr_gemm_epilogue.h"

namespace nsr {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// development switch: -DNSR_GEMM_K32_ONLY=1 keeps every launch on K tiles of 32 (A/B in profiles/)
#ifndef NSR_GEMM_K32_ONLY
#define NSR_GEMM_K32_ONLY 0
#endif
constexpr int kTM = 128, kTK = 32;   // row tile; default K tile (the kernels pad LDS rows by 8 halves)

// Staging row of a thread (16-byte chunks, kCh per row and plane).  K tiles of 32: a row is 64 B of data at an 80 B pitch
// and a ds_write_b128 is served eight consecutive lanes per LDS cycle out of 32 banks (128 B) -- lanes 0-3 on row r and
// lanes 4-7 on row r + 1 overlap by four banks, a 2-way conflict on EVERY staging store (PMC, profiles/r3_refine_pmc.json:
// 7.7 conflict cycles per store, 16 array cycles against the instruction's 13).  Rows r and r + 4 are 320 B = 16 banks
// apart and do not collide, so the eight lanes take rows {r, r + 4}.  K tiles of 64: eight lanes = one 128 B row, no conflict.
// (-DNSR_GEMM_NO_WROWS: the consecutive-row assignment, for A/B runs.)
template <int kCh>
__device__ __forceinline__ int stage_row(int tid) {
#ifndef NSR_GEMM_NO_WROWS
  if constexpr (kCh == 4) {
    const int g8 = tid >> 3;
    return (g8 >> 2) * 8 + (g8 & 3) + 4 * ((tid >> 2) & 1);
  }
#endif
  return tid / kCh;
}

__global__ void split_f16_kernel(const float* __restrict__ w, int64_t n, unsigned short* __restrict__ hi,
                                 unsigned short* __restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = kSplitScale * w[i];
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  hi[i] = __builtin_bit_cast(unsigned short, h);
  lo[i] = __builtin_bit_cast(unsigned short, l);
}

struct RowSrc {   // where the four A rows this thread stages come from (implicit im2col)
  int img, oy, ox;
  bool ok;
};

// (hi, lo) fp16 planes out: bias -> activation -> split -> two 4-byte stores per lane and row pair.  Lane (column li)
// of an accumulator block holds one column of 16 rows; neighbouring lanes swap one value per row pair so that every
// lane owns TWO adjacent columns of ONE row and the 2-byte elements leave as packed 4-byte words.
template <int BN, int BM, bool GROUPED>
__device__ __forceinline__ void epilogue_planes_t(const GemmF16Args& a, f32x16 (&acc)[BM][BN], const bool (&col_on)[BN], int64_t m0,
                                                  int n0, int wm, int wn, int li, int h) {
  const GemmArgs& g = a.g;
  const bool scaled = g.acc_scale != 0.0f && g.acc_scale != 1.0f;
  const bool odd = li & 1;
  // GROUPED (nsr_gemm.h): rows 8 q .. 8 q + 7 = one pixel of 8 images; output row of member r: row0 + r * px
  const unsigned per = GROUPED ? (unsigned)(a.conv.Ho * a.conv.Wo) : 1u;
  const unsigned n_q = (unsigned)(g.M / 8);
  int64_t row0[BM][4];
  if (GROUPED) {
#pragma unroll
    for (int bi = 0; bi < BM; ++bi)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const unsigned q = (unsigned)((m0 + 32 * BM * wm + 32 * bi) / 8) + rq;      // < 2^31: M / 8 pixels
        const unsigned b = q / per;
        row0[bi][rq] = (int64_t)b * 8 * per + (q - b * per);
      }
  }
#pragma unroll
  for (int bj = 0; bj < BN; ++bj) {
    if (!col_on[bj]) continue;                       // wave-uniform
    const int n = n0 + 32 * BN * wn + 32 * bj + li;
    const float bias = (g.bias && n < g.N) ? g.bias[n] : 0.0f;
    const int col = n & ~1;
#pragma unroll
    for (int bi = 0; bi < BM; ++bi) {
      const int64_t mb = m0 + 32 * BM * wm + 32 * bi + 4 * h;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = scaled ? fmaf(acc[bi][bj][4 * rq + e], g.acc_scale, bias) : acc[bi][bj][4 * rq + e] + bias;
          if (g.act == kActRelu) v = fmaxf(v, 0.0f);
          else if (g.act == kActSigmoid) v = 1.0f / (1.0f + expf(-v));
          else if (g.act == kActTanh) v = tanhf(v);
          x[e] = v;
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const float xa = x[2 * pr], xb = x[2 * pr + 1];
          const float got = __shfl_xor(odd ? xa : xb, 1, 64);     // even lanes receive the partner's xa, odd lanes its xb
          const float c0 = odd ? got : xa, c1 = odd ? xb : got;   // columns col, col + 1 of row (odd ? b : a)
          const int64_t m = mb + 8 * rq + 2 * pr + (odd ? 1 : 0);
          const _Float16 h0 = (_Float16)c0, h1 = (_Float16)c1;
          const _Float16 l0 = (_Float16)(c0 - (float)h0), l1 = (_Float16)(c1 - (float)h1);
          if (m < g.M && col < g.n_valid) {
            const int64_t out_row = GROUPED ? row0[bi][rq] + (int64_t)(4 * h + 2 * pr + (odd ? 1 : 0)) * per : m;
            const int64_t off = out_row * g.ldc + col;
            *reinterpret_cast<unsigned*>(a.Ch + off) =
                (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            *reinterpret_cast<unsigned*>(a.Ch + a.c_plane + off) =
                (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
          }
        }
        if (GROUPED) {   // max over the 8 members: four in this lane, four in the lane of the other half; then pair the columns
          float mx = nsr_max_nan(nsr_max_nan(x[0], x[1]), nsr_max_nan(x[2], x[3]));      // NaN-propagating, like torch.max
          mx = nsr_max_nan(mx, __shfl_xor(mx, 32, 64));
          const float nb = __shfl_xor(mx, 1, 64);
          const unsigned q = (unsigned)((m0 + 32 * BM * wm + 32 * bi) / 8) + rq;
          if (h == 0 && !odd && q < n_q && col < g.n_valid) {
            const _Float16 h0 = (_Float16)mx, h1 = (_Float16)nb;      // split(max) == lexicographic max of the splits (RNE is monotone)
            const _Float16 l0 = (_Float16)(mx - (float)h0), l1 = (_Float16)(nb - (float)h1);
            const int64_t off = (int64_t)q * a.ldm + col;
            *reinterpret_cast<unsigned*>(a.Mh + off) =
                (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            *reinterpret_cast<unsigned*>(a.Mh + a.m_plane + off) =
                (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
          }
        }
      }
    }
  }
}
// the plain instantiation carries none of the grouped-row arithmetic (the K = 32 first layers are epilogue-bound)
template <int BN, int BM = 2>
__device__ __forceinline__ void epilogue_planes(const GemmF16Args& a, f32x16 (&acc)[BM][BN], const bool (&col_on)[BN], int64_t m0,
                                                int n0, int wm, int wn, int li, int h) {
  if (a.group == 8) epilogue_planes_t<BN, BM, true>(a, acc, col_on, m0, n0, wm, wn, li, h);
  else epilogue_planes_t<BN, BM, false>(a, acc, col_on, m0, n0, wm, wn, li, h);
}

// launch index -> tile index, see the kernel
__device__ __forceinline__ int64_t xcd_tile(int64_t launch_idx, int64_t n_blocks) {
#ifdef NSR_GEMM_NO_XCD
  return launch_idx;
#else
  const int64_t per_xcd = (n_blocks + 7) / 8;
  return (launch_idx % 8) * per_xcd + launch_idx / 8;
#endif
}

// APL: A comes as (hi, lo) fp16 planes.  TK: K tile, 32 or (plane A only) 64 halves -- twice the MFMAs between two
// barriers for the same staging overhead, 110 KB of LDS on the 8-wave tile
// BN: accumulator blocks per wave along N.  2 (default): a wave owns 64 x 64.  4: a wave owns 64 x 128 -- the 128 x 256 tile
// is then ONE workgroup of four waves (one per SIMD), two such workgroups share a CU and run out of phase, and a k-step
// reads 12 fragments for 24 MFMAs instead of 8 for 12 (the 8-wave tile keeps the LDS port ~90 % as busy as the matrix pipe)
// FULLN: every column block of the tile is inside N (N a multiple of the tile width) -- the k-steps are then straight-line
// code.  With the test `col_on[bj]` in the loop the compiler cannot prove it wave-uniform, wraps every group of six MFMAs
// in an EXEC-mask branch, and each k-step becomes read 8 fragments -> wait -> 12 MFMAs with nothing in flight across the
// block boundaries.
template <int WN, bool APL, int TK = 32, int BN = 2, bool FULLN = false>
__global__ void __launch_bounds__(128 * WN) __attribute__((amdgpu_waves_per_eu(2, 2)))   // <= 256 registers
gemm_f16x3_kernel(GemmF16Args a, int n_col_tiles, int64_t n_blocks) {
  static_assert(TK == 32 || (APL && TK == 64), "K tiles of 64 are built for pre-split A only");
  static_assert(BN == 2 || (APL && TK == 32), "64 x 128 wave tiles: pre-split A, K tiles of 32 (register budget)");
  constexpr int kLd = TK + 8, kArrA = kTM * kLd;            // shadow the 32-wide constants of the file
  constexpr int kCh = TK / 8;                               // 16-byte chunks (8 halves) per row and plane
  constexpr int NT = 128 * WN, kTN = 32 * BN * WN, kArrB = kTN * kLd;
  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * kArrA + 2 * kArrB];   // A hi | A lo | B hi | B lo
  const GemmArgs& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1, li = lane & 31, h = lane >> 5;
  // XCD-aware tile order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (private L2s), so the
  // launch index is remapped such that each XCD works through ONE contiguous run of tiles: the column tiles that share an
  // A row panel, and the neighbouring row tiles whose 3 x 3 gathers overlap, then hit the same L2.  (Measured neutral
  // on the refinement pass, 44.4-45.3 ms either way: the 256 MB Infinity Cache already serves those re-reads.)
  const int64_t bid = xcd_tile(blockIdx.x, n_blocks);
  if (bid >= n_blocks) return;                     // padding of the launch to a multiple of 8 (whole workgroup)
  const int64_t m0 = (bid / n_col_tiles) * kTM;
  const int n0 = (int)(bid % n_col_tiles) * kTN;
  const int n_tiles = (int)(g.K / TK);
  const bool conv = a.conv.cin > 0;

  f32x16 acc[2][BN];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.0f;
  bool col_on[BN];
#pragma unroll
  for (int bj = 0; bj < BN; ++bj) col_on[bj] = (n0 + 32 * BN * wn + 32 * bj) < g.N;

  // ---- staging assignment.  fp32 A: NA x (row = (tid >> 3) + (NT / 8) i, float4 column c4 = tid & 7), split on the way
  // to LDS; plane A: NP x (row = (tid >> 2) + (NT / 4) i, 8-half chunk c8 = tid & 3) of each plane, copied as it is.
  // B: 2 x 2 x (row = (tid >> 2) + (NT / 4) i, 8-half chunk c8 = tid & 3)
  // (with K tiles of 64: eight chunks per row, rows (tid >> 3) + (NT / 8) i)
  constexpr int NA = APL ? kTM * kCh / NT : 1024 / NT, kARows = APL ? NT / kCh : NT / 8, kBRows = NT / kCh;
  constexpr int NBP = kTN * kCh / NT;                       // B chunks per thread and plane
  const int c4 = tid & 7, c8 = tid % kCh, br0 = stage_row<kCh>(tid);
  const int ar0 = APL ? stage_row<kCh>(tid) : (tid >> 3);
  const int a_col = APL ? 8 * c8 : 4 * c4;                  // column offset inside the K tile, in elements
  // row base of the A operand in ELEMENTS from its base pointer (fp32: g.A floats; planes: a.Ah halves); -1 = zero row
  int64_t arow[NA];
  RowSrc rs[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int64_t m = m0 + ar0 + kARows * i;
    m = m < g.M ? m : g.M - 1;
    if (conv) {
      const int64_t per = (int64_t)a.conv.Ho * a.conv.Wo;
      int64_t img = m / per, pix = m % per;
      if (a.group == 8) {                      // grouped rows: m = (b * px + pixel) * 8 + r (nsr_gemm.h)
        const int64_t q = m >> 3;
        img = (q / per) * 8 + (m & 7);
        pix = q % per;
      }
      rs[i].img = (int)img;
      rs[i].oy = (int)(pix / a.conv.Wo);
      rs[i].ox = (int)(pix % a.conv.Wo);
      arow[i] = 0;
    } else {
      arow[i] = m * g.lda;
    }
  }
  // BN == 2: one pointer pair per staged B row (rows past N clamp to the last one).  BN == 4 (full column tiles only,
  // N % 256 == 0): ONE base pointer, the rows kBRows apart and the lo plane at a wave-uniform distance -- 14 registers less
  constexpr int NPTR = BN == 2 ? NBP : 1;
  const unsigned short* bh_row[NPTR];
  const unsigned short* bl_row[NPTR];
#pragma unroll
  for (int i = 0; i < NPTR; ++i) {
    int n = n0 + br0 + kBRows * i;
    n = n < g.N ? n : g.N - 1;
    bh_row[i] = a.Bh + (int64_t)n * a.ldbh;
    bl_row[i] = a.Bl + (int64_t)n * a.ldbh;
  }
  const int64_t b_step = (int64_t)kBRows * a.ldbh, b_lo = a.Bl - a.Bh;
  f32x4 sa[APL ? 1 : NA];
  u32x4 sah[APL ? NA : 1], sal[APL ? NA : 1];
  u32x4 sbh[NBP], sbl[NBP];
  auto load = [&](int t) {
    const int64_t k0 = (int64_t)t * TK;
    int64_t koff = k0;
    if (conv) {
      // the gather address of a row changes only when the K tile enters the next tap (every cin / 32 tiles):
      // arow[i] then points at channel 0 of the source pixel under that tap, or is -1 inside the zero padding
      const int cbase = (int)(k0 % a.conv.cin);
      koff = cbase;
      if (cbase == 0) {
        const int tap = (int)(k0 / a.conv.cin), ky = tap / 3, kx = tap % 3;
        const int Hin = a.conv.up ? 2 * a.conv.Hs : a.conv.Hs, Win = a.conv.up ? 2 * a.conv.Ws : a.conv.Ws;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int iy = rs[i].oy * a.conv.stride + ky - 1, ix = rs[i].ox * a.conv.stride + kx - 1;
          arow[i] = -1;
          if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) {
            const int sy = a.conv.up ? iy >> 1 : iy, sx = a.conv.up ? ix >> 1 : ix;
            arow[i] = (((int64_t)rs[i].img * a.conv.Hs + sy) * a.conv.Ws + sx) * g.lda;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int64_t off = arow[i] + koff + a_col;
      if (APL) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        sah[i] = arow[i] >= 0 ? *reinterpret_cast<const u32x4*>(a.Ah + off) : zero;
        sal[i] = arow[i] >= 0 ? *reinterpret_cast<const u32x4*>(a.Ah + a.a_plane + off) : zero;
      } else {
        sa[i] = arow[i] >= 0 ? *reinterpret_cast<const f32x4*>(g.A + off) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
      if constexpr (BN == 2) {
        sbh[i] = *reinterpret_cast<const u32x4*>(bh_row[i] + k0 + 8 * c8);
        sbl[i] = *reinterpret_cast<const u32x4*>(bl_row[i] + k0 + 8 * c8);
      } else {
        const unsigned short* q = bh_row[0] + i * b_step + k0 + 8 * c8;
        sbh[i] = *reinterpret_cast<const u32x4*>(q);
        sbl[i] = *reinterpret_cast<const u32x4*>(q + b_lo);
      }
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int off = (ar0 + kARows * i) * kLd + a_col;
      if (APL) {
        *reinterpret_cast<u32x4*>(lds + off) = sah[i];
        *reinterpret_cast<u32x4*>(lds + kArrA + off) = sal[i];
      } else {
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const _Float16 x = (_Float16)sa[i][e];
          hi[e] = x;
          lo[e] = (_Float16)(sa[i][e] - (float)x);
        }
        *reinterpret_cast<h4*>(lds + off) = hi;
        *reinterpret_cast<h4*>(lds + kArrA + off) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
      const int off = (br0 + kBRows * i) * kLd + 8 * c8;
      *reinterpret_cast<u32x4*>(lds + 2 * kArrA + off) = sbh[i];
      *reinterpret_cast<u32x4*>(lds + 2 * kArrA + kArrB + off) = sbl[i];
    }
  };

  if (n_tiles > 0) load(0);
  for (int t = 0; t < n_tiles; ++t) {
    store();
    __syncthreads();
    if (t + 1 < n_tiles) load(t + 1);
    const _Float16* ap = lds + (64 * wm + li) * kLd + 8 * h;
    const _Float16* bp = lds + 2 * kArrA + (32 * BN * wn + li) * kLd + 8 * h;
    // the k-steps run at raised priority: the wave sharing this SIMD is, by then, usually staging its next tile (address
    // arithmetic, loads, LDS stores), and its VALU / SALU stream otherwise takes issue slots between this wave's MFMAs
    // (-1 % on the refinement pass, interleaved A/B, profiles/r3_refine_tiles.txt)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < TK / 16; ++s) {
      h8 ah[2], al[2], bh[BN], bl[BN];
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) {
        ah[bi] = *reinterpret_cast<const h8*>(ap + 32 * bi * kLd + 16 * s);
        al[bi] = *reinterpret_cast<const h8*>(ap + kArrA + 32 * bi * kLd + 16 * s);
      }
#pragma unroll
      for (int bj = 0; bj < BN; ++bj) {
        bh[bj] = *reinterpret_cast<const h8*>(bp + 32 * bj * kLd + 16 * s);
        bl[bj] = *reinterpret_cast<const h8*>(bp + kArrB + 32 * bj * kLd + 16 * s);
      }
#pragma unroll
      for (int bj = 0; bj < BN; ++bj)
        if (BN == 4 || FULLN || col_on[bj]) {     // BN == 4: full column tiles only
#pragma unroll
          for (int bi = 0; bi < 2; ++bi) {
            acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[bi], bh[bj], acc[bi][bj], 0, 0, 0);   // small terms first
            acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[bi], bl[bj], acc[bi][bj], 0, 0, 0);
            acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[bi], bh[bj], acc[bi][bj], 0, 0, 0);
          }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  if constexpr (BN == 2) {
    if (a.Ch) epilogue_planes<2>(a, acc, col_on, m0, n0, wm, wn, li, h);
    else gemm_epilogue<kTN>(g, acc, col_on, m0, n0, wm, wn, li, h, tid, 0, bid / n_col_tiles, reinterpret_cast<float*>(lds));
  } else {
    epilogue_planes<BN>(a, acc, col_on, m0, n0, wm, wn, li, h);       // launched for plane output only (gemm_f16x3)
  }
}


// (Round 2's LDS-DMA fed variant of this kernel -- pre-split operands streamed global -> LDS into a double-buffered,
// fragment-ordered image -- was correct and 9 % slower on the refinement pass (48.5 vs 44.5 ms: an LDS-DMA costs the
// issuing wave 60-180 cycles against a handful for a plain global_load_dwordx4, and this kernel has the registers to
// stage through); it was removed in round 3, the measurement is in DESIGN section 9.)

}  // namespace

NSR_INTERNAL int split_f16(const float* w, int64_t n, unsigned short* hi, unsigned short* lo, hipStream_t st) {
  if (!w || !hi || !lo || n < 0) return NSR_ERR_INVALID_ARG;
  if (n == 0) return NSR_OK;
  hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, n, hi, lo);
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

NSR_INTERNAL int gemm_f16x3(const GemmF16Args& a, hipStream_t st) {
  const GemmArgs& g = a.g;
  if (g.M < 0 || g.N <= 0 || g.K <= 0 || (g.K % kTK) != 0 || g.n_valid > g.N) return NSR_ERR_INVALID_ARG;
  if (!a.Bh || !a.Bl || g.Ct || g.splits > 1 || g.a_kmajor || g.b_kmajor) return NSR_ERR_INVALID_ARG;
  if ((a.ldbh % 8) || (reinterpret_cast<uintptr_t>(a.Bh) & 15) || (reinterpret_cast<uintptr_t>(a.Bl) & 15)) return NSR_ERR_INVALID_ARG;
  if (a.Ah) {   // pre-split A: 16-byte chunks of 8 halves
    if ((g.lda % 8) || (a.a_plane % 8) || (reinterpret_cast<uintptr_t>(a.Ah) & 15)) return NSR_ERR_INVALID_ARG;
  } else if (!g.A || (g.lda % 4) || (reinterpret_cast<uintptr_t>(g.A) & 15)) {
    return NSR_ERR_INVALID_ARG;
  }
  if (a.Ch) {   // plane output: packed pairs of columns
    if ((g.ldc % 2) || (a.c_plane % 2) || (g.n_valid % 2) || (reinterpret_cast<uintptr_t>(a.Ch) & 3) || g.col_sums || g.mask)
      return NSR_ERR_INVALID_ARG;
  } else if (!g.C) {
    return NSR_ERR_INVALID_ARG;
  }
  if (a.conv.cin > 0 && ((a.conv.cin % kTK) != 0 || g.K != 9 * a.conv.cin)) return NSR_ERR_INVALID_ARG;
  if (a.group != 0) {   // grouped rows: conv gather, plane output, whole groups, packed column pairs of the max planes
    if (a.group != 8 || a.conv.cin <= 0 || !a.Ch || !a.Mh || (g.M % 8) || (a.ldm % 2) || (a.m_plane % 2) ||
        (reinterpret_cast<uintptr_t>(a.Mh) & 3))
      return NSR_ERR_INVALID_ARG;
  }
  if (g.M == 0) return NSR_OK;
  // Tile choice (one MI355X box, 800 x 800 refinement pass, profiles/r3_refine_tiles.txt): two 4-wave workgroups per CU
  // beat one 8-wave workgroup whatever they compute -- they run out of phase, so one stages and waits at its barriers
  // while the other feeds the matrix pipe (8-wave 128 x 256: 41.4 ms; 4-wave 128 x 128 everywhere: 40.0).  Of the 4-wave
  // shapes, the 128 x 256 tile with 64 x 128 per wave ("quad", BN = 4: the A panel is read once and a k-step costs 12
  // fragment reads per 24 MFMAs) wins where there are enough row tiles to fill the chip with it; 128 x 128 otherwise.
  // Going further the same way -- ONE wave per SIMD with 128 x 128 (or 128 x 64) per wave, 16 fragment reads per 48 MFMAs --
  // was built and measured SLOWER (48.0 ms; every layer, e.g. 5.54 vs 5.08 ms): with a single wave nothing hides the
  // staging loads, the ds_writes and the two barriers of a K tile but the wave's own instruction order, and the
  // compiler's order does not (the inference kernel gets there with a hand-pinned schedule and an LDS-DMA ring).
  // Numbers and the kernel's description: profiles/r3_refine_tiles.txt.
  const int64_t row_tiles = (g.M + kTM - 1) / kTM;
  const bool quad_ok = g.N >= 256 && (g.N % 256) == 0 && a.Ah && a.Ch;
  bool wide = false;                                   // the 8-wave tile: NSR_GEMM_TILE=wide only
  bool four_wave_wide = quad_ok && row_tiles * (g.N / 256) >= 1024;
  // development switches (A/B runs on one box): NSR_GEMM_TILE=narrow|wide|quad overrides the choice, NSR_GEMM_TK=32 keeps K tiles of 32
  const char* e_tile = nsr_dev_env("NSR_GEMM_TILE");
  const char* e_tk = nsr_dev_env("NSR_GEMM_TK");
  const char* e_fulln = nsr_dev_env("NSR_GEMM_FULLN");      // =0: keep the column test in the k-steps (A/B)
  if (e_tile && e_tile[0] == 'n') four_wave_wide = false;
  if (e_tile && e_tile[0] == 'w') { wide = g.N >= 256; four_wave_wide = false; }
  if (e_tile && e_tile[0] == 'q') four_wave_wide = quad_ok;
  if (four_wave_wide) wide = true;   // 128 x 256 tile on four waves (64 x 128 each)
  const int tn = wide ? 256 : 128;
  const int n_col_tiles = (g.N + tn - 1) / tn;
  const int64_t n_blocks = row_tiles * n_col_tiles;
  const dim3 grid((unsigned)(((n_blocks + 7) / 8) * 8));
  if (a.Ah) {
    const bool k64 = (g.K % 64) == 0 && (a.conv.cin <= 0 || (a.conv.cin % 64) == 0) && !NSR_GEMM_K32_ONLY &&
                     !(e_tk && e_tk[0] == '3');
    const bool fulln = (g.N % 128) == 0 && !(e_fulln && e_fulln[0] == '0');   // four-wave 128 x 128 tiles only
    if (four_wave_wide) hipLaunchKernelGGL((gemm_f16x3_kernel<2, true, 32, 4>), grid, dim3(256), 0, st, a, n_col_tiles, n_blocks);
    else if (wide && k64) hipLaunchKernelGGL((gemm_f16x3_kernel<4, true, 64>), grid, dim3(512), 0, st, a, n_col_tiles, n_blocks);
    else if (wide) hipLaunchKernelGGL((gemm_f16x3_kernel<4, true>), grid, dim3(512), 0, st, a, n_col_tiles, n_blocks);
    else if (k64 && fulln) hipLaunchKernelGGL((gemm_f16x3_kernel<2, true, 64, 2, true>), grid, dim3(256), 0, st, a, n_col_tiles, n_blocks);
    else if (k64) hipLaunchKernelGGL((gemm_f16x3_kernel<2, true, 64>), grid, dim3(256), 0, st, a, n_col_tiles, n_blocks);
    else if (fulln) hipLaunchKernelGGL((gemm_f16x3_kernel<2, true, 32, 2, true>), grid, dim3(256), 0, st, a, n_col_tiles, n_blocks);
    else hipLaunchKernelGGL((gemm_f16x3_kernel<2, true>), grid, dim3(256), 0, st, a, n_col_tiles, n_blocks);
  } else {
    if (wide) hipLaunchKernelGGL((gemm_f16x3_kernel<4, false>), grid, dim3(512), 0, st, a, n_col_tiles, n_blocks);
    else hipLaunchKernelGGL((gemm_f16x3_kernel<2, false>), grid, dim3(256), 0, st, a, n_col_tiles, n_blocks);
  }
  if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

}  // namespace nsr
