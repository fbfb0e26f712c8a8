// fp32-MFMA GEMM used by the training step: C (M x N) = epilogue(sum_k A(i, k) B(j, k)).
// Each operand is addressed either K-contiguous (element (i, k) at p[i * ld + k]) or K-major (p[k * ld + i]), so
// the three products of a linear layer (nn.Linear, models/networks.py:150-170) run on ROW-MAJOR activations and the
// nn.Linear weight layout as they are -- nothing is transposed in memory, nothing is stored twice:
//   forward   Y (P x N)  = act(X (P x K) W (N x K)^T + b)     A = X  K-contiguous,  B = W  K-contiguous
//   dgrad     dX (P x K) = (dY (P x N) W) * [X > 0]           A = dY K-contiguous,  B = W  K-major   (k = n)
//   wgrad     dW (N x K) = sum_p dY[p][n] X[p][k]             A = dY K-major,       B = X  K-major   (k = p, split)
#pragma once
#include "nsr_common.h"

namespace nsr {

enum GemmAct { kActNone = 0, kActRelu = 1, kActSigmoid = 2, kActTanh = 3 };

struct GemmArgs {
  const float* A; int64_t lda; int a_kmajor;   // M x K
  const float* B; int64_t ldb; int b_kmajor;   // N x K
  float* C; int64_t ldc;            // M x N (may be null if Ct is given)
  float* Ct; int64_t ldct;          // N x M, the same values transposed (may be null)
  const float* bias;                // N (may be null)
  const float* mask; int64_t ldm;   // M x N (may be null): result *= (mask > 0)
  int64_t M; int N; int64_t K;      // K % 32 == 0; rows / columns past M / N are computed on clamped data and dropped
  int n_valid;                      // only columns < n_valid are written (<= N)
  int act;                          // GemmAct, applied after the bias, before the mask
  int splits;                       // > 1: split-K, block z handles K range z; raw sums go to C + z * split_stride
  int64_t split_stride;             //      (bias / act / mask must be off; Ct unused)
  float* col_sums;                  // may be null: (ceil(M / 128), N) per-row-tile column sums of the values written
                                    // (the bias gradient of the layer whose pre-activation gradient this GEMM makes)
  float acc_scale;                  // 0 or 1: off; else the accumulator is multiplied by it before the bias (removes the
                                    // power-of-two scale of pre-scaled split-fp16 weights, kSplitScale)
};

// Weights handed to gemm_f16x3 are multiplied by 2^6 before they are split into fp16 (hi, lo) halves (exact), and the
// consumer sets acc_scale = 2^-6: the lo half of a typical weight (|w| ~ 0.02-0.05 -> lo ~ 2^-17) otherwise sits on
// fp16's subnormal floor (2^-24) and loses three of its eleven bits.  |w| < 1023 stays inside fp16.
constexpr float kSplitScale = 64.0f, kSplitInvScale = 1.0f / 64.0f;

// enqueue; returns NSR_OK / NSR_ERR_*.  K-major operands need their non-K extent to be a multiple of 4.
NSR_INTERNAL int gemm(const GemmArgs& g, hipStream_t st);

// ---- split-fp16 variant (nsr_gemm_f16.hip): the same product on v_mfma_f32_32x32x16_f16 with every fp32 value carried
// as hi + lo fp16 halves (a b = a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate: products exact to ~2^-21, the
// scheme of the NSR_F16X3 inference kernel) at 3/16 of the fp32-MFMA cycle cost.  Forward products only: A (fp32,
// K-contiguous, values within fp16 range) is split while it is staged into LDS; B (weights, N x K, K-contiguous) is
// given pre-split (split_f16).  With `conv.cin > 0` A is not a matrix but an NHWC activation and the rows of the 3 x 3 /
// pad 1 im2col matrix (K = 9 cin, taps-major) are gathered on the fly (cin % 32 == 0): g.A = activation base,
// g.lda = its row stride in floats, g.M = n_img * Ho * Wo.
struct ConvGather {
  int cin, Hs, Ws, Ho, Wo, stride, up;   // source height / width (before the optional nearest x2 upsample), output size
};
struct GemmF16Args {
  GemmArgs g;                       // g.B / g.ldb unused; a_kmajor = b_kmajor = 0, splits <= 1, Ct unsupported
  const unsigned short* Bh;         // N x K fp16 bit patterns: high parts
  const unsigned short* Bl;         //                          low parts
  int64_t ldbh;                     // row stride of Bh / Bl in halves (multiple of 8)
  ConvGather conv;
  // PRE-SPLIT activations ("planes"): an fp32 tensor stored as two fp16 arrays of the same shape, hi = RNE(v) and
  // lo = RNE(v - hi).  Ah != null: A is given that way (hi plane Ah, lo plane Ah + a_plane, g.lda = row stride in
  // halves, multiple of 8; g.A unused) and is staged into LDS without any conversion -- a 3 x 3 convolution re-reads
  // every activation nine times, so splitting it once in the producer's epilogue instead of at every staging removes
  // 8/9 of that VALU work.  Ch != null: the result is WRITTEN that way (hi plane Ch, lo plane Ch + c_plane, g.ldc in
  // halves, even; n_valid even; g.C unused; col_sums unsupported).
  const unsigned short* Ah;
  int64_t a_plane;
  unsigned short* Ch;
  int64_t c_plane;
  // GROUPED ROWS (conv gather + plane output only; group == 8 or 0): the GEMM's row index m runs member-fastest over groups
  // of 8 images, m = (b * px + pixel) * 8 + r with px = Ho * Wo, i.e. the 8 rows 8 q .. 8 q + 7 are one output pixel of the
  // 8 images b * 8 + r.  Gather and per-image output use image b * 8 + r as usual (output row (b * 8 + r) * px + pixel);
  // additionally the epilogue writes the maximum over the 8 members to row q of the planes Mh (hi) / Mh + m_plane (lo),
  // row stride ldm halves -- torch.max over the reference patches (networks.py:980-983) without a pass of its own.
  int group;
  unsigned short* Mh;
  int64_t m_plane, ldm;
  // STREAM-ORDERED copy of B for conv_halo_kernel (nsr_gemm_f16.hip), or null (then that kernel is not used): 1-KiB units
  // [column block of 32][k-step = channel chunk cc of 16 x tap: cc * 9 + tap][hi, lo][lane li + 32 h][8 halves], the
  // halves of a lane = B[32 nb + li][tap * cin + 16 cc + 8 h .. + 8) -- one LDS-DMA piece = one contiguous KiB instead of
  // 32 rows x 32 B.  Only for conv.cin % 16 == 0 and N % 32 == 0 (no padding columns inside K).
  const unsigned short* Bs;
};
NSR_INTERNAL int gemm_f16x3(const GemmF16Args& a, hipStream_t st);
// v = kSplitScale * w[i];  hi[i] = fp16(v), lo[i] = fp16(v - hi[i])   (round to nearest)
NSR_INTERNAL int split_f16(const float* w, int64_t n, unsigned short* hi, unsigned short* lo, hipStream_t st);

// ---- weight gradient from two training panels on the fp16 MFMA (nsr_wgrad_f16.hip): partial[z] (M x N) =
// sum over slice z of the points of A[p][0..M) B[p][0..N)^T, M = 256 or 128, N = 256 or 64, both operands 2-byte panels
// in the chain kernels' unit layout (nsr_f16x3_core.h).  A's stored values x a_pscale[p] (a power of two per point) are the
// true gradients; they are brought into fp16's range by a further power of two derived from *a_max_bits (float bits of the
// panel's largest true magnitude, device memory), removed again before the store.
struct WgradArgs {
  const void* A; int64_t a_gbytes; int M;     // gradient panel, bytes between two point groups (64 M)
  const void* B; int64_t b_gbytes; int N;     // forward panel (64 N)
  const unsigned* a_max_bits;
  const float* a_pscale;                      // (points): A's stored values x a_pscale = the true values
  float* partial; int64_t split_stride;       // slot s of the product: partial + s * split_stride (M x N floats)
  float* row_sums;                            // may be null: slot s at row_sums + s * M: sums of A's rows over the slice (true scale)
};

// All products of one network pass as ONE launch (nsr_wgrad_f16.hip, wgrad_jobs_kernel).  Fill j[0..n).w (every product
// has jobs.n_groups point groups), call wgrad_jobs_plan, hand out the partial / row-sum slots (n_slots of each job),
// launch with the workgroup count the plan returned.
struct WgradJob {
  WgradArgs w;
  int64_t cost0;      // start of the product in the work list (plan)
  int cost;           // M + N: panel rows read per point group (plan)
  int w_first;        // first workgroup that touches the product (plan)
  int n_slots;        // workgroups that touch it = partial slots it owns (plan)
};
constexpr int kMaxWgradJobs = 16;
struct WgradJobs {
  WgradJob j[kMaxWgradJobs];
  int n;
  int64_t n_groups, total_cost, per_wg;
};
NSR_INTERNAL int wgrad_jobs_plan(WgradJobs& jobs, int64_t P, int n_wg);
NSR_INTERNAL int wgrad_jobs_f16(const WgradJobs& jobs, int n_wg, hipStream_t st);

}  // namespace nsr
