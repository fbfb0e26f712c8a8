// fp32-MFMA GEMM used by the training step: C = epilogue(A · B^T), both operands K-contiguous.
// One kernel serves the three products of a linear layer (nn.Linear, models/networks.py:150-170):
//   forward   Y (P x N)  = act(X (P x K) · W (N x K)^T + b)                  A = X,    B = W
//   dgrad     dX (P x K) = (dY (P x N) · W) ⊙ [X > 0]                        A = dY,   B = W^T (K x N)
//   wgrad     dW (N x K) = dY^T (N x P) · X^T (K x P)   (split over P)        A = dY^T, B = X^T
// which is why every activation / activation-gradient is kept in both orientations (the epilogue writes C and
// C^T): no operand ever needs a transposing load.
#pragma once
#include "nsr_common.h"

namespace nsr {

enum GemmAct { kActNone = 0, kActRelu = 1, kActSigmoid = 2 };

struct GemmArgs {
  const float* A; int64_t lda;      // M x K, K contiguous
  const float* B; int64_t ldb;      // N x K, K contiguous
  float* C; int64_t ldc;            // M x N (may be null)
  float* Ct; int64_t ldct;          // N x M, the same values transposed (may be null)
  const float* bias;                // N (may be null)
  const float* mask; int64_t ldm;   // M x N (may be null): result *= (mask > 0)
  int64_t M; int N; int64_t K;      // K % 32 == 0; rows / columns past M / N are computed on clamped data and dropped
  int n_valid;                      // only columns < n_valid are written (<= N)
  int act;                          // GemmAct, applied after the bias, before the mask
  int splits;                       // > 1: split-K, block z handles K range z; raw sums go to C + z * split_stride
  int64_t split_stride;             //      (bias / act / mask must be off; Ct unused)
};

// enqueue; returns NSR_OK / NSR_ERR_*
NSR_INTERNAL int gemm_nt(const GemmArgs& g, hipStream_t st);

}  // namespace nsr
