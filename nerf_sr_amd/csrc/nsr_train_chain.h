// Chain kernels of the training step (internal; see nsr_mlp_f16.hip TRAIN and nsr_train_chain.hip).
#pragma once
#include "nsr_common.h"

// bytes of one panel set (twelve fp16 panels, see nsr_f16x3_core.h) for P sample points
extern "C" NSR_INTERNAL int64_t nsr_f16x3_train_panel_bytes(int64_t P);
// forward pass: raw (R N, 4) = (rgb, sigma) + the activation panels; `packed` = nsr_f16x3_pack of the weights;
// status: the step's sticky NSR_FLAG_* word (input / activation range, non-finite outputs) or null
extern "C" NSR_INTERNAL int nsr_f16x3_train_forward(const void* packed, const float* rays, int ray_stride, const float* z, int64_t R,
                                                    int N, float* raw, void* pan, unsigned* sgn, unsigned* status, void* stream);
// NSR_FLAG_WEIGHT_RANGE into *word if a weight cannot be carried at `precision` (nsr_mlp.hip)
extern "C" NSR_INTERNAL int nsr_check_weights_range(const float* const* w, int precision, unsigned* word, void* stream);
// dwords of the sign panels (one bit per pre-activation) for P sample points
extern "C" NSR_INTERNAL int64_t nsr_f16x3_train_sign_words(int64_t P);
extern "C" NSR_INTERNAL size_t nsr_f16x3_packed_bytes(void);
extern "C" NSR_INTERNAL int nsr_f16x3_pack(const float* const* w, void* packed_dev, void* stream);
// backward chain: transposed weight stream, then d(rgb_pre) (P, stride) / d(sigma) (P, stride) -> gradient panels;
// gmax[10] (device): float bits of the largest TRUE magnitude of each gradient panel (zeroed, then atomicMax);
// pscale (10, ceil(P / 128) * 128): per panel and point the power of two that turns the stored fp16 values into true gradients
extern "C" NSR_INTERNAL size_t nsr_chain_bwd_packed_bytes(void);
// terms (round 6): MFMAs per product of the chain -- 3 = W_hi g_hi + W_hi g_lo + W_lo g_hi (fp32-grade), 2 = W_hi g_hi + W_lo g_hi,
// 1 = W_hi g_hi (the stream then carries the hi pieces only); pack and run must agree
extern "C" NSR_INTERNAL int nsr_chain_bwd_pack(const float* const* w, void* packed_dev, int stop_grad, int terms, void* stream);
extern "C" NSR_INTERNAL int nsr_chain_bwd(const void* packed, const unsigned* sgn, void* dpan, const float* d_rgb, int d_rgb_stride,
                                          const float* d_sigma, int d_sigma_stride, int64_t P, unsigned* gmax,
                                          float* pscale, int terms, int gmax_is_zero, void* stream);
// round 6, the step's launch count: both networks per launch; the backward pack also clears `n_zero` doubles at `zero` and
// writes `value` to `word` (the step's loss carries and its option word: two memset launches less)
extern "C" NSR_INTERNAL int nsr_check_weights_range2(const float* const* w0, const float* const* w1, int precision, unsigned* word, void* stream);
extern "C" NSR_INTERNAL int nsr_f16x3_pack2(const float* const* w0, void* packed0, const float* const* w1, void* packed1, void* stream);
extern "C" NSR_INTERNAL int nsr_chain_bwd_pack2(const float* const* w0, void* packed0, const float* const* w1, void* packed1, int stop_grad,
                                                int terms, double* zero, int n_zero, unsigned* word, unsigned value, void* stream);
