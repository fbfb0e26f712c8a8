/* nsr_train.h — C ABI of the TRAINING step through the NeRF-SR render path.
 *
 * Same conventions as nsr.h: raw DEVICE pointers, caller-owned memory, work enqueued on the caller's HIP
 * stream, int status returns, no global mutable state.  The reference interface replaced here is
 * NeRFDownXModel.optimize_parameters (models/nerf_downX_model.py:398-408):
 *     forward()  [train mode: randomized sampling, density noise]   -> nsr_train_loss_and_grads
 *     comp_low_res_output() + calculate_losses() + loss_tot.backward()  -> nsr_train_loss_and_grads
 *     optimizer.step()  [torch.optim.Adam, :201-204]                 -> nsr_adam_step
 * Weights, gradients and Adam moments are the 24 tensors of VanillaMLP.state_dict() in nn.Linear layout
 * (see NSR_N_STATE_TENSORS in nsr.h), handed over as HOST arrays of 24 DEVICE pointers, so a binding can
 * pass `p.data_ptr()` / `p.grad.data_ptr()` of the reference's own parameters.
 *
 * Arithmetic: fp32 storage and accumulation throughout, like the reference.  precision = NSR_FP32: every contraction on
 * v_mfma_f32_32x32x2_f32, layer by layer.  precision = NSR_F16X3: the network as two fused chain launches per pass
 * (DESIGN 7.1): forward pass and input gradients on split-fp16 v_mfma_f32_32x32x16_f16 with fp32-grade products (hi/lo
 * operands, three MFMAs per product; gradients scaled by exact powers of two per point); the WEIGHT gradients contract
 * the chain's fp16 `hi` operands (activation and input gradient rounded to 11 bits, one MFMA per product, fp32
 * accumulation over the sample points: a gradient tensor moves by < 1e-4 of its norm, less than fp32-vs-fp64 of the
 * reference's own arithmetic).  precision = NSR_F16X3_GEMM: layer-by-layer GEMMs, forward products split-fp16, every
 * gradient on the fp32 MFMA (round 1's design; the chain path's A/B partner).
 */
#ifndef NSR_TRAIN_H_
#define NSR_TRAIN_H_

#include "nsr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A third `precision` value, valid for the training entry points only (nsr_precision in nsr.h holds the others). */
#define NSR_F16X3_GEMM 18
/* The chain path with the arithmetic of its BACKWARD chain (the input gradients dz_{l-1} = W_l^T dz_l) named explicitly
 * (round 6; forward pass and weight gradients as under NSR_F16X3 in all three):
 *   NSR_F16X3_BWD3: W_hi g_hi + W_hi g_lo + W_lo g_hi, three MFMAs per product, fp32-grade (rounds 2-5's arithmetic);
 *   NSR_F16X3_BWD2: W_hi g_hi + W_lo g_hi: the gradient entering a layer is its fp16 `hi` alone (11 bits, scaled per point by a
 *                   power of two), the weights keep their 22 bits;
 *   NSR_F16X3_BWD1: W_hi g_hi: one MFMA per product, both operands rounded to 11 bits;
 *   NSR_F16X3_BWDM: mixed -- two terms (as _BWD2) on the six layers nearest the output (dir_encoding, xyz_encoding_final,
 *                   xyz_encoding_8 .. _5), one (as _BWD1) on the three below (xyz_encoding_4 .. _2), whose rounding passes
 *                   through the fewest further layers.
 * NSR_F16X3 selects the cheapest of them whose gradients stay inside the bounds of tests/test_gpu_train.py (every gradient
 * tensor within 2e-3 of its norm of the fp64 oracle, 5e-4 on the heads, a 200-step Adam trajectory no further from the fp32
 * run than another fp32-grade implementation is; the whole gradient within 2e-4 of the fp32-gradient path at 393,216 sample
 * points): NSR_F16X3_BWDM as of NSR_VERSION 131 (measured 4.0e-4 / 7e-5 on the heads / whole gradient 1.0e-4; NSR_VERSION 130:
 * NSR_F16X3_BWD2, 2.0e-4 / 7e-5 / 1.8e-5).  NSR_F16X3_BWD1 is a stated FAST path: it holds the per-tensor bounds (measured
 * 6.6e-4 / 2.2e-4 on the heads) and the trajectory bound, not the whole-gradient one (3.1e-4) -- measured: DESIGN.md 7.1. */
#define NSR_F16X3_BWD3 19
#define NSR_F16X3_BWD2 20
#define NSR_F16X3_BWD1 21
#define NSR_F16X3_BWDM 22

/* Workspace for one pass over `ray_chunk` rays (activations of one network, gradient buffers, padded weight copies,
 * split-K partials; for NSR_F16X3 the activation / gradient panels, ~24 KB per sample point in all).  0 on invalid
 * arguments. */
size_t nsr_train_workspace_bytes(int64_t ray_chunk, int n_coarse, int n_importance);
/* The same for the path `precision` selects (what nsr_train_loss_and_grads checks): NSR_F16X3 -> the chain path's 2-byte
 * panels, sign words and weight streams without the per-layer activation / gradient matrices of the GEMM path (~11 KB per
 * sample point); NSR_FP32 / NSR_F16X3_GEMM -> the GEMM path's buffers without the panels (~13 KB).
 * nsr_train_workspace_bytes is the union: sufficient whatever runs.  Both depend on their arguments only. */
size_t nsr_train_workspace_bytes_for(int precision, int64_t ray_chunk, int n_coarse, int n_importance);

/* Losses and d(loss_tot)/d(weights) of one batch.
 *   loss_tot = lambda_coarse * mse(mean_s2(coarse rgb), target) + lambda_fine * mse(mean_s2(fine rgb), target)
 *   (calculate_losses, nerf_downX_model.py:355-362; nn.MSELoss(reduction='mean'), criterions.py:7-15).
 * rays (R, ray_stride) LR-pixel-major with s2 sub-rays per LR pixel (s2 = 1: the vanilla `nerf` model);
 * target_lr (R / s2, 3).
 * Random draws are INPUTS (the reference draws them with torch.rand_like / torch.randn_like / torch.rand,
 * models/utils.py:37-41, 72-73, 199-212): u_coarse (R, Nc), u_fine (R, Ni) uniform in [0, 1); noise_coarse
 * (R, Nc), noise_fine (R, Nc + Ni) standard normal, scaled by noise_std.  NULL selects the deterministic
 * branch of the corresponding stage (randomized = False / noise off).
 * g_coarse / g_fine: 24 gradient tensors each, OVERWRITTEN.
 * precision: NSR_FP32 = every product on the fp32 MFMA; NSR_F16X3 = the chain kernels (forward and input gradients on the
 * split-fp16 MFMA, exact to ~2^-21, the inference path's scheme; weight gradients on one fp16 MFMA per product);
 * NSR_F16X3_GEMM = round 1's variant: per-layer GEMMs, forward products split-fp16, gradients on the fp32 MFMA.  The
 * path is a function of this argument alone: no environment variable, no process state.
 * ray_chunk: rays per pass (bounds the workspace; multiple of s2; 0 = R); gradients and losses of the passes
 * are accumulated, the result does not depend on the chunking beyond fp32 summation order.  Every pass -- the
 * shorter last one included -- must hold a multiple of 32 sample points in both networks
 * (rays_in_pass * n_coarse and rays_in_pass * (n_coarse + n_importance) divisible by 32: the weight-gradient
 * GEMM contracts over the points in K tiles of 32); otherwise NSR_ERR_UNSUPPORTED is returned before anything
 * is enqueued (outputs and gradients untouched).  All of the reference's scripts (64 + 64 samples) satisfy it
 * for any ray count.
 * render_flags (`white_bkgd` until NSR_VERSION 120): a word of options like the render path's (include/nsr.h): NSR_WHITE_BKGD and NSR_SIGMA_SOFTPLUS as there
 * (the backward pass carries sigmoid(sigma - 1) instead of [sigma > 0]), and the colour head's two, which the render
 * path keeps in the packed network: NSR_TRAIN_GAMMA_CORRECT = --gamma_correct while training (render_rays returns
 * pow(rgb, 1 / 2.2) per sample, nerf_downX_model.py:271-276; the colours are corrected between the network and the
 * compositor and the backward pass carries the slope y (1 - y^2.2) / 2.2 of the corrected sigmoid);
 * NSR_TRAIN_COLOR_NONE = --color_activation none (models/networks.py:173-180; slope 1); NSR_TRAIN_STOP_GRAD =
 * --stop_grad true (models/networks.py:218-219: the colour branch's input is detached -- xyz_encoding_final gets zero
 * gradients, the trunk the density head's alone).  Gamma and none together are
 * NSR_ERR_UNSUPPORTED (the power of an unbounded head is NaN for every negative value); any other bit NSR_ERR_INVALID_ARG.
 * outs: the 8 forward outputs in nsr_forward_rays order (entries may be NULL except the two comp_rgbs).
 * lr_coarse / lr_fine: (R / s2, 3) s2-means (comp_low_res_output, :326-348); losses: DEVICE float[2] =
 * { lambda_coarse * mse_coarse, lambda_fine * mse_fine }. */
#define NSR_TRAIN_GAMMA_CORRECT 4
#define NSR_TRAIN_COLOR_NONE 8
#define NSR_TRAIN_STOP_GRAD 16
int nsr_train_loss_and_grads(const float* const* w_coarse, const float* const* w_fine, float* const* g_coarse,
                             float* const* g_fine, const float* rays, int ray_stride, int64_t R, int s2,
                             const float* target_lr, int n_coarse, int n_importance, int render_flags, int lindisp,
                             const float* u_coarse, const float* u_fine, const float* noise_coarse,
                             const float* noise_fine, float noise_std, float lambda_coarse, float lambda_fine,
                             int precision, int64_t ray_chunk, float* const* outs, float* lr_coarse, float* lr_fine, float* losses,
                             void* workspace, size_t workspace_bytes, void* stream);

/* The optional variance losses of comp_low_res_output / calculate_losses (nerf_downX_model.py:332-336, 349-353, 374-378;
 * options --use_var_loss / --use_depth_var_loss, :107-112):
 *   loss_tot += lambda_coarse_var * sum_{LR pixels, channels} var_{s2 sub-rays}(coarse rgb) + lambda_fine_var * (fine rgb)
 *   loss_tot += lambda_coarse_depth_var * sum_{LR pixels} var_{s2 sub-rays}(coarse depth / far) + lambda_fine_depth_var * (fine)
 * with torch.var's default (unbiased: divisor s2 - 1) and `far` = the reference's self.far (:284: the far bound of the batch's
 * first ray).  A lambda of 0 switches its term off.  s2 must be >= 2 when any lambda is non-zero (the variance of one sub-ray
 * is 0 / 0: the reference would train on NaN) -- NSR_ERR_INVALID_ARG otherwise. */
typedef struct nsr_train_var_losses {
  float lambda_coarse_var, lambda_fine_var;
  float lambda_coarse_depth_var, lambda_fine_depth_var;
  float far;
} nsr_train_var_losses;
/* nsr_train_loss_and_grads with those terms in the loss and in the gradients.  var_losses (may be NULL): DEVICE float[4] =
 * { lambda_coarse_var * var_c, lambda_fine_var * var_f, lambda_coarse_depth_var * dvar_c, lambda_fine_depth_var * dvar_f },
 * the reference's loss_out_{coarse,fine}_var / loss_{coarse,fine}_depth_var times their lambdas.  Every other argument as above. */
int nsr_train_loss_and_grads_var(const float* const* w_coarse, const float* const* w_fine, float* const* g_coarse,
                                 float* const* g_fine, const float* rays, int ray_stride, int64_t R, int s2,
                                 const float* target_lr, int n_coarse, int n_importance, int render_flags, int lindisp,
                                 const float* u_coarse, const float* u_fine, const float* noise_coarse,
                                 const float* noise_fine, float noise_std, float lambda_coarse, float lambda_fine,
                                 int precision, int64_t ray_chunk, float* const* outs, float* lr_coarse, float* lr_fine, float* losses,
                                 void* workspace, size_t workspace_bytes, void* stream, const nsr_train_var_losses* var,
                                 float* var_losses);

/* Numerics status of the training step (the reference drops into pdb on NaN colours, nerf_downX_model.py:273-274; a
 * replacement reports instead).  The FIRST 64 BYTES of the workspace are a sticky status block: with NSR_F16X3 every
 * nsr_train_loss_and_grads ORs NSR_FLAG_WEIGHT_RANGE (a weight of the iteration's re-pack is non-finite or |w| >= 1023.75:
 * the split-fp16 stream cannot carry it), NSR_FLAG_INPUT_RANGE / NSR_FLAG_ACTIVATION_RANGE (a raw coordinate or a hidden
 * activation left the fp16 operand range: products degrade to 11 bits) and NSR_FLAG_OUTPUT_NONFINITE into its first word,
 * exactly as the inference entry points do into a packed network's tail (nsr.h).  Nothing is synchronised by the step.
 * nsr_train_status_reset zeroes the block (call it once when the workspace is allocated: the step never clears it);
 * nsr_train_status copies the word to the host (waits for the stream) and optionally clears it.  NSR_FP32 and
 * NSR_F16X3_GEMM raise nothing (the word stays as nsr_train_status_reset left it). */
int nsr_train_status_reset(void* workspace, void* stream);
int nsr_train_status(void* workspace, int clear, unsigned* flags_out, void* stream);

/* torch.optim.Adam (no weight decay, no amsgrad) on the 24 tensors of one network, in place:
 *   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;
 *   w -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)        (step counts from 1). */
int nsr_adam_step(float* const* w, const float* const* g, float* const* m, float* const* v, int step, float lr,
                  float beta1, float beta2, float eps, void* stream);

/* One nn.Linear (+ activation) on the training GEMM, exposed for testing the kernel on its own:
 *   y (P, N) = act(x (P, K) · w (N, K)^T + b),  act: 0 none, 1 relu, 2 sigmoid;  y_t (N, P) optional transpose.
 * K % 32 == 0, ldx % 4 == 0, ldw % 4 == 0, 16-byte aligned pointers (the training step pads its operands). */
int nsr_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b, int act, float* y,
               int64_t ldy, float* y_t, int64_t ldyt, int64_t P, int K, int N, void* stream);

/* The same nn.Linear on the split-fp16 MFMA (round 6: what a NON-default network architecture runs on layer by layer under
 * precision f16x3, ops.GenericMLP): three v_mfma_f32_32x32x16_f16 per product, x split into (hi, lo) fp16 halves on its way
 * to LDS, the weights pre-split ONCE by nsr_split_weights -- w_hi / w_lo (n halves each) = RNE_f16(64 w) and RNE_f16(64 w - hi);
 * the factor is undone in the epilogue -- products exact to ~2^-21, fp32 accumulation, fp32 in and out.  Same shape rules as
 * nsr_linear (K % 32 == 0, ldx % 4 == 0) plus ldw % 8 == 0 and 16-byte aligned w_hi / w_lo; no transposed output. */
int nsr_split_weights(const float* w, int64_t n, void* w_hi, void* w_lo, void* stream);
int nsr_linear_f16x3(const float* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw, const float* b, int act,
                     float* y, int64_t ldy, int64_t P, int K, int N, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_TRAIN_H_ */
