/* nsr_refine.h — C ABI of the refinement network's forward pass.
 *
 * Replaces MaxPoolingModel.forward(x_synth, list_x_candi) in eval mode (models/networks.py:735-990: the
 * Model_VNPCAT encoder on the synthesised patch and on each of its R reference patches, a max over the references
 * at every scale, the Model_VNPCAT decoder), as refine_model.py:99 calls it (`self.netRefine(sr_patch, ref_patches)`).
 * fp32 in / fp32 out; every 3x3 convolution runs as a GEMM (NSR_FP32: im2col + the fp32-MFMA GEMM of the training step,
 * nsr_gemm.hip; NSR_F16X3: the split-fp16 GEMM with implicit im2col on pre-split fp16 (hi, lo) activation planes) with
 * bias / ReLU / tanh in its epilogue; BatchNorm (eval: running statistics, eps 1e-5) is folded into the packed
 * weights; activations are NHWC so that channel concatenations and nearest-neighbour upsampling cost nothing
 * (GEMM outputs land directly in the concatenated buffers, the upsampling is an index map of the next im2col).
 *
 * Weights: the float tensors of MaxPoolingModel.state_dict() in state_dict order WITHOUT the 17 int64
 * `num_batches_tracked` entries: per layer weight (Cout, Cin, 3, 3), bias (Cout) and, for layers followed by a
 * BatchNorm2d, its weight, bias, running_mean, running_var.  Layer order (NSR_REFINE_N_LAYERS = 19):
 *   E.conv1 (no BN), E.conv2 .. E.conv7, D.conv1, D.conv2, D.conv2_up, D.conv3, D.conv4, D.conv4_up, D.conv5, D.conv6,
 *   D.conv6_up, D.conv7, D.conv8, D.conv9 (no BN)        ->  NSR_REFINE_N_TENSORS = 2 * 19 + 4 * 17 = 106.
 */
#ifndef NSR_REFINE_H_
#define NSR_REFINE_H_

#include "nsr.h"

#define NSR_REFINE_N_LAYERS 19
#define NSR_REFINE_N_TENSORS 106

#ifdef __cplusplus
extern "C" {
#endif

/* precision: NSR_FP32 (explicit im2col + fp32-MFMA GEMM) or NSR_F16X3 (split-fp16 MFMA, products exact to ~2^-21, with
 * implicit im2col: no col matrix is materialised); 0 / NSR_ERR_UNSUPPORTED for anything else.
 * The blob holds, per layer, the folded weights (row-major, K = tap-major), the folded bias and -- for the layers the
 * LDS-patch convolution kernel can take -- a second, fragment-ordered copy of the weights; its size is
 * whatever this function returns, the layout is private to the library.
 * Non-finite values travel as in torch: ReLU keeps NaN, the max over the reference patches propagates it, tanh(NaN) = NaN --
 * a NaN input pixel comes out as NaN over its receptive field, nothing is masked (there is no status word on this path). */
size_t nsr_refine_packed_bytes(int precision);
/* tensors: HOST array of NSR_REFINE_N_TENSORS DEVICE pointers (order above); packed: DEVICE, 16-byte aligned */
int nsr_refine_pack_weights(const float* const* tensors, void* packed, int precision, void* stream);

/* H and W multiples of 8 (three stride-2 levels); 0 on invalid arguments.  The plain form is the NSR_FP32 size (the larger
 * one: that mode materialises im2col matrices); `_for` gives the size a precision actually needs (NSR_F16X3: activations
 * + 128 bytes per input pixel for the first layer -- its input staged as pre-split 16-channel planes when the patch is whole
 * 16 x 16 blocks, a 32-column im2col matrix otherwise --, ~21 MB per 64 x 64 patch set with 8 references). */
size_t nsr_refine_workspace_bytes(int B, int R, int H, int W);
size_t nsr_refine_workspace_bytes_for(int precision, int B, int R, int H, int W);
/* x_synth (B, 3, H, W), x_candi (B, R, 3, H, W), out (B, 3, H, W): NCHW fp32 DEVICE (the reference's tensors) */
int nsr_refine_forward(const void* packed, int precision, const float* x_synth, const float* x_candi, int B, int R, int H,
                       int W, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---- `--not_use_ref` (MaxPoolingModel with Model_VNPCAT_Decoder_NoPooling, networks.py:866-945, 958-969): the encoder
 * runs on the synthesised patch only and the decoder's concatenations carry no F_max_i channels, i.e. D.conv1, D.conv3,
 * D.conv5, D.conv7 have 512 / 1024 / 512 / 256 input channels.  Same 106 tensors in the same order (four of them with
 * those shapes), its own packed blob; workspace: nsr_refine_workspace_bytes_for(precision, B, 1, H, W). */
size_t nsr_refine_packed_bytes_noref(int precision);
int nsr_refine_pack_weights_noref(const float* const* tensors, void* packed, int precision, void* stream);
int nsr_refine_forward_noref(const void* packed, int precision, const float* x_synth, int B, int H, int W, float* out,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- patch tiler / stitcher around the network (the 'test' path of data/llff_refine_dataset.py:303-340 and
 * models/refine_model.py:205-216): an SR image is cut into `patch` x `patch` tiles on a grid (x outer, y inner, starts
 * clamped to size - patch); each tile gets `n_ref` reference patches whose top-left corners are the first `n_ref`
 * in-image warp targets locs[y][x] (nsr_depth_warp / `{i}_locs.npz`) of the tile's pixels scanned x outer / y inner,
 * clamped likewise; missing ones are the SR tile itself (ref start -1); predictions are pasted back in tile order.
 * n = ceil(W / patch) * ceil(H / patch) tiles.  All pointers DEVICE. */
int nsr_refine_tile(const double* locs, int H, int W, int patch, int n_ref, int* starts, int* ref_starts, void* stream);
int nsr_refine_gather(const float* sr_img, const float* ref_img, int H, int W, int patch, int n_ref, const int* starts,
                      const int* ref_starts, int n, float* sr_patch, float* ref_patches, void* stream);
int nsr_refine_stitch(const float* patches, const int* starts, int n, int patch, int H, int W, float* image, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_REFINE_H_ */
