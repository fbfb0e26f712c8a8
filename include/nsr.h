/*
 * nsr.h — C ABI of libnsr: the MI355X-native (gfx950) supersampled volumetric
 * render hot path of NeRF-SR.
 *
 * The reference (cwchenwang/NeRF-SR) has no FFI: the path sits behind Python
 * functions / nn.Module.__call__ on fp32 tensors.  Each entry
 * point below replaces one of those call sites; the reference-side binding a
 * maintainer would add is a ctypes stub.  Citations are
 * file:line in the reference tree.
 *
 * Conventions
 *   - every pointer named *_dev / rays / z / out ... is a DEVICE pointer to
 *     contiguous fp32 unless stated otherwise; the caller (PyTorch) owns all
 *     buffers and pre-allocates outputs; the library allocates nothing.
 *   - `stream` is a hipStream_t passed as void*; work is only ENQUEUED on it,
 *     the library never synchronises the device or the host.
 *     Two documented exceptions, both off the per-frame path: nsr_pack_weights (load time) and
 *     nsr_weights_status wait for `stream` because they hand a device-side verdict back to the host.
 *   - every function returns NSR_OK (0) or a negative nsr_status; no exceptions,
 *     no global mutable state: re-entrant from any number of host threads
 *     (one per GPU under nn.DataParallel, models/networks.py:67).
 *   - the device is the caller's current HIP device (hipSetDevice).
 *   - ray records: `rays` is (R, ray_stride) fp32 with ray_stride = 8: [o(3), d(3), near, far], the encoded
 *     view direction is d (nerf_downX, models/nerf_downX_model.py:282-286); or ray_stride = 11: the vanilla
 *     model's rows with the view direction in columns 8:11 (models/nerf_model.py:209-213).  8-wide rows
 *     must be 16-byte aligned.
 *   - fixed architecture of the path: D=8, W=256, skips=[4], deg_pos=10,
 *     deg_dir=4, dim_pos=dim_dir=dim_rgb=3 (models/networks.py:124-126,
 *     models/nerf_model.py:52-57) — the only one the reference's scripts use.
 */
#ifndef NSR_H_
#define NSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSR_VERSION 131 /* 0.1.3: training precisions NSR_F16X3_BWD3 / _BWD2 / _BWD1 / _BWDM (nsr_train.h); NSR_F16X3 trains with the mixed two / one-term backward chain; `white_bkgd` arguments renamed `render_flags` (same values) */

typedef enum nsr_status {
  NSR_OK = 0,
  NSR_ERR_INVALID_ARG = -1, /* null pointer, negative size, misaligned pointer */
  NSR_ERR_UNSUPPORTED = -2, /* sample count / degree / precision outside the built path */
  NSR_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after an enqueue */
  NSR_ERR_WORKSPACE = -4,   /* workspace smaller than nsr_forward_rays_workspace_bytes() */
  NSR_ERR_RANGE = -5        /* a weight is non-finite or outside the operand range of the chosen precision
                               (nsr_pack_weights); the reference's analogue is the NaN trap of
                               models/nerf_downX_model.py:273-274 */
} nsr_status;

/* arithmetic the MLP contraction runs in (everything else is always fp32) */
typedef enum nsr_precision {
  NSR_FP32 = 0,   /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (parity path) */
  NSR_BF16 = 1,   /* v_mfma_f32_32x32x16_bf16: bf16 operands (RNE), fp32 accumulate.  FAST path, NOT a
                     parity path: colours move by ~5e-3 (PSNR vs the reference ~45 dB)             */
  NSR_F16X3 = 2,  /* split-fp16 on v_mfma_f32_32x32x16_f16: hi*hi + hi*lo + lo*hi, fp32 accumulate;
                     products exact to ~2^-21 -> fp32-grade results at 3/16 of the fp32-MFMA cost  */
  NSR_F16 = 3     /* v_mfma_f32_32x32x16_f16: fp16 operands (RNE), fp32 accumulate.  FAST path, NOT a
                     parity path: colours move by ~6e-4 (PSNR vs the reference ~60 dB)             */
} nsr_precision;

/* number of tensors in VanillaMLP.state_dict(), in state_dict order:
 * xyz_encoding_{1..8}.0.{weight,bias}, xyz_encoding_final.{weight,bias},
 * dir_encoding.0.{weight,bias}, sigma.{weight,bias}, rgb.0.{weight,bias}
 * (models/networks.py:131-180; SURVEY.md §5 "Checkpoint / resume") */
#define NSR_N_STATE_TENSORS 24

int nsr_version(void);
const char* nsr_status_string(int status);

/* ---- weights ------------------------------------------------------------
 * Replaces: handing `model.netCoarse` / `model.netFine` (nn.Module holding the
 * 24-tensor state_dict) to render_rays (models/nerf_downX_model.py:260-268).
 * The packed blob is the MFMA-fragment-ordered weight stream the MLP kernel
 * consumes; it is created in caller-owned device memory and must be re-packed
 * after every optimiser step. */
size_t nsr_packed_weights_bytes(int precision);
/* w: HOST array of 24 DEVICE pointers (nn.Linear layout (out,in) row-major).
 * Checked: returns NSR_ERR_RANGE (the blob is written all the same) when a weight or bias is non-finite, or when
 * a weight leaves the operand range of `precision` -- NSR_F16X3 carries 2^6 w as an fp16 (hi, lo) pair and NSR_F16
 * carries w as fp16, so |w| must stay below 1023.75 / 65520; NSR_FP32 and NSR_BF16 have the fp32 range.  To return
 * that verdict the call WAITS for `stream` (a load-time call; one of the two synchronising entry points). */
int nsr_pack_weights(const float* const* w, void* packed_dev, int precision, void* stream);
/* The same, enqueue only (re-packing inside a training loop): the range verdict lands in the blob's status word as
 * NSR_FLAG_WEIGHT_RANGE and is read with nsr_weights_status. */
int nsr_pack_weights_async(const float* const* w, void* packed_dev, int precision, void* stream);

/* ---- numerics status word ------------------------------------------------------
 * Replaces the reference's `if isnan(out_rgbs).any(): pdb.set_trace()` (models/nerf_downX_model.py:273-274;
 * SURVEY 8b "Error conventions": a replacement reports an error instead).  Every packed blob ends in a 32-bit STICKY
 * status word (cleared by nsr_pack_weights*).  Every launch that evaluates the network through that blob ORs flags into
 * it -- nothing is synchronised, nothing is checked on the host, until the caller asks:
 *   NSR_FLAG_WEIGHT_RANGE       set while packing, see nsr_pack_weights
 *   NSR_FLAG_INPUT_RANGE        a ray origin / direction / depth (or an embedded input row) was non-finite, or -- split
 *                               and single fp16 paths -- a sample position / direction beyond fp16's 65,504
 *   NSR_FLAG_ACTIVATION_RANGE   NSR_F16X3: a hidden activation left the range in which the (hi, lo) split keeps its 22
 *                               bits (|h| >= 1023.75: hi saturates there, the value degrades to 11-bit lo precision and
 *                               is lost beyond 66,528).  The networks trained in tests/test_gpu_trained.py reach 165
 *                               (6 x below the limit); a diverged network trips it.  Re-run with NSR_FP32.
 *   NSR_FLAG_OUTPUT_NONFINITE   a network output (r, g, b, sigma) was inf / NaN
 * nsr_weights_status copies the word to *flags_out (HOST), clears it on the device if `clear`, and WAITS for `stream`
 * (the second synchronising entry point).  The blob is therefore read-mostly, not read-only: `packed_dev` arguments
 * are const for the weight stream, the status word behind it is written by the kernels. */
#define NSR_FLAG_WEIGHT_RANGE 1u
#define NSR_FLAG_INPUT_RANGE 2u
#define NSR_FLAG_ACTIVATION_RANGE 4u
#define NSR_FLAG_OUTPUT_NONFINITE 8u
int nsr_weights_status(const void* packed_dev, int precision, int clear, unsigned* flags_out, void* stream);

/* Colour-head options of a packed network (stream-ordered write into the blob's tail, read by every later launch of every
 * entry point that evaluates this blob; cleared by re-packing).  `options` = the whole word, an OR of
 *   NSR_OPT_GAMMA       --gamma_correct (models/nerf_downX_model.py:271-276): render_rays returns pow(rgb, 1 / 2.2) per sample
 *   NSR_OPT_COLOR_NONE  --color_activation none (models/networks.py:173-180): the rgb head ends in nn.Identity, not nn.Sigmoid
 * NSR_ERR_INVALID_ARG for any other bit.  nsr_weights_set_gamma(enable) = nsr_weights_set_options(enable ? NSR_OPT_GAMMA : 0).
 * Inference only: the training step (nsr_train.h) evaluates its own copy of the weights with the default head. */
#define NSR_OPT_GAMMA 1u
#define NSR_OPT_COLOR_NONE 2u
int nsr_weights_set_options(void* packed_dev, int precision, unsigned options, void* stream);
int nsr_weights_set_gamma(void* packed_dev, int precision, int enable, void* stream);

/* The `render_flags` argument of the compositing entry points below (nsr_composite, nsr_render_rays_composited,
 * nsr_forward_rays*; named `white_bkgd` until NSR_VERSION 120) is a word of renderer options, like the reference's
 * VolumetricRenderer(opt) + white_bkgd pair (models/rendering.py:66-111):
 *   NSR_WHITE_BKGD      comp_rgb += 1 - opacity
 *   NSR_SIGMA_SOFTPLUS  --sigma_activation softplus (models/rendering.py:69-73): log(1 + exp(sigma - 1)) instead of relu(sigma)
 * It is NOT a C boolean: 0 and 1 keep the meaning they had when the argument was one (black / white background), but
 * any other "true" value (2, -1, ...) is an option word -- 2 selects softplus density on a black background, a bit outside
 * the defined ones returns NSR_ERR_INVALID_ARG (the break is listed in INTEGRATION.md, "Migration notes").  Pass
 * `flag ? NSR_WHITE_BKGD : 0`.  The training entry points take the same word plus their own bits (nsr_train.h). */
#define NSR_WHITE_BKGD 1
#define NSR_SIGMA_SOFTPLUS 2

/* ---- R1-R4: sub-pixel ray generation ---------------------------------------
 * Replaces get_ray_directions + get_rays (+ get_ndc_rays) + the einops regroup
 * '(h s1) (w s2) c -> (h w) (s1 s2) c' (models/utils.py:98-196;
 * data/llff_downX_dataset.py:473-490; data/blender_downX_dataset.py:207-215).
 * c2w: HOST pointer to 12 floats (3x4 row-major).  H, W, focal: HR image.
 * ndc != 0: LLFF path (near plane 1.0, near/far columns := 0/1, `near`,`far`
 * ignored); ndc == 0: near/far columns := near, far.
 * rays_dev: (H/s * W/s * s*s, 8) = [o(3), d(3), near, far], LR-pixel-major,
 * sub-pixel index dy*s+dx.  H % s == 0 and W % s == 0 required. */
int nsr_gen_rays(const float* c2w, int H, int W, double focal, int s, int ndc, float near_, float far_,
                 float* rays_dev, void* stream);
/* The same for the LR pixels [lr_lo, lr_hi) only (row-major LR index): rays_dev is ((lr_hi - lr_lo) * s*s, 8) and
 * holds exactly the rows [lr_lo * s*s, lr_hi * s*s) of what nsr_gen_rays writes.  This is the ray shard one GPU
 * renders when a frame is cut into contiguous LR-pixel blocks (replaces the scatter of
 * nn.DataParallel, models/networks.py:54-69): nothing is scattered, every rank generates its own block.
 * lr_lo == lr_hi is a valid empty shard. */
int nsr_gen_rays_range(const float* c2w, int H, int W, double focal, int s, int ndc, float near_, float far_,
                       int64_t lr_lo, int64_t lr_hi, float* rays_dev, void* stream);

/* ---- E1: positional encoding ------------------------------------------------
 * Replaces PositionalEncoding.__call__ (models/embedding.py:44-62), 3 input
 * channels: x (n,3) -> out (n, 3 + 6*deg) = [x, sin(2^0 x), cos(2^0 x), ...]. */
int nsr_posenc(const float* x, int64_t n, int deg, float* out, void* stream);

/* ---- S1: stratified sampling -------------------------------------------------
 * Replaces sample_along_rays (models/utils.py:17-44).  rays (R,8).
 * u == NULL: deterministic (eval) depths; u (R,n_samples): the uniform numbers
 * the reference draws with torch.rand_like (randomized=True branch).
 * z (R,n_samples); pts (R,n_samples,3) or NULL to skip cast_rays (utils.py:5-14). */
int nsr_sample_along_rays(const float* rays, int ray_stride, int64_t R, int n_samples, int lindisp, const float* u,
                          float* z, float* pts, void* stream);

/* ---- M1: the NeRF MLP --------------------------------------------------------
 * Replaces VanillaMLP.forward(x, sigma_only) (models/networks.py:182-226):
 * x (P,90) embedded rows -> out (P,4) = [rgb, sigma_raw], or (P,1) if sigma_only. */
int nsr_mlp_forward(const void* packed_dev, int precision, const float* x, int64_t P, int sigma_only,
                    float* out, void* stream);
/* Replaces render_rays(model, xyz, dir_embedded) (models/nerf_downX_model.py:260-278)
 * fused with cast_rays and both positional encodings: rays (R,8), z (R,N) ->
 * out (R*N, 4) = [rgb, sigma_raw] per sample point; the (P,90) matrix is never
 * materialised. */
int nsr_render_rays(const void* packed_dev, int precision, const float* rays, int ray_stride, const float* z,
                    int64_t R, int n_samples, float* out, void* stream);

/* render_rays followed by VolumetricRenderer.forward (models/nerf_downX_model.py:289-291 and :303-305) in ONE launch: a
 * 128-point tile of the MLP kernel holds whole rays when n_samples is 64 or 128, so the kernel composites them itself
 * and the (R, N, 4) network output need not go to HBM.  raw (R * N, 4) or NULL; comp_rgb (R, 3), depth (R), opacity (R),
 * weights (R, N): any may be NULL.  Results are bit-identical to nsr_render_rays + nsr_composite (the same device code).
 * NSR_ERR_UNSUPPORTED for other sample counts and for the NSR_F16 / NSR_BF16 fast paths (callers fall back to the pair). */
int nsr_render_rays_composited(const void* packed_dev, int precision, const float* rays, int ray_stride, const float* z,
                               int64_t R, int n_samples, int render_flags, float* raw, float* comp_rgb, float* depth,
                               float* opacity, float* weights, void* stream);

/* ---- V1: volumetric compositing ------------------------------------------------
 * Replaces VolumetricRenderer.forward (models/rendering.py:75-111).
 * rgb: element (r,k,c) at rgb[(r*N+k)*rgb_stride + c]; sigma: (r,k) at
 * sigma[(r*N+k)*sigma_stride]  (strides 3/1 = the reference's separate tensors,
 * 4/4 with sigma = rgb+3 = the interleaved (P,4) MLP output).
 * Outputs: comp_rgb (R,3), depth (R), opacity (R), weights (R,N); any may be NULL. */
int nsr_composite(const float* rgb, int rgb_stride, const float* sigma, int sigma_stride, const float* z,
                  int64_t R, int n_samples, int render_flags, float* comp_rgb, float* depth, float* opacity,
                  float* weights, void* stream);

/* ---- S2: hierarchical inverse-CDF resampling -----------------------------------
 * Replaces resample_along_rays (models/utils.py:47-95).  z (R,Nc), weights (R,Nc),
 * u == NULL: linspace(0,1,Ni) (eval) else u (R,Ni) (randomized).  z_out (R,Nc+Ni)
 * sorted; pts (R,Nc+Ni,3) or NULL (needs rays when non-NULL). */
int nsr_resample_along_rays(const float* rays, int ray_stride, const float* z, const float* weights, int64_t R,
                            int n_coarse, int n_importance, const float* u, float* z_out, float* pts, void* stream);

/* ---- D3: forward_rays, fused driver ----------------------------------------------
 * Replaces NeRFDownXModel.forward_rays / forward in eval mode
 * (models/nerf_downX_model.py:280-324) for the whole ray batch in one enqueue
 * sequence (no ray_chunk / point_chunk slicing, no host syncs).
 * outs[8] (HOST array of DEVICE pointers, any entry may be NULL), in this order:
 *   0 coarse_comp_rgbs (R,3)  1 coarse_depth (R)  2 coarse_opacity (R)  3 coarse_weights (R,Nc)
 *   4 fine_comp_rgbs  (R,3)   5 fine_depth (R)    6 fine_opacity (R)    7 fine_weights (R,Nc+Ni)
 * n_importance == 0 skips the fine pass (outs[4..7] untouched, packed_fine may be NULL).
 * workspace: device scratch of nsr_forward_rays_workspace_bytes_for(precision, ...) bytes: the depths and the coarse
 * weights, 12 * Nc + 4 * Nf bytes per ray, on the fused route (NSR_FP32 / NSR_F16X3 with 64 or 128 samples per pass: the
 * launch composites its own rays and the (R, N, 4) network output never exists); 16 * (Nc + Nf) bytes per ray more when
 * a pass has to go through nsr_render_rays + nsr_composite.  nsr_forward_rays_workspace_bytes is the precision-agnostic
 * upper bound (always sufficient). */
size_t nsr_forward_rays_workspace_bytes_for(int precision, int64_t R, int n_coarse, int n_importance);
size_t nsr_forward_rays_workspace_bytes(int64_t R, int n_coarse, int n_importance);
int nsr_forward_rays(const void* packed_coarse, const void* packed_fine, int precision, const float* rays,
                     int ray_stride, int64_t R, int n_coarse, int n_importance, int render_flags, int lindisp,
                     float* const* outs, void* workspace, size_t workspace_bytes, void* stream);

/* Same as nsr_forward_rays, plus HIP-event instrumentation for benchmarks: `events` is a HOST
 * array of 4 hipEvent_t (or NULL) recorded on `stream` immediately before / after the coarse
 * MLP launch ([0],[1]) and the fine MLP launch ([2],[3]) — the kernels that carry >99 % of the
 * path's FLOPs — so that a harness can read per-launch durations without a profiler. */
int nsr_forward_rays_profiled(const void* packed_coarse, const void* packed_fine, int precision, const float* rays,
                              int ray_stride, int64_t R, int n_coarse, int n_importance, int render_flags, int lindisp,
                              float* const* outs, void* workspace, size_t workspace_bytes, void* stream,
                              void* const* events);
/* Thin wrappers over hipEventCreate / hipEventDestroy / hipEventSynchronize+hipEventElapsedTime so a
 * ctypes harness uses the same HIP runtime instance libnsr is linked to.  elapsed: milliseconds. */
int nsr_event_create(void** event_out);
int nsr_event_destroy(void* event);
int nsr_event_elapsed_ms(void* start, void* stop, float* ms_out);

/* ---- A1 / A2: supersampling epilogue -----------------------------------------------
 * nsr_sr_mean replaces reshape(N_lr, s^2, c).mean(1) (models/nerf_downX_model.py:337-348).
 * nsr_unflatten replaces unflatten_reshape '(h1 w1) (s1 s2) c -> (h1 s1) (w1 s2) c'
 * (models/nerf_downX_model.py:410-416): x (H/s*W/s*s*s, c) -> out (H, W, c). */
int nsr_sr_mean(const float* hr, int64_t n_lr, int s2, int c, float* lr, void* stream);
int nsr_unflatten(const float* x, int H, int W, int s, int c, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_H_ */
