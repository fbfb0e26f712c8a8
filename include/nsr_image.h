/* nsr_image.h — C ABI of the LR-target construction of the downX datasets.
 *
 * Replaces, per training image, `img.resize(img_wh, Image.LANCZOS)`, `img.resize((W/s, H/s), Image.LANCZOS)`,
 * `ToTensor` and the regroup '(h s1) (w s2) c -> (h w) (s1 s2) c' of data/llff_downX_dataset.py:312-329 (RGB) and
 * data/blender_downX_dataset.py:104-135 (RGBA: Pillow resamples such images premultiplied -- PIL/Image.py `resize`:
 * convert("RGBa") -> resample -> convert("RGBA"), src/libImaging/Convert.c rgbA2rgba / rgba2rgbA -- and the dataset
 * then blends onto white, `rgb * a + (1 - a)`; nsr_rgba_premultiply_u8 + nsr_image_to_targets_rgba).  The resampling arithmetic is Pillow's 8-bit path (src/libImaging/Resample.c:
 * precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; the reference does not
 * pin a Pillow version, the algorithm has been stable since 3.x): LANCZOS (a = 3) weights normalised in double,
 * rounded to 22-bit fixed point, int32 accumulation, clip to [0, 255]; horizontal pass first.  Results are
 * bit-identical to Pillow's.
 *
 * Same conventions as nsr.h (device pointers, caller-owned memory, enqueue on the caller's stream, int status).
 * The coefficient tables are computed on the HOST in double precision exactly as Pillow does (nsr_lanczos_coeffs,
 * plain C, no device work) and handed to the passes as device arrays by the caller.
 */
#ifndef NSR_IMAGE_H_
#define NSR_IMAGE_H_

#include "nsr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* taps per output sample when `in_size` samples are resampled to `out_size`: 2 * ceil(3 * max(in/out, 1)) + 1 */
int nsr_lanczos_ksize(int in_size, int out_size);
/* HOST: bounds (out_size, 2) = (first tap, tap count), kk (out_size, nsr_lanczos_ksize()) fixed-point weights */
int nsr_lanczos_coeffs(int in_size, int out_size, int32_t* bounds_host, int32_t* kk_host);
/* One resampling pass of an interleaved 8-bit image src (H, W, C): axis 1 resamples every row W -> out_size
 * (dst (H, out_size, C)), axis 0 every column H -> out_size (dst (out_size, W, C)).  bounds / kk: DEVICE copies of
 * the tables for (in = W or H, out = out_size). */
int nsr_resample_pass_u8(const uint8_t* src, int H, int W, int C, int axis, int out_size, const int32_t* bounds_dev,
                         const int32_t* kk_dev, int ksize, uint8_t* dst, void* stream);
/* ToTensor + regroup: img (H, W, 3) uint8 -> out (H/s * W/s, s*s, 3) fp32 = img / 255, LR-pixel-major, sub-pixel
 * index dy*s+dx (the layout of the ray tensor, nsr_gen_rays).  s = 1: the plain (H*W, 3) target tensor. */
int nsr_image_to_targets(const uint8_t* img, int H, int W, int s, float* out, void* stream);

/* RGBA images.  nsr_rgba_premultiply_u8: Pillow's RGBA -> "RGBa" (inverse == 0: colour bytes c -> MULDIV255(c, a)) and
 * "RGBa" -> RGBA (inverse != 0: c -> clip8(255 c / a), copied when a is 0 or 255) over n_px interleaved 4-byte pixels
 * (4-byte aligned; src == dst allowed).  An RGBA resize = premultiply, the two nsr_resample_pass_u8 passes with C = 4,
 * un-premultiply.  nsr_image_to_targets_rgba: img (H, W, 4) uint8 -> out (H/s * W/s, s*s, 3) fp32 =
 * rgb / 255 * (a / 255) + (1 - a / 255) (data/blender_downX_dataset.py:117-120), same order as nsr_image_to_targets. */
int nsr_rgba_premultiply_u8(const uint8_t* src, int64_t n_px, int inverse, uint8_t* dst, void* stream);
int nsr_image_to_targets_rgba(const uint8_t* img, int H, int W, int s, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_IMAGE_H_ */
