/* nsr_warp.h — C ABI of the depth warp that feeds the refinement stage.
 *
 * Replaces the per-pixel double loop of warp.py:100-176 (one image per call; the reference loops over the
 * scene's images on the host): every pixel (x, y) of a rendered view is lifted with its NeRF depth, moved into the
 * reference view and projected to an integer pixel there.
 *   D     = depth_kind == NSR_DEPTH_NDC    ? 1 / (1 - d + 1e-6)      (warp.py:118, float32: LLFF / NDC scenes)
 *         : depth_kind == NSR_DEPTH_METRIC ? d                        (warp.py:120-126: the `spheric_poses` branch
 *                                                                      leaves the rendered depth as it is)
 *         : depth_kind == NSR_DEPTH_RAY    ? d / |((x + .5 - W/2) / f, (y + .5 - H/2) / f, 1)|   (float32)
 *     NSR_DEPTH_RAY has no counterpart in the reference (its warp script is LLFF-only): it is what BASELINE config
 *     #5 needs.  Blender rays keep get_rays' UNIT-norm directions (models/utils.py:150,
 *     data/blender_downX_dataset.py:207-215: no NDC warp, near / far 2 / 6), so the rendered depth sum(w z) is a
 *     distance ALONG THE RAY, while the lift below wants the depth along the camera axis; the two differ by the
 *     norm of the pixel's camera-space direction (up to 9 % in the corners of an 800 x 800 lego frame).
 *   p_cam = ((x + .5 - W/2) / f * D, -(y + .5 - H/2) / f * D, -D)          (:127-131, float64)
 *   p_w   = c2w[:, :3] p_cam + c2w[:, 3];  q = ref_w2c[:, :3] p_w + ref_w2c[:, 3];  q /= -q[2]   (:154-158)
 *   u, v  = trunc(q[0] f + W/2), trunc(q[1] (-f) + H/2)        (:160-161)
 *   locs[y][x] = (u, v, -1);   warped[:, y, x] = inside(u, v) ? ref_rgb[:, v, u] : 0             (:165-168)
 * depth (H, W) fp32 DEVICE; c2w: 12 floats HOST (row-major 3x4, the float32 pose of the view); ref_w2c: 12 doubles
 * HOST (float64 world-to-camera of the reference view); ref_rgb (3, H, W) fp32 DEVICE or NULL; locs (H, W, 3)
 * float64 DEVICE (the content of `{i}_locs.npz`); warped (3, H, W) fp32 DEVICE or NULL.
 * Integer pixel targets are bit-exact with the reference (float64 arithmetic in the reference's operation order).
 */
#ifndef NSR_WARP_H_
#define NSR_WARP_H_

#include "nsr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum nsr_depth_kind {
  NSR_DEPTH_METRIC = 0, /* depth along the camera axis, used as it is (warp.py:120-126) */
  NSR_DEPTH_NDC = 1,    /* NDC depth of a forward-facing scene -> 1 / (1 - d + 1e-6) (warp.py:118) */
  NSR_DEPTH_RAY = 2     /* distance along a unit-norm ray (Blender rays) -> depth along the camera axis */
} nsr_depth_kind;

int nsr_depth_warp(const float* depth, int H, int W, double focal, const float* c2w, const double* ref_w2c,
                   int depth_kind, const float* ref_rgb, double* locs, float* warped, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_WARP_H_ */
