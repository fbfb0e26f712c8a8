/* nsr_warp.h — C ABI of the depth warp that feeds the refinement stage.
 *
 * Replaces the per-pixel double loop of warp.py:100-176 (one image per call; the reference loops over the
 * scene's images on the host): every pixel (x, y) of a rendered view is lifted with its NeRF depth, moved into the
 * reference view and projected to an integer pixel there.
 *   D     = ndc ? 1 / (1 - d + 1e-6) : d                       (warp.py:118, float32)
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

int nsr_depth_warp(const float* depth, int H, int W, double focal, const float* c2w, const double* ref_w2c, int ndc,
                   const float* ref_rgb, double* locs, float* warped, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_WARP_H_ */
