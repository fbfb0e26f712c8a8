"""CPU oracle of the reference's depth warp -- test infrastructure only.

Restates the per-pixel stage of ``warp.py:100-176``: every pixel of image ``i`` is lifted with its rendered NeRF
depth, moved into the reference view (image 0) and projected to an INTEGER pixel there:

    D      = 1 / (1 - d_ndc + 1e-6)                      (:118, LLFF/NDC scenes only; float32)
    p_cam  = ((x + .5 - W/2) / f * D, -(y + .5 - H/2) / f * D, -D)       (:127-131; float64 under NumPy >= 2)
    p_w    = c2w[:, :3] @ p_cam + c2w[:, 3]              (:154; c2w float32 values, float64 arithmetic)
    q      = ref_w2c[:, :3] @ p_w + ref_w2c[:, 3];  q /= -q[2]           (:157-158)
    u, v   = int(q[0] * f + W/2), int(q[1] * (-f) + H/2)  (:160-161; truncation toward zero)
    locs[y, x] = (u, v, -1);  warped[:, y, x] = ref_rgb[:, v, u] if 0 <= u < W and 0 <= v < H else 0   (:165-168)

Vectorised over the image; each 3-vector product is spelled out as ((a0 x0 + a1 x1) + a2 x2) + t in float64.
Pinned by ``tests/golden/warp_llff.npz`` (produced by executing the reference's own class, ``make_golden_warp.py``).
"""
from __future__ import annotations

import numpy as np


def metric_depth_from_ndc(d_ndc: np.ndarray) -> np.ndarray:
    """``1 / (1 - d + 1e-6)`` in float32 (warp.py:118)."""
    d = np.asarray(d_ndc, np.float32)
    return (np.float32(1.0) / ((np.float32(1.0) - d) + np.float32(1e-6))).astype(np.float32)


def _affine(m: np.ndarray, x0, x1, x2):
    return [((m[r, 0] * x0 + m[r, 1] * x1) + m[r, 2] * x2) + m[r, 3] for r in range(3)]


def axis_depth_from_ray_distance(t: np.ndarray, focal: float) -> np.ndarray:
    """Distance along the unit-norm ray through each pixel centre -> depth along the camera axis, float32:
    ``t / |((x + .5 - W/2) / f, -(y + .5 - H/2) / f, -1)|``.  NOT a restatement: the reference's warp script is
    LLFF-only; this is the variant BASELINE config #5 (Blender rays: ``get_rays`` normalises, models/utils.py:150)
    needs, defined in include/nsr_warp.h (NSR_DEPTH_RAY) and restated here operation by operation."""
    t = np.asarray(t, np.float32)
    H, W = t.shape
    f32 = np.float32(focal)
    gx, gy = np.meshgrid(np.arange(W, dtype=np.float32) + np.float32(0.5) - np.float32(W / 2),
                         -(np.arange(H, dtype=np.float32) + np.float32(0.5) - np.float32(H / 2)), indexing="xy")
    cx, cy = gx / f32, gy / f32
    nrm = np.sqrt((cx * cx + cy * cy) + np.float32(1.0)).astype(np.float32)
    return (t / nrm).astype(np.float32)


def depth_warp(depth: np.ndarray, c2w: np.ndarray, ref_w2c: np.ndarray, focal: float, ndc=True,
               ref_rgb: np.ndarray = None):
    """depth (H, W) float32; ``ndc``: True / 'ndc' = NDC depth (:118), False / 'metric' = depth along the camera axis
    as it is (the ``spheric_poses`` branch, :120-126), 'ray' = distance along a unit-norm ray (see
    ``axis_depth_from_ray_distance``); c2w (3, 4) float32; ref_w2c (3, 4) float64.
    Returns locs (H, W, 3) float64 and, if ``ref_rgb`` (3, H, W) is given, the warped image (3, H, W) float32."""
    H, W = depth.shape
    f = np.float64(focal)
    kind = ndc if isinstance(ndc, str) else ("ndc" if ndc else "metric")
    if kind == "ndc":
        D = metric_depth_from_ndc(depth)
    elif kind == "ray":
        D = axis_depth_from_ray_distance(depth, focal)
    else:
        D = np.asarray(depth, np.float32)
    i_idx, j_idx = np.meshgrid(np.arange(W, dtype=np.float32) + np.float32(0.5),
                               np.arange(H, dtype=np.float32) + np.float32(0.5), indexing="xy")
    Dd = D.astype(np.float64)
    x0 = (i_idx - np.float32(W / 2)).astype(np.float64) / f * Dd
    x1 = (-(j_idx - np.float32(H / 2))).astype(np.float64) / f * Dd
    x2 = (-D).astype(np.float64)
    pw = _affine(np.asarray(c2w, np.float32).astype(np.float64), x0, x1, x2)
    q = _affine(np.asarray(ref_w2c, np.float64), *pw)
    den = -q[2]
    qx, qy = q[0] / den, q[1] / den
    u = np.trunc(qx * f + W / 2)
    v = np.trunc(qy * (-f) + H / 2)
    locs = np.stack([u, v, q[2] / den], -1)
    if ref_rgb is None:
        return locs
    inside = (u >= 0) & (u < W) & (v >= 0) & (v < H)
    ui, vi = np.where(inside, u, 0).astype(np.int64), np.where(inside, v, 0).astype(np.int64)
    warped = np.where(inside[None], np.asarray(ref_rgb, np.float32)[:, vi, ui], np.float32(0.0)).astype(np.float32)
    return locs, warped
