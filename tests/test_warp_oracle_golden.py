"""Pins oracle/warp_oracle.py to the fixture produced by executing the reference's own warp.py class
(tests/golden/make_golden_warp.py).  Integer pixel targets: bit-exact.  CPU only."""
import os

import numpy as np

from oracle import warp_oracle as wo


def test_depth_warp_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "warp_llff.npz"))
    W, H = int(g["W"]), int(g["H"])
    for i in range(int(g["n_img"])):
        locs, warped = wo.depth_warp(g[f"depth_{i}"], g[f"c2w_{i}"], g["ref_w2c"], float(g["focal"]), True, g["ref_rgbs"])
        assert locs.dtype == np.float64 and locs.shape == (H, W, 3)
        assert np.array_equal(locs, g[f"locs_{i}"]), f"image {i}: {(locs != g[f'locs_{i}']).sum()} entries differ"
        assert np.array_equal(warped, g[f"warped_{i}"])
    # image 0 is the reference view: the warp is the identity wherever rounding keeps the pixel
    u0 = g["locs_0"][..., 0]
    assert (u0 == np.arange(W)[None, :]).mean() > 0.95


def test_metric_depth_branch_matches_reference(golden_dir):
    """The reference's other depth branch (warp.py:120-126, spheric_poses: the depth map is used as it is) -- the one
    BASELINE config #5 (Blender, no NDC) goes through; fixture from the reference's own class with that switch set."""
    g = np.load(os.path.join(golden_dir, "warp_spheric.npz"))
    for i in range(int(g["n_img"])):
        locs, warped = wo.depth_warp(g[f"depth_{i}"], g[f"c2w_{i}"], g["ref_w2c"], float(g["focal"]), False, g["ref_rgbs"])
        assert np.array_equal(locs, g[f"locs_{i}"]), f"image {i}: {(locs != g[f'locs_{i}']).sum()} entries differ"
        assert np.array_equal(warped, g[f"warped_{i}"])
    # the NDC conversion must NOT have been applied: with it the targets differ
    other = wo.depth_warp(g["depth_1"], g["c2w_1"], g["ref_w2c"], float(g["focal"]), True)
    assert not np.array_equal(other, g["locs_1"])


def test_ray_distance_variant_is_consistent():
    """NSR_DEPTH_RAY (build-defined, no reference counterpart): a depth map of ray DISTANCES to a fronto-parallel
    plane z = -Z0 converts to the constant camera-axis depth Z0, and a view warped into itself is the identity."""
    H, W, f, Z0 = 40, 56, 61.5, 3.25
    gx, gy = np.meshgrid(np.arange(W) + 0.5 - W / 2, np.arange(H) + 0.5 - H / 2, indexing="xy")
    t = (Z0 * np.sqrt((gx / f) ** 2 + (gy / f) ** 2 + 1.0)).astype(np.float32)
    D = wo.axis_depth_from_ray_distance(t, f)
    assert D.dtype == np.float32 and np.abs(D - Z0).max() < 1e-6 * Z0 * 4
    c2w = np.array([[0.8, 0, 0.6, 0.3], [0, 1, 0, -0.2], [-0.6, 0, 0.8, 4.0]], np.float32)
    w2c = np.linalg.inv(np.concatenate([c2w.astype(np.float64), [[0, 0, 0, 1]]], 0))[:3]
    locs = wo.depth_warp(t, c2w, w2c, f, "ray")
    assert (locs[..., 0] == np.arange(W)[None, :]).mean() > 0.95 and (locs[..., 1] == np.arange(H)[:, None]).mean() > 0.95
