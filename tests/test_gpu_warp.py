"""N2 depth warp of the HIP path (include/nsr_warp.h) against the fixture produced by executing the reference's own
warp.py class, and against the CPU oracle at full frame size.  Integer pixel targets: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import warp_oracle as wo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def warp():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from nerf_sr_amd import warp as _w
    return _w


def test_depth_warp_vs_reference_fixture(warp, golden_dir):
    g = np.load(os.path.join(golden_dir, "warp_llff.npz"))
    ref = torch.from_numpy(g["ref_rgbs"]).cuda()
    for i in range(int(g["n_img"])):
        locs, warped = warp.depth_warp(torch.from_numpy(g[f"depth_{i}"]).cuda(), g[f"c2w_{i}"], g["ref_w2c"],
                                       float(g["focal"]), True, ref)
        assert np.array_equal(locs.cpu().numpy(), g[f"locs_{i}"]), i
        assert np.array_equal(warped.cpu().numpy(), g[f"warped_{i}"]), i
    only = warp.depth_warp(torch.from_numpy(g["depth_1"]).cuda(), g["c2w_1"], g["ref_w2c"], float(g["focal"]))
    assert np.array_equal(only.cpu().numpy(), g["locs_1"])


def test_metric_depth_branch_vs_reference_fixture(warp, golden_dir):
    """warp.py:120-126 (spheric_poses: depth used as it is; the branch config #5 takes), fixture from the reference's
    own class: bit-exact."""
    g = np.load(os.path.join(golden_dir, "warp_spheric.npz"))
    ref = torch.from_numpy(g["ref_rgbs"]).cuda()
    for i in range(int(g["n_img"])):
        locs, warped = warp.depth_warp(torch.from_numpy(g[f"depth_{i}"]).cuda(), g[f"c2w_{i}"], g["ref_w2c"],
                                       float(g["focal"]), "metric", ref)
        assert np.array_equal(locs.cpu().numpy(), g[f"locs_{i}"]), i
        assert np.array_equal(warped.cpu().numpy(), g[f"warped_{i}"]), i


def test_ray_distance_variant_vs_oracle(warp):
    """NSR_DEPTH_RAY at BASELINE config #5's frame size (800 x 800, Blender focal, poses on the radius-4 sphere,
    ray distances in [2, 6]): bit-exact with the oracle's restatement; a view warped into itself is the identity."""
    from nerf_sr_amd import cameras
    H = W = 800
    rng = np.random.default_rng(5)
    t = (2.0 + 4.0 * rng.random((H, W))).astype(np.float32)
    ref = rng.random((3, H, W)).astype(np.float32)
    c2w = cameras.spheric_pose(40.0, -30.0, 4.0).astype(np.float32)
    ref_c2w = np.concatenate([cameras.spheric_pose(25.0, -30.0, 4.0).astype(np.float32), np.array([[0, 0, 0, 1]])], 0)
    f = cameras.blender_focal(W)
    want_l, want_w = wo.depth_warp(t, c2w, np.linalg.inv(ref_c2w)[:3], f, "ray", ref)
    got_l, got_w = warp.depth_warp(torch.from_numpy(t).cuda(), c2w, np.linalg.inv(ref_c2w)[:3], f, "ray",
                                   torch.from_numpy(ref).cuda())
    assert np.array_equal(got_l.cpu().numpy(), want_l) and np.array_equal(got_w.cpu().numpy(), want_w)
    own = np.linalg.inv(np.concatenate([c2w.astype(np.float64), [[0, 0, 0, 1]]], 0))[:3]
    ident = warp.depth_warp(torch.from_numpy(t).cuda(), c2w, own, f, "ray").cpu().numpy()
    assert (ident[..., 0] == np.arange(W)[None, :]).mean() > 0.97 and (ident[..., 1] == np.arange(H)[:, None]).mean() > 0.97


def test_depth_warp_full_frame_vs_oracle(warp):
    """504 x 378 (configs #2/#5 geometry), metric and NDC depth, random reference image: bit-exact with the oracle."""
    from nerf_sr_amd import cameras
    H, W = 378, 504
    rng = np.random.default_rng(3)
    depth = (0.2 + 0.7 * rng.random((H, W))).astype(np.float32)
    ref = rng.random((3, H, W)).astype(np.float32)
    c2w = cameras.spiral_pose(0.9).astype(np.float32)
    ref_c2w = np.concatenate([cameras.spiral_pose(0.1).astype(np.float32), np.array([[0, 0, 0, 1]])], 0)
    ref_w2c = np.linalg.inv(ref_c2w)[:3]
    f = cameras.llff_focal(W)
    for ndc in (True, False):
        want_l, want_w = wo.depth_warp(depth if ndc else depth * 5 + 1, c2w, ref_w2c, f, ndc, ref)
        got_l, got_w = warp.depth_warp(torch.from_numpy(depth if ndc else depth * 5 + 1).cuda(), c2w, ref_w2c, f, ndc,
                                       torch.from_numpy(ref).cuda())
        assert np.array_equal(got_l.cpu().numpy(), want_l)
        assert np.array_equal(got_w.cpu().numpy(), want_w)
    assert warp.depth_warp(torch.zeros(0, 7).cuda(), c2w, ref_w2c, f).shape == (0, 7, 3)
