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
