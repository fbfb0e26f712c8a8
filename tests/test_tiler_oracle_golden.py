"""Pins oracle/refine_tiler_oracle.py to what the reference's own LLFFRefineDataset produced
(tests/golden/make_golden_tiler.py).  Pure data movement: everything is exact.  CPU only."""
import os

import numpy as np

from oracle import refine_tiler_oracle as rt


def test_tiler_and_stitcher_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "refine_tiler.npz"))
    W, H, P, NR = int(g["W"]), int(g["H"]), int(g["patch_len"]), int(g["num_ref"])
    for img in range(2):
        starts, refs = rt.tile(g[f"locs_{img}"], W, H, P, NR)
        assert np.array_equal(starts, g[f"start_locs_{img}"].astype(np.int32))
        sr, ref = rt.gather(g["sr_imgs"][img], g["ref_img"], starts, refs, P)
        assert np.array_equal(sr, g[f"sr_patch_{img}"])
        assert np.array_equal(ref, g[f"ref_patches_{img}"])
        assert np.array_equal(rt.stitch(g[f"pred_{img}"], starts, P, W, H), g[f"stitched_{img}"])
    # image 0: the left patch column has no valid warp target at all -> its references are the SR patch itself
    _, refs0 = rt.tile(g["locs_0"], W, H, P, NR)
    assert (refs0[:2] == -1).all() and (refs0[2:] >= 0).any()
