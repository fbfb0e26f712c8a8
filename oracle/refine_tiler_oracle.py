"""CPU oracle of the refinement stage's patch tiler and stitcher (config #5 tail) -- test infrastructure only.

Restates ``LLFFRefineDataset.__getitem__`` for the 'test' / 'test_train' splits (data/llff_refine_dataset.py:258-340)
and the stitching loop of ``RefineModel.test`` (models/refine_model.py:205-216):

  * SR patches on a ``patch_len`` grid, x outer / y inner, start clamped to ``size - patch_len`` (:306-312);
  * per patch, scan its pixels x outer / y inner and take, for the first ``num_ref_patches`` whose warped location
    ``locs[y, x]`` (the output of warp.py / nsr_depth_warp) lies inside the image, the reference patch whose top-left
    corner is that location clamped to ``size - patch_len`` (:315-327); pad with the SR patch itself (:328-329);
  * stitch predictions in patch order, later patches overwrite earlier ones (refine_model.py:211-214).

Pinned by ``tests/golden/refine_tiler.npz`` (the reference's own dataset class on a fabricated scene).
"""
from __future__ import annotations

import numpy as np


def tile(locs: np.ndarray, W: int, H: int, patch: int, n_ref: int):
    """-> starts (n, 2) int32 [x, y]; ref_starts (n, n_ref, 2) int32, (-1, -1) = "use the SR patch"."""
    starts, refs = [], []
    for i in range(0, W, patch):
        for j in range(0, H, patch):
            x0, y0 = min(W - patch, i), min(H - patch, j)
            found = []
            for m in range(x0, x0 + patch):
                for n in range(y0, y0 + patch):
                    lx, ly = locs[n, m, 0], locs[n, m, 1]
                    if 0 <= lx < W and 0 <= ly < H:
                        found.append((min(W - patch, int(lx)), min(H - patch, int(ly))))
                        if len(found) >= n_ref:
                            break
                if len(found) >= n_ref:
                    break
            starts.append((x0, y0))
            refs.append(found + [(-1, -1)] * (n_ref - len(found)))
    return np.array(starts, np.int32), np.array(refs, np.int32).reshape(len(starts), n_ref, 2)


def gather(sr_img: np.ndarray, ref_img: np.ndarray, starts, ref_starts, patch: int):
    """-> sr_patch (n, 3, p, p), ref_patches (n, n_ref, 3, p, p)."""
    n, n_ref = ref_starts.shape[:2]
    sr = np.stack([sr_img[:, y:y + patch, x:x + patch] for x, y in starts], 0)
    ref = np.empty((n, n_ref) + sr.shape[1:], sr.dtype)
    for k in range(n):
        for r in range(n_ref):
            rx, ry = ref_starts[k, r]
            ref[k, r] = sr[k] if rx < 0 else ref_img[:, ry:ry + patch, rx:rx + patch]
    return sr, ref


def stitch(patches: np.ndarray, starts, patch: int, W: int, H: int) -> np.ndarray:
    img = np.zeros((3, H, W), patches.dtype)
    for k, (x, y) in enumerate(starts):
        img[:, y:y + patch, x:x + patch] = patches[k]
    return img
