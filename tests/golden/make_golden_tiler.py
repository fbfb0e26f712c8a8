#!/usr/bin/env python3
"""Golden vectors of the reference's refinement patch tiler (config #5 tail): ``LLFFRefineDataset.__getitem__`` in its
'test' split (data/llff_refine_dataset.py:303-340: 64-pixel grid of SR patches, for each the first
``num_ref_patches`` reference patches whose top-left corners are the warped locations ``{i}_locs.npz`` of the patch's
pixels scanned column by column, missing ones filled with the SR patch itself) and the stitching loop of
``RefineModel.test`` (models/refine_model.py:211-216).  Development container only; data only.

The reference's own dataset class is instantiated on a fabricated scene (COLMAP binaries written by
tests/colmap_writer.py, random PNGs, random integer ``locs``), small sizes: 40 x 24 image, 16-pixel patches, 3 refs.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
import make_golden_warp as mw  # noqa: E402
from colmap_writer import write_reconstruction  # noqa: E402


class Normalize:
    def __init__(self, mean, std):
        self.m, self.s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)

    def __call__(self, t):
        return (t - self.m) / self.s


class Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


def main():
    mg.install_shim()
    import torchvision.transforms as T
    T.ToTensor, T.ToPILImage, T.Normalize, T.Compose = mw.ToTensor, mw.ToPILImage, Normalize, Compose
    rng = np.random.default_rng(21)
    W, H, P, NR, SPLIT = 40, 24, 16, 3, 2
    root = tempfile.mkdtemp(prefix="nsr_tiler_root_")
    syn = tempfile.mkdtemp(prefix="nsr_tiler_syn_")
    os.makedirs(os.path.join(root, "images"))
    names = ["a.png", "b.png", "c.png"]
    for n in names:
        Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "images", n))
    c2ws = [mw.look_at_colmap(np.array([0.3 * i, 0.0, 0.0]), np.array([0.0, 0.0, 4.0])) for i in range(3)]
    pts = rng.normal(0, 0.5, (20, 3)) + np.array([0.0, 0.0, 4.0])
    write_reconstruction(os.path.join(root, "sparse", "0"), W, H, 2.0 * W, names, c2ws, pts, [[1, 2, 3]] * len(pts))
    locs_all = []
    for i in range(2):                                   # the 'test' split hard-codes two synthesised images
        Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(os.path.join(syn, f"{i}-fine-ori.png"))
        # integer-valued warp targets as nsr_depth_warp / warp.py write them; many outside the image, so that some
        # patches find fewer than NR references (the fill-with-SR-patch branch) and scan order matters
        locs = np.stack([rng.integers(-30, W + 30, (H, W)), rng.integers(-20, H + 20, (H, W)), -np.ones((H, W))], -1).astype(np.float64)
        locs[:, :16] = -5.0 if i == 0 else locs[:, :16]  # image 0: the left patch column has no valid reference at all
        np.savez(os.path.join(syn, f"{i}_locs.npz"), locs)
        locs_all.append(locs)
    from data.llff_refine_dataset import LLFFRefineDataset
    opt = types.SimpleNamespace(dataset_root=root, img_wh=(W, H), ref_idx=0, syn_dataroot=syn, patch_len=P, num_ref_patches=NR,
                                test_img_split=SPLIT)
    ds = LLFFRefineDataset(opt, "test")
    out = {"W": W, "H": H, "patch_len": P, "num_ref": NR, "ref_img": mg.np32(ds.ref_img), "sr_imgs": mg.np32(ds.sr_imgs)}
    for img in range(2):
        out[f"locs_{img}"] = locs_all[img]
        chunks = [ds[img * SPLIT + c] for c in range(SPLIT)]
        sr = torch.cat([c["sr_patch"] for c in chunks], 0)
        ref = torch.cat([c["ref_patches"] for c in chunks], 0)
        st = torch.cat([c["start_locs"] for c in chunks], 0)
        out[f"sr_patch_{img}"], out[f"ref_patches_{img}"], out[f"start_locs_{img}"] = mg.np32(sr), mg.np32(ref), mg.np32(st)
        # RefineModel.test's stitching, with a stand-in "prediction" = 0.5 * sr_patch + patch index / 100
        pred = 0.5 * sr + torch.arange(sr.shape[0]).view(-1, 1, 1, 1) / 100.0
        canvas = torch.zeros(3, H, W)
        for p_idx, patch in enumerate(pred):
            x0, y0 = int(st[p_idx][0]), int(st[p_idx][1])
            canvas[:, y0:y0 + P, x0:x0 + P] = patch
        out[f"pred_{img}"], out[f"stitched_{img}"] = mg.np32(pred), mg.np32(canvas)
        print(img, "patches", tuple(sr.shape), "refs", tuple(ref.shape))
    path = os.path.join(HERE, "refine_tiler.npz")
    np.savez_compressed(path, **out)
    print("->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
