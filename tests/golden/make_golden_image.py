#!/usr/bin/env python3
"""Golden vectors of the LR-target construction (data/llff_downX_dataset.py:312-329).

Runs ONLY in the development container.  The arithmetic is Pillow's (``Image.resize(size, Image.LANCZOS)`` on 8-bit
RGB; a dependency of the reference that is not in its tree), so the fixture is made by running Pillow itself (version
recorded in the file) on small synthetic images; the float tensors follow the dataset's own lines: ``ToTensor`` (uint8
-> float32 / 255, CHW), ``view(3, -1).permute(1, 0)``, and the einops regroup of the HR image
``'(h s1) (w s2) c -> (h w) (s1 s2) c'``.

Fixture ``lanczos.npz``: source images, Pillow's resize results for down-, up- and mixed scaling, and for the dataset
case (scene image -> HR 64 x 48 -> LR 32 x 24 / 16 x 12) the tensors ``rgbs`` and ``rgbs_ori`` for s = 2 and s = 4.
"""
import os

import einops
import numpy as np
import PIL
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def to_tensor(img: Image.Image) -> torch.Tensor:
    """torchvision.transforms.ToTensor for an 8-bit RGB PIL image (torchvision is not installed here)."""
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def main():
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:96, 0:128]
    smooth = np.stack([127 + 120 * np.sin(xx / 9.0 + yy / 17.0), 127 + 120 * np.cos(xx / 5.0), yy * 255 / 96], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    out = {"pillow_version": PIL.__version__, "smooth": smooth, "noise": noise}
    for name, img in (("smooth", smooth), ("noise", noise)):
        for (w, h) in ((64, 48), (32, 24), (40, 30), (100, 70), (83, 20), (200, 61)):
            out[f"{name}_{w}x{h}"] = np.asarray(Image.fromarray(img).resize((w, h), Image.LANCZOS))
    # the dataset case (:312-329, ds_method = 'lanc')
    img_wh = (64, 48)
    for s in (2, 4):
        img = Image.fromarray(smooth).convert("RGB").resize(img_wh, Image.LANCZOS)
        imgX = img.resize((img_wh[0] // s, img_wh[1] // s), Image.LANCZOS)
        imgX = to_tensor(imgX)
        img_t = to_tensor(img)
        img_t = img_t.view(3, -1).permute(1, 0)
        imgX = imgX.view(3, -1).permute(1, 0)
        img_t = img_t.view(img_wh[1], img_wh[0], -1)
        img_t = einops.rearrange(img_t, "(h s1) (w s2) c -> (h w) (s1 s2) c", s1=s, s2=s)
        out[f"rgbs_s{s}"] = imgX.contiguous().numpy()
        out[f"rgbs_ori_s{s}"] = img_t.contiguous().numpy()
    path = os.path.join(HERE, "lanczos.npz")
    np.savez_compressed(path, **out)
    print("image fixture ->", path, f"{os.path.getsize(path) / 1024:.0f} KiB, Pillow {PIL.__version__}")


if __name__ == "__main__":
    main()
