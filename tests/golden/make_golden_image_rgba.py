#!/usr/bin/env python3
"""Golden vectors of the Blender datasets' LR-target construction on RGBA images (data/blender_downX_dataset.py:104-118).

Runs ONLY in the development container.  The arithmetic is Pillow's: ``Image.resize(size, Image.LANCZOS)`` on an RGBA
image converts to premultiplied "RGBa", resamples the four 8-bit channels, and converts back (PIL/Image.py ``resize``;
src/libImaging/Convert.c ``rgbA2rgba`` / ``rgba2rgbA``) -- so the fixture is made by running Pillow itself (version
recorded) on small synthetic RGBA images; the float tensors follow the dataset's own lines: ``ToTensor``,
``view(4, -1).permute(1, 0)``, ``rgb * a + (1 - a)`` (blend onto white), and the einops regroup of the HR image.

Fixture ``lanczos_rgba.npz``: source images, Pillow's RGBA resize results, the RGBa round trip on its own, and for the
dataset case (scene image -> HR 64 x 64 -> LR 32 x 32 / 16 x 16) the tensors ``rgbs`` and ``rgbs_ori`` for s = 2 and 4.
"""
import os

import einops
import numpy as np
import PIL
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def to_tensor(img: Image.Image) -> torch.Tensor:
    """torchvision.transforms.ToTensor for an 8-bit PIL image (torchvision is not installed here)."""
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def main():
    rng = np.random.default_rng(33)
    yy, xx = np.mgrid[0:96, 0:96]
    rr = np.hypot(xx - 48.0, yy - 40.0)
    alpha = np.clip((36.0 - rr) * 40.0, 0, 255)                 # an object with a soft edge on a transparent background
    alpha[60:80, 10:90] = 255
    obj = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 13.0), 127 + 120 * np.cos(xx / 5.0), yy * 255 / 96, alpha], -1).astype(np.uint8)
    noise = rng.integers(0, 256, (61, 83, 4), dtype=np.uint8)
    noise[:8, :, 3] = 0
    noise[8:16, :, 3] = 255
    out = {"pillow_version": PIL.__version__, "obj": obj, "noise": noise}
    for name, img in (("obj", obj), ("noise", noise)):
        im = Image.fromarray(img, "RGBA")
        out[f"{name}_RGBa"] = np.asarray(im.convert("RGBa"))
        out[f"{name}_RGBa_back"] = np.asarray(im.convert("RGBa").convert("RGBA"))
        for (w, h) in ((64, 64), (32, 32), (40, 30), (120, 70)):
            out[f"{name}_{w}x{h}"] = np.asarray(im.resize((w, h), Image.LANCZOS))
    img_wh = (64, 64)
    for s in (2, 4):
        img = Image.fromarray(obj, "RGBA").resize(img_wh, Image.LANCZOS)
        imgX = img.resize((img_wh[0] // s, img_wh[1] // s), Image.LANCZOS)
        imgX = to_tensor(imgX)
        img_t = to_tensor(img)
        img_t = img_t.view(4, -1).permute(1, 0)
        img_t = img_t[:, :3] * img_t[:, -1:] + (1 - img_t[:, -1:])
        imgX = imgX.view(4, -1).permute(1, 0)
        imgX = imgX[:, :3] * imgX[:, -1:] + (1 - imgX[:, -1:])
        img_t = img_t.view(img_wh[1], img_wh[0], -1)
        img_t = einops.rearrange(img_t, "(h s1) (w s2) c -> (h w) (s1 s2) c", s1=s, s2=s)
        out[f"rgbs_s{s}"] = imgX.contiguous().numpy()
        out[f"rgbs_ori_s{s}"] = img_t.contiguous().numpy()
    path = os.path.join(HERE, "lanczos_rgba.npz")
    np.savez_compressed(path, **out)
    print("RGBA image fixture ->", path, f"{os.path.getsize(path) / 1024:.0f} KiB, Pillow {PIL.__version__}")


if __name__ == "__main__":
    main()
