#!/usr/bin/env python3
"""Render one supersampled frame with the HIP path and write it as PNGs (needs an MI355X).

    python examples/render_frame.py [--checkpoint-dir DIR --name EXP --epoch 30] [--wh 504 378] [--downscale 2] [--out out]

Without a checkpoint the networks are the synthetic "smooth" field of nerf_sr_amd.weights (there is nothing to
download); with one, `{epoch}_net_Coarse.pth` / `{epoch}_net_Fine.pth` of a NeRF-SR experiment are loaded as the
reference's test.py does.  Output: `<out>_hr.png` (H x W, the s x s sub-pixel rays as pixels) and `<out>_lr.png` (the
s^2 means = the image the reference trains against).
"""
import argparse
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import cameras, io  # noqa: E402
from nerf_sr_amd.model import NeRFDownXModel, default_options  # noqa: E402
from nerf_sr_amd.weights import make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint-dir")
    ap.add_argument("--name", default="exp")
    ap.add_argument("--epoch", default="latest")
    ap.add_argument("--wh", type=int, nargs=2, default=(504, 378))
    ap.add_argument("--downscale", type=int, default=2)
    ap.add_argument("--pose-t", type=float, default=0.4, help="position on the LLFF spiral path")
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--out", default="frame")
    a = ap.parse_args()
    if a.checkpoint_dir:
        pc, pf = io.checkpoint_paths(a.checkpoint_dir, a.name, a.epoch)
        sd_c, sd_f = io.load_network_state(pc), io.load_network_state(pf)
    else:
        sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    opt = default_options(img_wh=tuple(a.wh), downscale=a.downscale, white_bkgd=False, precision=a.precision)
    model = NeRFDownXModel(opt, device="cuda").load_networks(sd_c, sd_f).eval()
    res = model.render_image(cameras.spiral_pose(a.pose_t), cameras.llff_focal(a.wh[0]), ndc=True)
    torch.cuda.synchronize()
    to8 = lambda t: (t.clamp(0, 1) * 255).byte().cpu().numpy()
    Image.fromarray(to8(res["hr_rgb"])).save(a.out + "_hr.png")
    W, H = a.wh
    Image.fromarray(to8(res["lr_rgb"].view(H // a.downscale, W // a.downscale, 3))).save(a.out + "_lr.png")
    print("wrote", a.out + "_hr.png", a.out + "_lr.png")


if __name__ == "__main__":
    main()
