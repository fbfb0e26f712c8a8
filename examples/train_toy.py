#!/usr/bin/env python3
"""A few hundred training iterations of the HIP path on a synthetic scene (needs an MI355X): the "teacher" is the
synthetic field of nerf_sr_amd.weights rendered by the inference path, the "student" starts from a different seed and
is trained with Trainer.optimize_parameters on random LR-pixel batches of the teacher's frames -- the loop structure of
the reference's train.py (set_input -> optimize_parameters) without its datasets / options / visualiser.

    python examples/train_toy.py [--iters 300] [--batch 512]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import cameras, ops, train  # noqa: E402
from nerf_sr_amd.weights import make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--batch", type=int, default=512, help="LR pixels per step (x4 sub-rays)")
    a = ap.parse_args()
    W, H, s = 504, 378, 2
    teacher_c = ops.VanillaMLP(precision="f16x3").load_state_dict(make_state_dict(99))
    teacher_f = ops.VanillaMLP(precision="f16x3").load_state_dict(make_state_dict(100))
    frames = []
    for t in (0.1, 0.5, 0.9):                                    # three training views
        rays = ops.subpixel_rays(cameras.spiral_pose(t), (W, H), cameras.llff_focal(W), s, True)          # (N_lr, 4, 8)
        out = ops.forward_rays(teacher_c, teacher_f, rays.view(-1, 8), 64, 64, False)
        frames.append((rays, ops.sr_mean(out["fine_comp_rgbs"].clone(), rays.shape[0], s * s)))
    student = train.Trainer(make_state_dict(7, field="plain"), make_state_dict(8, field="plain"), randomized=True, noise_std=1.0,
                            lr=5e-4, ray_chunk=4 * a.batch)
    t0 = time.time()
    for it in range(a.iters):
        rays, target = frames[it % len(frames)]
        sel = torch.randint(0, rays.shape[0], (a.batch,), device="cuda")
        student.set_input(rays[sel], target[sel])
        losses = student.optimize_parameters()
        if it % 50 == 0 or it == a.iters - 1:
            lc, lf = losses.tolist()
            print(f"iter {it:4d}  coarse mse {lc:.5f}  fine mse {lf:.5f}  ({(it + 1) / (time.time() - t0):.1f} it/s)")


if __name__ == "__main__":
    main()
