"""Is a training run a function of its seeds alone?  Trains the analytic scene twice in one process with the caching allocator's
free blocks poisoned in between (NaN / random bytes), and once more after unrelated GPU work; compares the weights bit for bit.
usage (GPU box): python scripts/train_determinism_probe.py [steps=300]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import trained_field as tf
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def poison(fill):
    xs = [torch.full((64 << 20,), fill, device="cuda") for _ in range(24)]      # 6 GiB of fp32
    xs += [torch.full((1 << 20,), fill, device="cuda") for _ in range(64)]
    torch.cuda.synchronize()
    del xs


def same(a, b):
    bad = [k for n in ("sd_coarse", "sd_fine") for k in a[n] if not (a[n][k] == b[n][k]).all()]
    return bad


r0 = tf.train_field("llff", steps=steps)
poison(float("nan"))
r1 = tf.train_field("llff", steps=steps)
print("run 2 (free blocks poisoned with NaN) vs run 1: differing tensors", same(r0, r1))
poison(1e30)
# unrelated work: a render of another field
nc, nf = ops.VanillaMLP(precision="f16x3").load_state_dict(make_state_dict(99)), ops.VanillaMLP(precision="f16x3").load_state_dict(make_state_dict(100))
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).view(-1, 8)
ops.forward_rays(nc, nf, rays[:65536].contiguous(), 64, 64, False)
r2 = tf.train_field("llff", steps=steps)
print("run 3 (after 1e30 poison + a render) vs run 1: differing tensors", same(r0, r2))
print("history run 1", r0["history"][-1], "run 2", r1["history"][-1], "run 3", r2["history"][-1])
