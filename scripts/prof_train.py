"""Development aid: a few training iterations for rocprofv3 (kernel-trace / stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import train as tr, ops, cameras
from nerf_sr_amd.weights import make_state_dict
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prec = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
t = tr.Trainer(make_state_dict(99), make_state_dict(100), randomized=True, noise_std=1.0, ray_chunk=R, precision=prec)
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)[:R].contiguous()
t.set_input(rays, torch.rand(R // 4, 3, device="cuda"))
for i in range(n): t.optimize_parameters()
torch.cuda.synchronize()
