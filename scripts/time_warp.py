"""Development aid: depth-warp kernel time and achieved HBM bandwidth (52 algorithmic bytes per pixel)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import warp, cameras
for (W, H) in ((504, 378), (1008, 756), (4032, 3024)):
    depth = torch.rand(H, W, device="cuda") * 0.7 + 0.2
    ref = torch.rand(3, H, W, device="cuda")
    c2w = cameras.spiral_pose(0.9).astype(np.float32)
    ref_w2c = np.linalg.inv(np.concatenate([cameras.spiral_pose(0.1), np.array([[0, 0, 0, 1.0]])], 0))[:3]
    f = cameras.llff_focal(W)
    for _ in range(3): warp.depth_warp(depth, c2w, ref_w2c, f, True, ref)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 20; e0.record()
    for _ in range(n): warp.depth_warp(depth, c2w, ref_w2c, f, True, ref)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{W}x{H}: {ms*1e3:.1f} us per image (incl. output allocation), {H*W*52/ms/1e6:.1f} GB/s of 52 B/pixel")
