"""Development aid: the refinement pass of BASELINE config #5 (169 tiles of an 800 x 800 frame) for rocprofv3, at a given
tile batch size.  usage: prof_refine.py [batch] [reps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import refine, warp, cameras, pipeline
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = H = 800
net = refine.MaxPoolingModel().load_state_dict(refine.make_refine_state_dict(7))
g = torch.Generator().manual_seed(5)
sr = (torch.rand(3, H, W, generator=g) * 2 - 1).cuda()
ref = (torch.rand(3, H, W, generator=g) * 2 - 1).cuda()
depth = (2.0 + 4.0 * torch.rand(H, W, generator=g)).cuda()
c2w, ref_c2w = cameras.spheric_pose(40.0, -30.0, 4.0), cameras.spheric_pose(25.0, -30.0, 4.0)
locs = warp.depth_warp(depth, c2w, pipeline.world_to_camera(ref_c2w), cameras.blender_focal(W), "ray")
for i in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.time()
    out = refine.refine_image(net, sr, ref, locs, batch=batch)
    torch.cuda.synchronize()
    if i: print(f"batch {batch}: {1e3 * (time.time() - t0):.2f} ms per frame")
