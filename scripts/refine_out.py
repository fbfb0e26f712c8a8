"""Development aid: one refinement pass of BASELINE config #5 on fixed inputs: time per frame, output saved for a bit-wise
comparison between library builds (NSR_LIB_PATH).  usage: refine_out.py out.pt [reps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import refine, warp, cameras, pipeline
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W = H = 800
net = refine.MaxPoolingModel().load_state_dict(refine.make_refine_state_dict(7))
g = torch.Generator().manual_seed(5)
sr = (torch.rand(3, H, W, generator=g) * 2 - 1).cuda()
ref = (torch.rand(3, H, W, generator=g) * 2 - 1).cuda()
depth = (2.0 + 4.0 * torch.rand(H, W, generator=g)).cuda()
c2w, ref_c2w = cameras.spheric_pose(40.0, -30.0, 4.0), cameras.spheric_pose(25.0, -30.0, 4.0)
locs = warp.depth_warp(depth, c2w, pipeline.world_to_camera(ref_c2w), cameras.blender_focal(W), "ray")
ts = []
for i in range(reps + 1):
    torch.cuda.synchronize(); t0 = time.time()
    out = refine.refine_image(net, sr, ref, locs, batch=256)
    torch.cuda.synchronize()
    if i: ts.append(1e3 * (time.time() - t0))
print(f"{os.environ.get('NSR_LIB_PATH', 'product')}: {min(ts):.2f} ms min, {sorted(ts)[len(ts) // 2]:.2f} ms median per frame; finite {bool(torch.isfinite(out).all())}")
torch.save(out.cpu(), sys.argv[1])
