"""Development aid: a few fine-pass MLP launches for rocprofv3 (kernel-trace / PMC passes)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
net = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(100))
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)
z = torch.sort(torch.rand(rays.shape[0], 128, device='cuda'), -1)[0].contiguous()
fused = prec in ("fp32", "f16x3")      # the launch forward_rays makes: network + compositing of the tile's own rays
for i in range(n):
    if fused: ops.render_rays_composited(net, rays, z, False)
    else: ops.render_rays(net, rays, z)
torch.cuda.synchronize()
