#!/bin/bash
# time the fine-pass MLP for each ablation library (development aid)
R=${GRAFT_REPO_ROOT:-$PWD}
for v in "" _nodrain _nobar _nodrainbar _nodma _noconv; do
  NSR_LIB_PATH=$R/nerf_sr_amd/libnsr$v.so python - <<PY
import sys, time, torch
sys.path.insert(0, "$R")
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict
net = ops.VanillaMLP(precision="f16x3").load_state_dict(make_state_dict(100))
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)
z = torch.sort(torch.rand(rays.shape[0], 128, device='cuda'), -1)[0].contiguous()
for i in range(2): ops.render_rays(net, rays, z)
torch.cuda.synchronize(); t0 = time.time()
for i in range(5): ops.render_rays(net, rays, z)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print("variant '%s': fine pass %.2f ms" % ("$v", dt * 1e3))
PY
done
