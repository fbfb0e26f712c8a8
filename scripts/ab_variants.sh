#!/bin/bash
# A/B of library variants on ONE box, interleaved (development aid): fine-pass launch of forward_rays (network + compositing,
# config #2's 190,512 rays x 128 samples), 10 launches per sample, variants alternated `rounds` times.
# usage: scripts/ab_variants.sh <rounds> <variant> [<variant> ...]     ("" = the product libnsr.so)
R=${GRAFT_REPO_ROOT:-$PWD}
rounds=$1; shift
for r in $(seq 1 $rounds); do
for v in "$@"; do
  lib=$R/nerf_sr_amd/libnsr${v:+_$v}.so
  NSR_LIB_PATH=$lib python - <<PY
import sys, time, torch
sys.path.insert(0, "$R")
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict
net = ops.VanillaMLP(precision="f16x3").load_state_dict(make_state_dict(100))
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)
z = torch.sort(torch.rand(rays.shape[0], 128, device='cuda'), -1)[0].contiguous()
for i in range(3): ops.render_rays_composited(net, rays, z, False)
torch.cuda.synchronize(); t0 = time.time()
for i in range(10): ops.render_rays_composited(net, rays, z, False)
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print("round $r variant '%s': fine pass %.3f ms" % ("$v", dt * 1e3))
PY
done
done
