"""Scratch timing of forward_rays on config #2 (development aid; bench.py is the contract)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict, FLOP_PER_POINT
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
net_c = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(99))
net_f = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(100))
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)
R = rays.shape[0]
ws = torch.empty(ops._lib.load().nsr_forward_rays_workspace_bytes(R, 64, 64), dtype=torch.uint8, device='cuda')
outs = {}
for i in range(2):
    ops.forward_rays(net_c, net_f, rays, 64, 64, False, workspace=ws, outs=outs)
torch.cuda.synchronize()
t0 = time.time(); n = 3
for i in range(n):
    ops.forward_rays(net_c, net_f, rays, 64, 64, False, workspace=ws, outs=outs)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"{prec}: {dt*1e3:.1f} ms/image  {R/dt:.0f} rays/s  {R/dt*192*FLOP_PER_POINT/1e12:.1f} TFLOP/s")
