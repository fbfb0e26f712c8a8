"""Development aid: tail of the RGB error distribution vs the oracle on a slab of BASELINE config #2."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
sd_c, sd_f = make_state_dict(99), make_state_dict(100)
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)
lo = 95256
r = rays[lo:lo + n].contiguous()
torch.set_num_threads(32)
o32 = oc.forward_rays(oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), r.cpu(), 64, 64, False)
o64 = oc.forward_rays(oc.to_torch_sd(sd_c, torch.float64), oc.to_torch_sd(sd_f, torch.float64), r.cpu().double(), 64, 64, False)
def stats(name, a, b):
    d = (a.double() - b.double()).abs().max(-1)[0]
    print(f"{name:28s} max {d.max():.2e}  p99.9 {d.quantile(0.999):.2e}  p99 {d.quantile(0.99):.2e}  median {d.median():.2e}  #>1e-4 {(d > 1e-4).sum().item()} / {d.numel()}  psnr {oc.psnr(a, b):.1f} dB")
stats("oracle fp64 vs oracle fp32", o64["fine_comp_rgbs"], o32["fine_comp_rgbs"])
for prec in ("fp32", "f16x3"):
    nc = ops.VanillaMLP(precision=prec).load_state_dict(sd_c); nf = ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
    out = ops.forward_rays(nc, nf, r, 64, 64, False)
    stats(f"hip {prec} vs oracle fp32", out["fine_comp_rgbs"].cpu(), o32["fine_comp_rgbs"])
    stats(f"hip {prec} vs oracle fp64", out["fine_comp_rgbs"].cpu(), o64["fine_comp_rgbs"])
    stats(f"hip {prec} coarse vs oracle", out["coarse_comp_rgbs"].cpu(), o32["coarse_comp_rgbs"])
