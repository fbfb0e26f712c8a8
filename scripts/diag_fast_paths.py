import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc
sd_c, sd_f = make_state_dict(99), make_state_dict(100)
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)
outs = {}
for prec in ("fp32", "f16", "bf16", "f16x3"):
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(sd_c)
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
    o = ops.forward_rays(net_c, net_f, rays, 64, 64, False)
    outs[prec] = (ops.sr_mean(o["fine_comp_rgbs"].clone(), rays.shape[0] // 4, 4).cpu(), o["fine_comp_rgbs"].clone().cpu(), o["coarse_comp_rgbs"].clone().cpu())
for prec in ("f16", "bf16", "f16x3"):
    for i, nm in enumerate(("lr", "fine", "coarse")):
        d = (outs[prec][i] - outs["fp32"][i]).abs().max(-1)[0]
        q = torch.quantile(d, torch.tensor([0.5, 0.9, 0.99, 0.999]))
        print(prec, nm, "psnr %.1f" % oc.psnr(outs[prec][i], outs["fp32"][i]), "median %.2e p90 %.2e p99 %.2e p99.9 %.2e max %.2e" % (*q.tolist(), d.max()), "frac>5e-3 %.2e frac>1e-4 %.2e" % ((d > 5e-3).float().mean(), (d > 1e-4).float().mean()))
