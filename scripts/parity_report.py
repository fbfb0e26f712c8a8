"""Parity report at the BASELINE frame geometries: HIP path (each contract-grade precision) vs the CPU oracle on a
contiguous block of rays from the middle of one frame per configuration (the same protocol as
tests/test_gpu_frames.py, at a larger block: `parity_report.py 65536`).  Writes gpurun_out/parity_report.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
CONFIGS = {2: ((504, 378), 2, True, False), 3: ((400, 400), 2, False, True), 4: ((1008, 756), 4, True, False),
           5: ((800, 800), 4, False, True)}
sd_c, sd_f = make_state_dict(99), make_state_dict(100)
torch.set_num_threads(32)
rep = {"rays_per_config": N, "weights": "make_state_dict(99 / 100), 'smooth' field", "configs": {}}
for cid, (wh, s, ndc, white) in CONFIGS.items():
    if ndc: c2w, f, nf = cameras.spiral_pose(0.4), cameras.llff_focal(wh[0]), (0.0, 1.0)
    else: c2w, f, nf = cameras.spheric_pose(40.0, -30.0, 4.0), cameras.blender_focal(wh[0]), (2.0, 6.0)
    rays = ops.subpixel_rays(c2w, wh, f, s, ndc, *nf).view(-1, 8)
    lo = (rays.shape[0] // 2) - (rays.shape[0] // 2) % (s * s)
    blk = rays[lo:lo + N].contiguous()
    t0 = time.time()
    with torch.no_grad():
        ref = oc.forward_rays(oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), blk.cpu(), 64, 64, white)
        ref64 = oc.forward_rays(oc.to_torch_sd(sd_c, torch.float64), oc.to_torch_sd(sd_f, torch.float64), blk.cpu().double(), 64, 64, white)
    t_cpu = time.time() - t0
    d_or = (ref["fine_comp_rgbs"].double() - ref64["fine_comp_rgbs"]).abs().max(-1)[0]
    entry = {"img_wh": wh, "downscale": s, "oracle_seconds": round(t_cpu, 1),
             "oracle_fp32_vs_fp64_fine": {"max": float(d_or.max()), "p999": float(torch.quantile(d_or, 0.999)), "median": float(d_or.median()),
                                          "rays_over_1e-4": int((d_or > 1e-4).sum())}}
    for prec in ("f16x3", "fp32"):
        nc = ops.VanillaMLP(precision=prec).load_state_dict(sd_c); nf_ = ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
        o = ops.forward_rays(nc, nf_, blk, 64, 64, white)
        e = {}
        for k in ("coarse_comp_rgbs", "fine_comp_rgbs", "fine_opacity"):
            d = (o[k].cpu().double() - ref[k].double()).abs()
            d = d.max(-1)[0] if d.ndim == 2 else d
            e[k] = {"max": float(d.max()), "p999": float(torch.quantile(d, 0.999)), "median": float(d.median())}
        dh = (o["fine_comp_rgbs"].cpu().double() - ref["fine_comp_rgbs"].double()).abs().max(-1)[0]
        over = dh > 1e-4
        exempt = d_or > 1e-4          # the oracle's own fp32 evaluation is > 1e-4 away from its fp64 evaluation
        e["fine_rays_over_1e-4"] = int(over.sum())
        e["fine_rays_over_1e-4_not_exempt"] = int((over & ~exempt).sum())
        e["fine_max_not_exempt"] = float(dh[~exempt].max())
        # on those rays: how far the reference's own fp32 evaluation is from its fp64 evaluation (resampling conditioning)
        e["oracle_fp32_vs_fp64_on_those_rays"] = [float(x) for x in d_or[over].tolist()[:8]]
        e["vs_fp64_oracle_max"] = float((o["fine_comp_rgbs"].cpu().double() - ref64["fine_comp_rgbs"]).abs().max())
        e["psnr_fine_vs_oracle_db"] = oc.psnr(o["fine_comp_rgbs"].cpu(), ref["fine_comp_rgbs"])
        lr_h = ops.sr_mean(o["fine_comp_rgbs"].clone(), N // (s * s), s * s).cpu()
        e["lr_max"] = float((lr_h - oc.sr_mean(ref["fine_comp_rgbs"], N // (s * s), s * s)).abs().max())
        entry[prec] = e
    rep["configs"][str(cid)] = entry
    print(cid, {p: (entry[p]["fine_comp_rgbs"]["max"], entry[p]["coarse_comp_rgbs"]["max"]) for p in ("f16x3", "fp32")}, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/parity_report.json", "w"), indent=1)
