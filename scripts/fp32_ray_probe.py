"""Round 6 (VERDICT r5 "next" #2): root cause of the ONE hard violation of the `fp32` kernel in the frame-scale parity record
(profiles/r5_parity_report.json, geometry 5: a ray at |dRGB| 1.63e-4 whose oracle fp32-vs-fp64 gap is 1.6e-6).

Finds the ray again (same block of 65,536 rays of config #5's frame, same field), then walks it stage by stage through the HIP
path (the stage-by-stage C-ABI route: nsr_sample_along_rays -> nsr_render_rays -> nsr_composite -> nsr_resample_along_rays -> ...)
next to the oracle in fp32 and fp64, and CROSS-FEEDS the stages (the oracle's fine pass on the HIP path's fine depths and
vice versa) so that the stage whose output diverges is named, not guessed.
usage (GPU box): python scripts/fp32_ray_probe.py [precision=fp32] [config=5] [N=65536] > gpurun_out/fp32_ray_probe.json
       `config` = llff | blender: the TRAINED field of tests/test_gpu_trained.py (4,000 steps, then the test's block of N rays)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_sr_amd import cameras, ops
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc       # checker (evidence tooling, not the product)
from tests.util import oracle_fp32_and_fp64

PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32"
TRAINED = len(sys.argv) > 2 and sys.argv[2] in ("llff", "blender")
CID = sys.argv[2] if TRAINED else (int(sys.argv[2]) if len(sys.argv) > 2 else 5)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
CONFIGS = {2: ((504, 378), 2, True, False), 3: ((400, 400), 2, False, True), 4: ((1008, 756), 4, True, False), 5: ((800, 800), 4, False, True)}
if TRAINED:
    from tests import trained_field as tf
    res = tf.train_field(CID, steps=4000)
    sd_c, sd_f = res["sd_coarse"], res["sd_fine"]
    blk, ref, ref64 = tf.oracle_block(CID, sd_c, sd_f, N)
    blk = blk.cuda().contiguous()
    white = tf.FAMILIES[CID][3]
else:
    wh, s, ndc, white = CONFIGS[CID]
    if ndc:
        c2w, f, nf = cameras.spiral_pose(0.4), cameras.llff_focal(wh[0]), (0.0, 1.0)
    else:
        c2w, f, nf = cameras.spheric_pose(40.0, -30.0, 4.0), cameras.blender_focal(wh[0]), (2.0, 6.0)
    rays = ops.subpixel_rays(c2w, wh, f, s, ndc, *nf).view(-1, 8)
    lo = (rays.shape[0] // 2) - (rays.shape[0] // 2) % (s * s)
    blk = rays[lo:lo + N].contiguous()
    sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    ref, ref64 = oracle_fp32_and_fp64(sd_c, sd_f, blk.cpu(), white)
nc, nfn = ops.VanillaMLP(precision=PREC).load_state_dict(sd_c), ops.VanillaMLP(precision=PREC).load_state_dict(sd_f)
hip = ops.forward_rays(nc, nfn, blk, 64, 64, white)
d = (hip["fine_comp_rgbs"].cpu().double() - ref["fine_comp_rgbs"].double()).abs().max(-1)[0]
gap = (ref["fine_comp_rgbs"].double() - ref64["fine_comp_rgbs"]).abs().max(-1)[0]
viol = torch.nonzero((d > 1e-4) & (d > 2 * gap)).flatten().tolist()
order = torch.argsort(d, descending=True)[:4].tolist()
out = {"precision": PREC, "config": CID, "rays": N, "violations": viol, "top_rays": [{"i": i, "d": float(d[i]), "gap": float(gap[i])} for i in order], "probes": []}


def mx(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


for i in (viol or order[:1]):
    r = blk[i:i + 1].contiguous()
    rc = r.cpu()
    o_, d_, near, far = r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8]
    # ---- HIP, stage by stage
    z_h, _ = ops.sample_along_rays(o_, d_, near, far, 64, False, False)
    rgb_h, sig_h = ops.render_rays(nc, r, z_h)
    comp_h = ops.VolumetricRenderer()(rgb_h.contiguous(), sig_h.contiguous(), z_h, white)
    zf_h, _ = ops.resample_along_rays(o_, d_, z_h, comp_h[3], 64, False)
    rgbf_h, sigf_h = ops.render_rays(nfn, r, zf_h)
    compf_h = ops.VolumetricRenderer()(rgbf_h.contiguous(), sigf_h.contiguous(), zf_h, white)
    staged_vs_fused = mx(compf_h[0], hip["fine_comp_rgbs"][i:i + 1])
    rec = {"i": i, "d": float(d[i]), "oracle_gap": float(gap[i]), "staged_route_equals_fused": staged_vs_fused}
    stages = {}
    for tag, dt in (("o32", torch.float32), ("o64", torch.float64)):
        sc, sf = oc.to_torch_sd(sd_c, dt), oc.to_torch_sd(sd_f, dt)
        rr = rc.to(dt)
        with torch.no_grad():
            z, xyz = oc.sample_coarse(rr[:, 0:3], rr[:, 3:6], rr[:, 6:7], rr[:, 7:8], 64)
            de = oc.posenc(rr[:, 8:11] if rr.shape[1] == 11 else rr[:, 3:6], 4)
            rgb, sig = oc.render_points(sc, xyz, de)
            comp = oc.composite(rgb, sig, z, white)
            zf, xyzf = oc.resample_fine(rr[:, 0:3], rr[:, 3:6], z, comp[3], 64)
            rgbf, sigf = oc.render_points(sf, xyzf, de)
            compf = oc.composite(rgbf, sigf, zf, white)
            # cross-feed: this oracle's fine pass on the HIP path's fine depths / coarse weights
            zf_x = zf_h.cpu().to(dt)
            xyz_x = oc.points_on_rays(rr[:, 0:3], rr[:, 3:6], zf_x)
            rgbx, sigx = oc.render_points(sf, xyz_x, de)
            compx = oc.composite(rgbx, sigx, zf_x, white)
            zf_w, _ = oc.resample_fine(rr[:, 0:3], rr[:, 3:6], z, comp_h[3].cpu().to(dt), 64)
            amp, margin, width = oc.resample_conditioning(z, comp[3], 64)
        stages[tag] = dict(z=z, sig=sig, rgb=rgb, w=comp[3], zf=zf, sigf=sigf, rgbf=rgbf, wf=compf[3], out=compf[0], out_on_hip_zf=compx[0],
                           zf_from_hip_w=zf_w, amp=float(amp), margin=float(margin), width=float(width))
    a, b = stages["o32"], stages["o64"]
    rec["conditioning_fp64"] = {"amp": b["amp"], "margin_to_snap": b["margin"], "max_bin_width": b["width"]}
    rec["hip_vs_o32"] = {"z_coarse": mx(z_h, a["z"]), "sigma_coarse": mx(sig_h, a["sig"]), "rgb_coarse": mx(rgb_h, a["rgb"]),
                         "weights_coarse": mx(comp_h[3], a["w"]), "z_fine": mx(zf_h, a["zf"]), "sigma_fine": mx(sigf_h, a["sigf"]),
                         "rgb_fine_points": mx(rgbf_h, a["rgbf"]), "weights_fine": mx(compf_h[3], a["wf"]), "out": mx(compf_h[0], a["out"])}
    rec["o32_vs_o64"] = {"sigma_coarse": mx(a["sig"], b["sig"]), "weights_coarse": mx(a["w"], b["w"]), "z_fine": mx(a["zf"], b["zf"]),
                         "sigma_fine": mx(a["sigf"], b["sigf"]), "weights_fine": mx(a["wf"], b["wf"]), "out": mx(a["out"], b["out"])}
    # cross-feeds: who moves the colour?
    rec["cross"] = {"o32_fine_pass_on_HIP_z_fine_vs_HIP_out": mx(a["out_on_hip_zf"], compf_h[0]),
                    "o32_fine_pass_on_HIP_z_fine_vs_o32_out": mx(a["out_on_hip_zf"], a["out"]),
                    "o64_fine_pass_on_HIP_z_fine_vs_o64_out": mx(b["out_on_hip_zf"], b["out"]),
                    "o32_resampler_on_HIP_coarse_weights_vs_HIP_z_fine": mx(a["zf_from_hip_w"], zf_h),
                    "o32_resampler_on_HIP_coarse_weights_vs_o32_z_fine": mx(a["zf_from_hip_w"], a["zf"])}
    dz = (zf_h.cpu().double() - a["zf"].double()).abs()[0]
    k = int(dz.argmax())
    rec["z_fine_largest_move"] = {"index": k, "hip": float(zf_h[0, k]), "o32": float(a["zf"][0, k]), "o64": float(b["zf"][0, k]),
                                  "bin_width_there": float((a["z"][0, 1:] - a["z"][0, :-1]).max())}
    # where the fine colour is sensitive: the largest fine weights and the density around them
    wf = a["wf"][0]
    top = torch.argsort(wf, descending=True)[:4].tolist()
    rec["fine_weight_peaks_o32"] = [{"k": t, "w": float(wf[t]), "sigma_o32": float(a["sigf"][0, t]), "sigma_hip": float(sigf_h[0, t]),
                                     "sigma_o64": float(b["sigf"][0, t]), "z": float(a["zf"][0, t]), "z_hip": float(zf_h[0, t])} for t in top]
    # the coarse pdf around the snap threshold
    w = (a["w"][0, 1:-1].double() + 1e-5)
    pdf = w / w.sum()
    rec["coarse_pdf"] = {"bins_below_2e-5": int((pdf < 2e-5).sum()), "min_pdf": float(pdf.min()), "closest_to_1e-5": float((pdf - 1e-5).abs().min())}
    out["probes"].append(rec)
print(json.dumps(out, indent=1))
