"""GPU record (round 6): the backward chain of the training step on 3 / 2 / 1 MFMA terms per product (precisions
f16x3_bwd3 / _bwd2 / _bwd1, include/nsr_train.h) -- what each does to the gradients and to a 200-step Adam trajectory.

  1. reference-made fixtures (tests/golden/train_*.npz): every gradient tensor against the fp64 oracle, worst tensor, worst
     head tensor, whole network -- the quantities tests/test_gpu_train.py bounds (2e-3 / 5e-4 / 2e-3);
  2. bench scale (2,048 rays, 393,216 sample points): against the layer-by-layer fp32-gradient path on identical draws --
     worst tensor and whole gradient (bounds of test_chain_path_matches_gemm_path_at_bench_scale: 1e-3 / 2e-4);
  3. 200 Adam steps on the analytic scene against the all-fp32 run (tests/trained_field.py), with f16x3_gemm as yardstick.

usage: python scripts/bwd_terms_check.py > profiles/r6_bwd_terms_gpu.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_sr_amd import train as tr, build, ops, cameras
from nerf_sr_amd.weights import make_state_dict, STATE_DICT_SPEC
from oracle import train_oracle as to          # checker only (this script is evidence tooling, not the product)
from tests.util import train_draws
from tests.trained_field import adam_trajectory, trajectory_drift

VARIANTS = ("f16x3_bwd3", "f16x3_bwd2", "f16x3_bwdm", "f16x3_bwd1")
HEAD = ("rgb.0.weight", "rgb.0.bias", "dir_encoding.0.weight", "dir_encoding.0.bias", "xyz_encoding_final.weight",
        "xyz_encoding_final.bias", "sigma.weight", "sigma.bias")
out = {"csrc_sha256": build.source_hash(), "fixtures": {}, "bench_scale": {}, "trajectory": {}}

for case in ("llff_det", "llff_rand", "blender_rand"):
    g = np.load(f"tests/golden/train_{case}.npz")
    sd_c, sd_f = make_state_dict(int(g["seed_coarse"])), make_state_dict(int(g["seed_fine"]))
    draws = train_draws(g)
    _, gc64, gf64 = to.loss_and_grads(sd_c, sd_f, g["rays"], g["target_lr"], int(g["s2"]), 64, 64, bool(g["white_bkgd"]),
                                      float(g["lambda_coarse"]), float(g["lambda_fine"]), dtype=torch.float64, **draws)
    out["fixtures"][case] = {}
    for prec in VARIANTS + ("f16x3_gemm", "fp32"):
        t = tr.Trainer(sd_c, sd_f, white_bkgd=bool(g["white_bkgd"]), downscale=int(round(int(g["s2"]) ** 0.5)),
                       randomized=bool(g["randomized"]), noise_std=float(g["noise_std"]), lr=float(g["lr"]), beta1=float(g["beta1"]),
                       precision=prec)
        t.set_input(torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["target_lr"]).cuda())
        t.loss_and_grads({k: v for k, v in draws.items() if k != "noise_std"})
        worst = worst_head = num = den = 0.0
        worst_k = ""
        for n, ref in enumerate((gc64, gf64)):
            for k in STATE_DICT_SPEC:
                e, nrm = float((t.grads[n][k].cpu().double() - ref[k]).norm()), float(ref[k].norm())
                num, den = num + e * e, den + nrm * nrm
                if nrm > 0 and e / nrm > worst:
                    worst, worst_k = e / nrm, ("c." if n == 0 else "f.") + k
                if k in HEAD and nrm > 0:
                    worst_head = max(worst_head, e / nrm)
        out["fixtures"][case][prec] = {"worst_tensor": worst, "worst_tensor_name": worst_k, "worst_head_tensor": worst_head,
                                       "whole_network": (num / den) ** 0.5}

R = 2048
frame = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True)
sel = torch.randperm(frame.shape[0], generator=torch.Generator().manual_seed(3))[: R // 4].cuda()
rays = frame[sel].reshape(-1, 8).contiguous()
tgt = torch.rand(R // 4, 3, generator=torch.Generator().manual_seed(4)).cuda()
res = {}
for prec in ("f16x3_gemm",) + VARIANTS:
    t = tr.Trainer(make_state_dict(99), make_state_dict(100), randomized=True, noise_std=1.0, ray_chunk=R, precision=prec)
    t.set_input(rays, tgt)
    torch.manual_seed(77)
    t.loss_and_grads()
    torch.cuda.synchronize()
    res[prec] = t
a = res["f16x3_gemm"]
for prec in VARIANTS:
    b = res[prec]
    rec = {"loss_diff": float((a.losses - b.losses).abs().max()), "nets": []}
    for n in range(2):
        num = den = worst = 0.0
        for k in STATE_DICT_SPEC:
            x, y = a.grads[n][k].double(), b.grads[n][k].double()
            e, nx = float((x - y).norm()), float(x.norm())
            num, den = num + e * e, den + nx * nx
            worst = max(worst, e / nx if nx > 0 else 0.0)
        rec["nets"].append({"worst_tensor": worst, "whole_gradient": (num / den) ** 0.5})
    out["bench_scale"][prec] = rec

runs = {p: adam_trajectory(p, steps=200) for p in ("fp32", "f16x3_gemm") + VARIANTS}
for p in ("f16x3_gemm",) + VARIANTS:
    out["trajectory"][p + "_vs_fp32"] = trajectory_drift(runs[p], runs["fp32"])
print(json.dumps(out, indent=1))
