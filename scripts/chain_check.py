"""Development aid: the training step's CHAIN path (fused forward + backward chain, nsr_train_chain.hip) against the
layer-by-layer GEMM path (precision f16x3_gemm) and the fp64 CPU oracle, tensor by tensor, then the step time of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_sr_amd import train as tr
from nerf_sr_amd.weights import make_state_dict, STATE_DICT_SPEC
from oracle import train_oracle as to
from tests.util import train_draws


def run(path, g, sd_c, sd_f, draws, precision=None):
    precision = precision or ("f16x3_gemm" if path == "gemm" else "f16x3")
    t = tr.Trainer(sd_c, sd_f, white_bkgd=bool(g["white_bkgd"]), downscale=int(round(int(g["s2"]) ** 0.5)),
                   randomized=bool(g["randomized"]), noise_std=float(g["noise_std"]), lr=float(g["lr"]), beta1=float(g["beta1"]),
                   precision=precision)
    t.set_input(torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["target_lr"]).cuda())
    t.loss_and_grads({k: v for k, v in draws.items() if k != "noise_std"})
    torch.cuda.synchronize()
    return t


cases = sys.argv[1:] or ["llff_det", "llff_rand", "blender_rand"]
for case in cases:
    g = np.load(f"tests/golden/train_{case}.npz")
    sd_c, sd_f = make_state_dict(int(g["seed_coarse"])), make_state_dict(int(g["seed_fine"]))
    draws = train_draws(g)
    res64, gc64, gf64 = to.loss_and_grads(sd_c, sd_f, g["rays"], g["target_lr"], int(g["s2"]), 64, 64, bool(g["white_bkgd"]),
                                          float(g["lambda_coarse"]), float(g["lambda_fine"]), dtype=torch.float64, **draws)
    ts = {p: run(p, g, sd_c, sd_f, draws) for p in ("gemm", "chain")}
    print(f"== {case}: losses oracle {res64['loss_coarse_mse']:.8f} {res64['loss_fine_mse']:.8f}")
    for p, t in ts.items():
        print(f"   {p:5s} losses {t.losses.tolist()}  " + "  ".join(
            f"{k} {float((t.out[k].cpu().double() - res64[k]).abs().max()):.1e}" for k in ("coarse_comp_rgbs", "fine_comp_rgbs", "fine_weights")))
    for n, (name, ref) in enumerate((("coarse", gc64), ("fine", gf64))):
        for k in STATE_DICT_SPEC:
            b = ref[k]
            nrm = max(float(b.norm()), 1e-30)
            e = {p: float((t.grads[n][k].cpu().double() - b).norm()) / nrm for p, t in ts.items()}
            bad = not np.isfinite(e["chain"]) or e["chain"] > max(3 * e["gemm"], 3e-4)
            print(f"   {name:6s} {k:28s} |g| {nrm:.2e}  gemm {e['gemm']:.1e}  chain {e['chain']:.1e}{'   <-- ' if bad else ''}")
            if bad and b.shape[0] == 256:   # per 32-feature output block, then per register position inside the blocks
                d = (ts["chain"].grads[n][k].cpu().double() - b).reshape(256, -1)
                blk = [float(d[32 * i:32 * i + 32].norm()) / nrm for i in range(8)]
                pos = [float(d[[32 * i + j for i in range(8)]].norm()) / nrm for j in range(32)]
                print("          per block:", " ".join(f"{x:.1e}" for x in blk))
                print("          per row-in-block:", " ".join(f"{x:.0e}" for x in pos))

# speed: 2,048 rays, 64 + 128 samples (the bench's training shape)
from nerf_sr_amd import ops, cameras
R = 2048
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)[:R].contiguous()
for p in ("gemm", "chain"):
    t = tr.Trainer(make_state_dict(99), make_state_dict(100), N_importance=128, randomized=True, noise_std=1.0,
                   precision="f16x3_gemm" if p == "gemm" else "f16x3")
    t.set_input(rays, torch.rand(R // 4, 3, device="cuda"))
    for i in range(3):
        t.optimize_parameters()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 10
    for i in range(n):
        t.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print(f"{p}: train step {R} rays: {dt * 1e3:.2f} ms  {R / dt:.0f} rays/s  losses {t.losses.tolist()}")
