"""Development aid: training GEMM vs torch matmul, then loss/gradients/Adam of the HIP path vs the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_sr_amd import train as tr
from nerf_sr_amd.weights import make_state_dict, STATE_DICT_SPEC, FLOP_PER_POINT
from oracle import train_oracle as to
from tests.util import train_draws

torch.manual_seed(0)
for P, K, N in ((1000, 64, 256), (4096, 256, 256), (777, 320, 256), (640, 288, 128), (512, 128, 32), (300, 256, 288)):
    x = torch.randn(P, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    if P % 4: P4 = P
    y, yt = tr.linear(x, w, b, act=1, transposed=True) if P % 4 == 0 else (tr.linear(x, w, b, act=1), None)
    ref = torch.relu(x.double() @ w.double().T + b.double())
    e = (y.double() - ref).abs().max().item()
    et = (yt.double().T - ref).abs().max().item() if yt is not None else -1
    print(f"gemm P={P} K={K} N={N}: max err {e:.2e} (transposed copy {et:.2e})")

for case in ("llff_det", "llff_rand", "blender_rand"):
    g = np.load(f"tests/golden/train_{case}.npz")
    sd_c, sd_f = make_state_dict(int(g["seed_coarse"])), make_state_dict(int(g["seed_fine"]))
    draws = train_draws(g)
    res, gc, gf = to.loss_and_grads(sd_c, sd_f, g["rays"], g["target_lr"], int(g["s2"]), 64, 64, bool(g["white_bkgd"]),
                                    float(g["lambda_coarse"]), float(g["lambda_fine"]), **draws)
    res64, gc64, gf64 = to.loss_and_grads(sd_c, sd_f, g["rays"], g["target_lr"], int(g["s2"]), 64, 64, bool(g["white_bkgd"]),
                                          float(g["lambda_coarse"]), float(g["lambda_fine"]), dtype=torch.float64, **draws)
    t = tr.Trainer(sd_c, sd_f, white_bkgd=bool(g["white_bkgd"]), downscale=int(round(int(g["s2"]) ** 0.5)),
                   randomized=bool(g["randomized"]), noise_std=float(g["noise_std"]), lr=float(g["lr"]), beta1=float(g["beta1"]))
    t.set_input(torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["target_lr"]).cuda())
    t.loss_and_grads({k: v for k, v in draws.items() if k != "noise_std"})
    torch.cuda.synchronize()
    print(case, "losses hip", t.losses.tolist(), "oracle", res["loss_coarse_mse"], res["loss_fine_mse"])
    for k in ("fine_comp_rgbs", "coarse_comp_rgbs", "fine_weights"):
        print("   ", k, "max err %.2e" % (t.out[k].cpu() - res[k]).abs().max())
    for name, grads, ref, ref64 in (("coarse", t.grads[0], gc, gc64), ("fine", t.grads[1], gf, gf64)):
        worst = 0
        for k in STATE_DICT_SPEC:
            a = grads[k].cpu().double(); b = ref64[k]; o32 = ref[k].double()
            rel = (a - b).norm() / max(b.norm().item(), 1e-30)
            rel32 = (o32 - b).norm() / max(b.norm().item(), 1e-30)
            worst = max(worst, rel.item())
            if rel > 1e-4: print("      ", name, k, "rel err vs fp64 oracle %.2e (fp32 oracle %.2e) |g| %.2e" % (rel, rel32, b.norm()))
        print("   ", name, "worst relative gradient error vs fp64 oracle %.2e" % worst)

# speed: config-#1-like batch, 1024 LR pixels x 4 sub-rays
R = 4096
t = tr.Trainer(make_state_dict(99), make_state_dict(100), randomized=True, noise_std=1.0, ray_chunk=4096)
from nerf_sr_amd import ops, cameras
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)[:R].contiguous()
t.set_input(rays, torch.rand(R // 4, 3, device="cuda"))
for i in range(2): t.optimize_parameters()
torch.cuda.synchronize(); t0 = time.time(); n = 5
for i in range(n): t.optimize_parameters()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"train step {R} rays: {dt*1e3:.1f} ms  {R/dt:.0f} rays/s  {R*192*FLOP_PER_POINT*3/dt/1e12:.1f} TFLOP/s (3x forward flops)  losses {t.losses.tolist()}")
