"""CPU study for the training step's next design (DESIGN 7.1 "round 4"): what happens to the weight gradients when BOTH operands
of dW_l = sum_points delta_l (x) a_{l-1} are rounded to fp16 (11 significant bits, one MFMA per product, 2-byte panels)
instead of the present split-fp16 pair (22 bits, three MFMAs, 4-byte panels)?

Everything but those two operands stays exact: the training oracle (oracle/train_oracle.py, torch autograd) runs in fp64 with
torch.nn.functional.linear replaced by a Function whose backward rounds `grad_output` (scaled by a power of two so that its
largest entry sits at 2^14, as the weight-gradient kernel pre-scales its A panel) and `input` to fp16 before the
contraction over the points.  Reported per tensor: |dW_q - dW| / |dW| against the fp64 gradient, next to the same figure
for the plain fp32 oracle (the error class the product is held to: 2e-3 of the norm, tests/test_gpu_train.py).

usage: python scripts/study_fp16_wgrad.py [fixture.npz ...]      (default: tests/golden/train_llff_rand.npz, train_blender_rand.npz)
       python scripts/study_fp16_wgrad.py --trajectory [steps]    (two Adam runs, exact vs fp16-operand weight gradients)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import train_oracle as to  # noqa: E402  (test infrastructure; this script is a study, not the product)

_orig_linear = torch.nn.functional.linear


class LinearFp16Wgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return _orig_linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w
        amax = float(g.abs().max())
        s = 2.0 ** (14 - np.ceil(np.log2(amax))) if amax > 0 else 1.0
        q = lambda t: t.to(torch.float16).to(t.dtype)
        gw = (q(g * s).t() @ q(x)) / s
        return gx, gw, g.sum(0)


def grads(fix, dtype, quantised):
    draws = {k: fix[k] for k in ("u_coarse", "noise_coarse", "u_fine", "noise_fine") if k in fix.files}
    if "noise_std" in fix.files:
        draws["noise_std"] = float(fix["noise_std"])
    from nerf_sr_amd.weights import make_state_dict          # numpy on the host: the fixtures' networks are seeds
    sd_c, sd_f = make_state_dict(int(fix["seed_coarse"])), make_state_dict(int(fix["seed_fine"]))
    torch.nn.functional.linear = (lambda x, w, b=None: LinearFp16Wgrad.apply(x, w, b)) if quantised else _orig_linear
    try:
        _, gc, gf = to.loss_and_grads(sd_c, sd_f, fix["rays"], fix["target_lr"], int(fix["s2"]), fix["u_coarse"].shape[1],
                                      fix["u_fine"].shape[1], bool(fix["white_bkgd"]), float(fix["lambda_coarse"]),
                                      float(fix["lambda_fine"]), dtype=dtype, **draws)
    finally:
        torch.nn.functional.linear = _orig_linear
    return {**{"c." + k: v.double() for k, v in gc.items()}, **{"f." + k: v.double() for k, v in gf.items()}}


def main():
    paths = sys.argv[1:] or [os.path.join(REPO, "tests", "golden", f) for f in ("train_llff_rand.npz", "train_blender_rand.npz")]
    for p in paths:
        fix = np.load(p)
        print(os.path.basename(p), f"({fix['rays'].shape[0]} rays x {fix['u_coarse'].shape[1]} + {fix['u_coarse'].shape[1] + fix['u_fine'].shape[1]} points)")
        ref = grads(fix, torch.float64, False)
        f32 = grads(fix, torch.float32, False)
        q16 = grads(fix, torch.float64, True)
        rel = lambda a, b: float((a - b).norm() / b.norm()) if float(b.norm()) > 0 else 0.0
        worst32 = worst16 = 0.0
        for k in ref:
            if k.endswith("weight"):
                e32, e16 = rel(f32[k], ref[k]), rel(q16[k], ref[k])
                worst32, worst16 = max(worst32, e32), max(worst16, e16)
                print(f"  {k:34s} |dW| {float(ref[k].norm()):9.3e}   fp32 oracle {e32:8.2e}   fp16 operands {e16:8.2e}")
        num = sum(float((q16[k] - ref[k]).norm() ** 2) for k in ref)
        den = sum(float(ref[k].norm() ** 2) for k in ref)
        print(f"  worst tensor: fp32 oracle {worst32:.2e}, fp16 operands {worst16:.2e};  whole network, fp16 operands: {(num / den) ** 0.5:.2e}"
              f"   (gate of tests/test_gpu_train.py: 2e-3)")


if __name__ == "__main__" and "--trajectory" not in sys.argv:
    main()


def trajectory(steps: int = 150, lr_pixels: int = 64, seed: int = 0):
    """Second part (--trajectory [steps]): two Adam runs from the same kaiming start on the same batches and the same random
    draws, fp64 everywhere, one with exact weight gradients and one with fp16-rounded weight-gradient operands, on the analytic
    hard-surface scene of tests/trained_field.py (forward-facing family: NDC rays, s = 2, density noise 1, randomized sampling,
    Adam 5e-4, exactly what the trained-field GPU tests train with the HIP step): how far do the losses and the weights drift
    apart?"""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import trained_field as tf
    from nerf_sr_amd.weights import make_state_dict
    wh, s, ndc, white, nf, noise = tf.FAMILIES["llff"]
    s2, nc, ni = s * s, 64, 64
    rays, tgts = [], []
    for k in range(4):
        c2w, focal = tf.train_pose("llff", k, 4)
        r = to.oc.subpixel_ray_grid(torch.from_numpy(np.asarray(c2w, np.float32)), wh[1], wh[0], focal, s, ndc, *nf)
        rays.append(r.double())
        tgts.append(tf.analytic_colours(r.view(-1, 8), "llff").view(r.shape[0], s2, 3).mean(1).double())
    n_lr, R = rays[0].shape[0], lr_pixels * s2
    runs = {}
    for name, quantised, dtype in (("exact", False, torch.float64), ("fp16 operands", True, torch.float64), ("fp32 oracle", False, torch.float32)):
        pc = {k: torch.as_tensor(v).double() for k, v in make_state_dict(1000 + seed, "plain", bias_scale=0.0).items()}
        pf = {k: torch.as_tensor(v).double() for k, v in make_state_dict(2000 + seed, "plain", bias_scale=0.0).items()}
        mc, vc = ({k: torch.zeros_like(v) for k, v in pc.items()} for _ in range(2))
        mf, vf = ({k: torch.zeros_like(v) for k, v in pf.items()} for _ in range(2))
        gd = torch.Generator().manual_seed(seed + 1)
        losses = []
        for t in range(1, steps + 1):
            k = int(torch.randint(4, (1,), generator=gd))
            idx = torch.randint(n_lr, (lr_pixels,), generator=gd)
            draws = {"u_coarse": torch.rand(R, nc, generator=gd, dtype=torch.float64), "u_fine": torch.rand(R, ni, generator=gd, dtype=torch.float64),
                     "noise_coarse": torch.randn(R, nc, generator=gd, dtype=torch.float64),
                     "noise_fine": torch.randn(R, nc + ni, generator=gd, dtype=torch.float64), "noise_std": float(noise)}
            torch.nn.functional.linear = (lambda x, w, b=None: LinearFp16Wgrad.apply(x, w, b)) if quantised else _orig_linear
            try:
                res, gc, gf = to.loss_and_grads({k_: v.numpy() for k_, v in pc.items()}, {k_: v.numpy() for k_, v in pf.items()},
                                                rays[k][idx].reshape(-1, 8).numpy(), tgts[k][idx].numpy(), s2, nc, ni, bool(white),
                                                dtype=dtype, **draws)
            finally:
                torch.nn.functional.linear = _orig_linear
            to.adam_step(pc, {k_: v.double() for k_, v in gc.items()}, mc, vc, t)
            to.adam_step(pf, {k_: v.double() for k_, v in gf.items()}, mf, vf, t)
            losses.append((res["loss_coarse_mse"], res["loss_fine_mse"]))
            if t % 25 == 0:
                print(f"    [{name}] step {t}: mse coarse {losses[-1][0]:.5f} fine {losses[-1][1]:.5f}", flush=True)
        runs[name] = (losses, pc, pf)
    a = runs["exact"]
    print(f"trajectory: {steps} Adam steps, {R} rays x (64 + 128) points per step, lr 5e-4; 'exact' and 'fp16 operands' in fp64 (only the "
          "weight-gradient operands rounded), 'fp32 oracle' = the whole forward / backward in fp32 (the reference's arithmetic)")
    start = [make_state_dict(1000 + seed, "plain", bias_scale=0.0), make_state_dict(2000 + seed, "plain", bias_scale=0.0)]
    mov = sum(float((a[i][k_] - torch.as_tensor(start[i - 1][k_]).double()).norm() ** 2) for i in (1, 2) for k_ in a[i])
    tail = max(1, steps // 5)
    ma = sum(x[1] for x in a[0][-tail:]) / tail
    for other in ("fp16 operands", "fp32 oracle"):
        b = runs[other]
        print(f" {other} against exact:")
        for t in sorted(t for t in set([1, 2, 5, 10, 25, 50, 100, steps]) if 1 <= t <= steps):
            ea, eb = a[0][t - 1], b[0][t - 1]
            print(f"  step {t:4d}: fine mse exact {ea[1]:.6e}   {other} {eb[1]:.6e}   rel diff {abs(ea[1] - eb[1]) / max(ea[1], 1e-300):.1e}")
        mb = sum(x[1] for x in b[0][-tail:]) / tail
        print(f"  mean fine mse over the last {tail} steps: exact {ma:.6e}, {other} {mb:.6e}  (rel diff {abs(ma - mb) / ma:.1e})")
        num = sum(float((a[i][k_] - b[i][k_]).norm() ** 2) for i in (1, 2) for k_ in a[i])
        print(f"  weights after {steps} steps: |w - w_exact| / |w_exact - w_start| = {(num / mov) ** 0.5:.2e}")


if __name__ == "__main__" and "--trajectory" in sys.argv:
    i = sys.argv.index("--trajectory")
    trajectory(int(sys.argv[i + 1]) if len(sys.argv) > i + 1 else 150)
