"""CPU study (round 6): how many MFMA terms does the backward CHAIN of the training step need?

The chain kernel (csrc/nsr_train_chain.hip) computes dz_{l-1} = (W_l^T dz_l) * mask with split-fp16 operands, three MFMAs per
product: W_hi g_hi + W_hi g_lo + W_lo g_hi (fp32-grade).  Candidates:
   3 terms  -- today
   2 terms  -- W_hi g_hi + W_lo g_hi: the weights keep 22 bits, the incoming gradient is rounded to fp16 per layer
   1 term   -- W_hi g_hi: both operands rounded to 11 bits
Emulated with the training oracle (oracle/train_oracle.py, torch autograd) in fp64: torch.nn.functional.linear is replaced by
a Function whose backward rounds exactly those operands (the gradient scaled per POINT by a power of two, as the kernel's
per-point phi does; the weights after their 2^6 stream scale), AND keeps the weight-gradient operand rounding the product
already has (round 5: both operands of dW rounded to fp16).  Reported per tensor: |dW_q - dW| / |dW| against the exact fp64
gradient, next to the gate of tests/test_gpu_train.py (2e-3 of the norm; heads 5e-4).

usage: python scripts/study_bwd_terms.py [fixture.npz ...]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import train_oracle as to  # noqa: E402  (test infrastructure; this script is a study, not the product)

_orig_linear = torch.nn.functional.linear
q16 = lambda t: t.to(torch.float16).to(t.dtype)


def point_scale(g):
    """power of two per row (point) that puts the row's largest magnitude in [2^-5, 2^-4) -- where the chain kernel's phi
    leaves its operands (stage_factors: e = -4 - E)"""
    amax = g.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    return torch.exp2(-5.0 - torch.floor(torch.log2(amax)))


class LinearTerms(torch.autograd.Function):
    terms = 3

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return _orig_linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        t = LinearTerms.terms
        s = point_scale(g)
        gh = q16(g * s)
        if t == 3:
            gx = g @ w
        elif t == 2:
            gx = (gh @ w) / s
        else:
            gx = (gh @ (q16(w * 64.0) / 64.0)) / s
        # the weight gradient as the product computes it since round 5: both operands rounded to fp16
        amax = float(g.abs().max())
        sw = 2.0 ** (14 - np.ceil(np.log2(amax))) if amax > 0 else 1.0
        gw = (q16(g * sw).t() @ q16(x)) / sw
        return gx, gw, g.sum(0)


def grads(fix, dtype, terms):
    draws = {k: fix[k] for k in ("u_coarse", "noise_coarse", "u_fine", "noise_fine") if k in fix.files}
    if "noise_std" in fix.files:
        draws["noise_std"] = float(fix["noise_std"])
    from nerf_sr_amd.weights import make_state_dict
    sd_c, sd_f = make_state_dict(int(fix["seed_coarse"])), make_state_dict(int(fix["seed_fine"]))
    LinearTerms.terms = terms
    torch.nn.functional.linear = (lambda x, w, b=None: LinearTerms.apply(x, w, b)) if terms else _orig_linear
    try:
        _, gc, gf = to.loss_and_grads(sd_c, sd_f, fix["rays"], fix["target_lr"], int(fix["s2"]), fix["u_coarse"].shape[1],
                                      fix["u_fine"].shape[1], bool(fix["white_bkgd"]), float(fix["lambda_coarse"]),
                                      float(fix["lambda_fine"]), dtype=dtype, **draws)
    finally:
        torch.nn.functional.linear = _orig_linear
    return {**{"c." + k: v.double() for k, v in gc.items()}, **{"f." + k: v.double() for k, v in gf.items()}}


def main():
    paths = sys.argv[1:] or [os.path.join(REPO, "tests", "golden", f) for f in ("train_llff_rand.npz", "train_blender_rand.npz", "train_llff_det.npz")]
    for p in paths:
        fix = np.load(p)
        print(os.path.basename(p))
        ref = grads(fix, torch.float64, 0)
        runs = {t: grads(fix, torch.float64, t) for t in (3, 2, 1)}
        rel = lambda a, b: float((a - b).norm() / b.norm()) if float(b.norm()) > 0 else 0.0
        worst = {t: 0.0 for t in runs}
        for k in ref:
            e = {t: rel(runs[t][k], ref[k]) for t in runs}
            for t in runs:
                worst[t] = max(worst[t], e[t])
            print(f"  {k:34s} 3 terms {e[3]:8.2e}   2 terms {e[2]:8.2e}   1 term {e[1]:8.2e}")
        for t in runs:
            num = sum(float((runs[t][k] - ref[k]).norm() ** 2) for k in ref)
            den = sum(float(ref[k].norm() ** 2) for k in ref)
            print(f"  {t} term(s): worst tensor {worst[t]:.2e}, whole network {(num / den) ** 0.5:.2e}   (gates: 2e-3 per tensor, 5e-4 heads, 2e-3 network)")


if __name__ == "__main__" and "--trajectory" not in sys.argv:
    main()


def trajectory_main(terms: int, steps: int):
    """--trajectory TERMS [steps]: scripts/study_fp16_wgrad.py's 200-step Adam comparison (exact fp64 run vs a run with rounded
    operands vs the plain fp32 oracle) with THIS file's operand rounding: backward-chain products on TERMS MFMA terms plus the
    fp16 weight-gradient operands the product already has."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("study_fp16_wgrad", os.path.join(REPO, "scripts", "study_fp16_wgrad.py"))
    sys.argv = [sys.argv[0]]            # that file's __main__ guards look at argv
    mod = importlib.util.module_from_spec(spec)
    mod.__name__ = "study_fp16_wgrad"
    spec.loader.exec_module(mod)
    LinearTerms.terms = terms
    mod.LinearFp16Wgrad = LinearTerms
    print(f"backward chain on {terms} term(s) + fp16 weight-gradient operands ('fp16 operands' below)")
    mod.trajectory(steps)


if __name__ == "__main__" and "--trajectory" in sys.argv:
    i = sys.argv.index("--trajectory")
    trajectory_main(int(sys.argv[i + 1]), int(sys.argv[i + 2]) if len(sys.argv) > i + 2 else 200)
