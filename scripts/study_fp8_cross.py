"""CPU emulation of the "two MFMAs per product" idea for the headline kernel: keep
a_hi * b_hi on the fp16 matrix pipe and compute the two cross terms a_hi * b_lo + a_lo * b_hi as ONE K-concatenated fp8
(e4m3) MFMA -- 32 cycles instead of 64, i.e. two fp16-equivalents per product instead of three.

What is emulated, layer by layer, inside the oracle's own `forward_rays` (torch.nn.functional.linear is replaced for the
duration of a run; everything else -- encodings, sampling, compositing, resampling -- is the fp32 oracle):

  a = 2^6 W, b = the layer's input; hi = RN_f16(v), lo = RN_f16(v - hi)                      (the kernel's split)
  "f16x3":      y = a_hi b_hi + a_hi b_lo + a_lo b_hi                                          (what the kernel issues)
  "fp8 cross":  y = a_hi b_hi + q8(a_hi) q8(b_lo) + q8(a_lo) q8(b_hi)
  "fp8 cross, lo exact":  the lo factors keep fp16, only the hi factor of each cross term is rounded to e4m3
                (a lower bound on the error of any fp8 form: the 11-bit hi factor cannot be carried by a 4-bit operand)
  "f16":        y = a_hi b_hi                                                                  (the one-MFMA fast path)

q8 = round to e4m3 (torch.float8_e4m3fn) after an exact power-of-two scale chosen PER ROW of the operand (per output feature
for weights, per point for activations: the best case -- the kernel would have to carry those scales) that puts the row's
largest magnitude in [128, 256).  Products of the emulated operands are formed in fp32 (exact: <= 22 significant bits) and
accumulated in fp32, as the MFMA does.

Reported: |dRGB| of the fine colours against the plain fp32 oracle on N consecutive rays from the middle of a frame of
BASELINE config #2 (504 x 378, 2 x 2 sub-pixels, NDC) with the synthetic field the parity tests use, and the number of rays
over the contract's 1e-4.

usage: python scripts/study_fp8_cross.py [N=4096] [out.txt]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nerf_sr_amd.weights import make_state_dict  # noqa: E402  (numpy on the host)
from oracle import nerf_oracle as oc  # noqa: E402  (test infrastructure; this script is a study, not the product)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
OUT = sys.argv[2] if len(sys.argv) > 2 else None
_orig_linear = torch.nn.functional.linear


def split(v):
    hi = v.to(torch.float16).float()
    lo = (v - hi).to(torch.float16).float()
    return hi, lo


def q8(v):
    """e4m3 rounding of each row at its own power-of-two scale (largest magnitude of the row in [128, 256))."""
    amax = v.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(7 - torch.floor(torch.log2(amax)))
    return (v * s).to(torch.float8_e4m3fn).float() / s


def make_linear(mode):
    def lin(x, w, b=None):
        if mode == "fp32" or w.shape[0] < 4:      # the 1- and 3-row heads: sigma rides the MFMA in the kernel, rgb is fp32 VALU;
            if w.shape[0] == 3 or mode == "fp32":  # emulate sigma like the rest, rgb exactly
                return _orig_linear(x, w, b)
        a_hi, a_lo = split(w * 64.0)
        b_hi, b_lo = split(x)
        y = b_hi @ a_hi.t()
        if mode == "f16x3":
            y = y + (b_lo @ a_hi.t() + b_hi @ a_lo.t())
        elif mode == "fp8_cross":
            y = y + (q8(b_lo) @ q8(a_hi).t() + q8(b_hi) @ q8(a_lo).t())
        elif mode == "fp8_cross_lo_exact":
            y = y + (b_lo @ q8(a_hi).t() + q8(b_hi) @ a_lo.t())
        elif mode != "f16":
            raise ValueError(mode)
        y = y * (1.0 / 64.0)
        return y if b is None else y + b
    return lin


def run(mode, sd_c, sd_f, rays, dtype=torch.float32):
    torch.nn.functional.linear = make_linear(mode) if mode != "oracle" else _orig_linear
    try:
        with torch.no_grad():
            return oc.forward_rays(oc.to_torch_sd(sd_c, dtype), oc.to_torch_sd(sd_f, dtype), rays.to(dtype), 64, 64, False)["fine_comp_rgbs"]
    finally:
        torch.nn.functional.linear = _orig_linear


def main():
    from nerf_sr_amd import cameras                    # pure host code (no GPU, no library)
    wh, s = (504, 378), 2
    # the oracle's own ray generator (the product's needs the GPU)
    c2w = torch.as_tensor(np.asarray(cameras.spiral_pose(0.4)), dtype=torch.float32)
    rays = oc.subpixel_ray_grid(c2w, wh[1], wh[0], cameras.llff_focal(wh[0]), s, True, 0.0, 1.0).reshape(-1, 8)
    lo = (rays.shape[0] // 2) - (rays.shape[0] // 2) % (s * s)
    rays = rays[lo:lo + N].contiguous()
    sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    ref = run("oracle", sd_c, sd_f, rays)
    ref64 = run("oracle", sd_c, sd_f, rays, torch.float64)
    gap = (ref.double() - ref64).abs().amax(-1)
    lines = [f"fp8 cross-term emulation, {N} rays of config #2 (mid-frame), synthetic field make_state_dict(99 / 100), 64 + 64 samples",
             f"oracle fp32 vs fp64: max {gap.max():.3e}  p99.9 {torch.quantile(gap, 0.999):.3e}  median {gap.median():.3e}",
             "mode                     max        p99.9      p99        median     rays>1e-4  rays>max(1e-4, 2 x oracle gap)"]
    for mode in ("f16x3", "fp8_cross_lo_exact", "fp8_cross", "f16"):
        d = (run(mode, sd_c, sd_f, rays).double() - ref.double()).abs().amax(-1)
        viol = int((d > torch.maximum(torch.full_like(gap, 1e-4), 2 * gap)).sum())
        lines.append(f"{mode:<24} {d.max():.3e}  {torch.quantile(d, 0.999):.3e}  {torch.quantile(d, 0.99):.3e}  {d.median():.3e}  "
                     f"{int((d > 1e-4).sum()):9d}  {viol:9d}")
        print(lines[-1], flush=True)
    text = "\n".join(lines) + "\n"
    print(text)
    if OUT:
        open(OUT, "w").write(text)


if __name__ == "__main__":
    main()
