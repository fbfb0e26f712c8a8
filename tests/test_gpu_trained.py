"""Frame-scale parity of the contract-grade precisions on TRAINED density fields (``pytest -m gpu``).

The frame-scale tests of ``test_gpu_frames.py`` use the synthetic "smooth" field of ``nerf_sr_amd.weights`` -- a random
network made well conditioned on purpose.  A trained NeRF is not that: it has surfaces, empty space and grown weights.
No checkpoint can be downloaded, so a module-scoped fixture TRAINS both networks here, with this repository's own HIP
training step, on analytic hard-surface scenes whose pixel colours are exact by ray intersection
(``tests/trained_field.py``: opaque spheres in front of a textured wall in NDC space for the forward-facing family,
against a white background in world space for the Blender family; 4,000 steps of 2,048 rays from 12 poses, ~25 s per
family).  Then the protocol of ``test_gpu_frames.py`` runs on those weights, from a pose that was not trained on:
16,384 consecutive rays from the middle of config #2's / config #3's frame, f16x3 (the dtype the bench is quoted on) and
fp32, against the fp32 oracle and its fp64 evaluation.

The contract, per ray: |dRGB| of the fine colours <= max(1e-4, 2 x that ray's oracle fp32-vs-fp64 gap).  The second term
is the conditioning of the reference algorithm itself (models/utils.py:61-95: the inverse CDF amplifies the fp32 rounding
of the coarse weights, and ``denom < 1e-5 -> 1`` at :87-88 flips under it): two fp32 evaluations of such a ray can each
sit `gap` away from the exact result on opposite sides, so no second fp32 implementation -- the reference itself on
another BLAS included -- can be held closer than that.  The number of rays whose bound is the second term ("exempt") is
printed and bounded (<= 1.5 % of the block: 16-18 / 101 of 16,384 measured); violations were ZERO in every measured run
(at most two are tolerated, and only on rays the oracle itself finds ill-conditioned, see the test body); and the error
distribution must sit inside the oracle's own fp32-vs-fp64 envelope.

Also here: trained-scale activations against the split-fp16 operand range (|h| < 1023.75, include/nsr.h
NSR_FLAG_ACTIVATION_RANGE) -- per-layer maxima in fp64 on a sample of the fine-pass points, and the device's own status word
over every point of the block.
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as oc
from tests import trained_field as tf

pytestmark = pytest.mark.gpu

N_RAYS = 16384
STEPS = 4000
MAX_EXEMPT_FRACTION = 0.015
MAX_MARGINAL = 2
REPORT = {}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from nerf_sr_amd import ops as _ops   # raises if libnsr.so is missing: no fallback
    return _ops


@pytest.fixture(scope="module", params=["llff", "blender"])
def trained(request, ops):
    """(family, coarse state dict, fine state dict, oracle block): trained once per family, shared by both precisions."""
    family = request.param
    t0 = time.time()
    res = tf.train_field(family, steps=STEPS, log=print)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    final_psnr = -10.0 * np.log10(max(res["history"][-1][2], 1e-12))
    print(f"[trained {family}] {STEPS} steps in {t_train:.1f} s, final fine PSNR on the training rays {final_psnr:.1f} dB")
    assert res["history"][-1][2] < 0.25 * res["history"][0][2], "training did not reduce the loss"
    t0 = time.time()
    blk, ref, ref64 = tf.oracle_block(family, res["sd_coarse"], res["sd_fine"], N_RAYS)
    print(f"[trained {family}] oracle fp32 + fp64 on {N_RAYS} rays: {time.time() - t0:.1f} s")
    REPORT[family] = {"train_steps": STEPS, "train_seconds": round(t_train, 1), "history": res["history"]}
    return family, res["sd_coarse"], res["sd_fine"], blk, ref, ref64


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_trained_field_frame_scale_parity(ops, trained, prec):
    family, sd_c, sd_f, blk, ref, ref64 = trained
    white = tf.FAMILIES[family][3]
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(sd_c)      # checked packing: trained weights are in range
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
    hip = ops.forward_rays(net_c, net_f, blk.cuda(), 64, 64, white, check=True)     # raises on any numerics flag
    st = tf.parity_stats(hip, ref, ref64, cross_feed=(sd_f, blk, white))
    REPORT[family][prec] = st
    print(f"[trained {family} {prec}] " + json.dumps(st))
    # the scene has surfaces: the oracle's own fp32-vs-fp64 gap is orders of magnitude above the smooth field's (< 1e-5)
    assert st["oracle64_vs_oracle32"]["max"] > 1e-4
    # coarse colours have no resampling in front of them: fp32-grade agreement (the density head of a trained network is
    # large -- |sigma| reaches 300-400 -- so the yardstick is the oracle's own fp32-vs-fp64 distance, 3-5e-6 here)
    assert st["coarse_max"] <= max(1e-5, 3.0 * st["coarse_oracle_gap_max"])
    # the contract, on every ray.  The factor 2 is the honest-noise argument of the module docstring; the HIP evaluation's own
    # rounding is a second, independent draw of that noise, so on a ray the oracle itself finds ill-conditioned (gap >=
    # 2.5e-5) a draw up to 4 x gap is tolerated -- on at most MAX_MARGINAL rays, printed.  Anything else fails.
    # Round 6: "cannot explain" is now CHECKED instead of inferred from the gap (one draw of rounding noise, which can miss the
    # resampler's amplification on a ray: DESIGN 4).  A ray over its bound must be marginal in the sense above, or the
    # oracle's own fp32 fine pass on the HIP coarse weights must reproduce the HIP colour (tests/util.py::
    # explained_by_resampler_conditioning).  The networks are re-trained inside this test by a chaotic 4,000-step run, so the
    # block changes with every rounding-level change of the training kernels: round 6's sources produced one such ray per
    # re-training (forward-facing family; two-term chain: d 1.67e-4 at gap 3.2e-8; mixed chain: d 1.9 / 2.0e-4 at gap 4.3e-5, i.e.
    # 4.3 - 4.7 x gap, a hair outside the marginal allowance; both reproduced to 1.2e-7 by the oracle's fp32 fine pass on the HIP
    # coarse weights, which sit 2.3e-6 from the fp32 oracle's where the fp32 and fp64 oracles sit 3.8e-6 apart:
    # profiles/r6_trained_ray_probe_*.json), round 5's none.
    assert st["unexplained_violations"] == 0, f"rays outside the contract that conditioning does not explain: {st['rays_over_bound']}"
    assert st["violations"] <= MAX_MARGINAL, f"{st['violations']} rays outside max(1e-4, 2 x oracle gap): {st['worst_rays']}"
    assert st["exempt_rays"] <= MAX_EXEMPT_FRACTION * N_RAYS
    # inside the oracle's own envelope: no statistic of HIP-vs-oracle32 worse than 2 x the oracle's fp64-vs-fp32
    h, g = st["hip_vs_oracle32"], st["oracle64_vs_oracle32"]
    assert h["median"] <= 2 * g["median"] + 1e-7 and h["p99"] <= 2 * g["p99"] + 1e-6 and h["p999"] <= 2 * g["p999"] + 1e-5
    assert h["over_1e-4"] <= 2 * g["over_1e-4"] + 2 and h["max"] <= max(2 * g["max"], 1e-4)
    # PSNR of the s^2-mean image against a common target moves by < 1e-3 dB
    s2 = tf.FAMILIES[family][1] ** 2
    lr = ops.sr_mean(hip["fine_comp_rgbs"], N_RAYS // s2, s2).cpu()
    lr_ref = oc.sr_mean(ref["fine_comp_rgbs"], N_RAYS // s2, s2)
    tgt = oc.sr_mean(tf.analytic_colours(blk, family), N_RAYS // s2, s2)        # the scene itself
    assert abs(oc.psnr(lr, tgt) - oc.psnr(lr_ref, tgt)) <= 1e-3
    print(f"[trained {family} {prec}] PSNR of the rendered block vs the analytic scene: {oc.psnr(lr_ref, tgt):.2f} dB (oracle), "
          f"{oc.psnr(lr, tgt):.2f} dB (HIP)")
    assert oc.psnr(lr_ref, tgt) > 18.0                                           # the field did learn the scene


def test_trained_activations_leave_headroom_in_the_split_fp16_range(ops, trained):
    """max |pre-activation| per layer on the fine-pass points of every 8th ray (fp64, CPU): the split-fp16 path carries
    |h| < 1023.75 at full precision; trained weights must sit well inside, and the device's own range tracking (status
    word, every point of the block) must agree."""
    family, sd_c, sd_f, blk, ref, _ = trained
    o, d, near, far = blk[::8, 0:3], blk[::8, 3:6], blk[::8, 6:7], blk[::8, 7:8]
    z_c, xyz_c = oc.sample_coarse(o, d, near, far, 64)
    z_f, xyz_f = oc.resample_fine(o, d, z_c, ref["coarse_weights"][::8], 64)
    de = oc.posenc(d, 4)
    worst = 0.0
    for name, sd, xyz in (("coarse", sd_c, xyz_c), ("fine", sd_f, xyz_f)):
        x = torch.cat([oc.posenc(xyz.reshape(-1, 3), 10), de.repeat_interleave(xyz.shape[1], 0)], -1)
        m = tf.layer_abs_max(sd, x)
        REPORT[family][f"layer_abs_max_{name}"] = m
        hidden = max(v for k, v in m.items() if k.startswith("xyz_encoding") or k == "dir_encoding")
        worst = max(worst, hidden)
        print(f"[trained {family}] {name} network: max |pre-activation| per layer " + json.dumps({k: round(v, 1) for k, v in m.items()}))
        assert m["max_weight"] < 1023.75 / 16
    assert worst < 1023.75 / 2, f"trained activations reach {worst:.0f}: less than 2x headroom under the split-fp16 range"
    net_c = ops.VanillaMLP(precision="f16x3").load_state_dict(sd_c)
    net_f = ops.VanillaMLP(precision="f16x3").load_state_dict(sd_f)
    ops.forward_rays(net_c, net_f, blk.cuda(), 64, 64, tf.FAMILIES[family][3])
    assert net_c.status() == 0 and net_f.status() == 0
    # ... and the tracking is live on these very weights: the same networks with the first trunk layers scaled trip it
    big = {k: (v * np.float32(16.0) if k in ("xyz_encoding_1.0.weight", "xyz_encoding_2.0.weight") else v) for k, v in sd_f.items()}
    net_b = ops.VanillaMLP(precision="f16x3").load_state_dict(big)
    ops.forward_rays(net_c, net_b, blk[:1024].cuda(), 64, 64, tf.FAMILIES[family][3])
    assert net_b.status() & 4
    out = os.environ.get("NSR_PARITY_REPORT")
    if out:                                         # scripts/gpu_*.sh: keep the numbers (profiles/r3_parity_trained.json)
        with open(out, "w") as f:
            json.dump(REPORT, f, indent=1)
