"""CPU checks of the parity RULES themselves (no GPU): the cross-feed rule of round 6 that decides whether a ray over the
per-ray contract is explained by the conditioning of the reference's inverse-CDF resampler
(tests/util.py::explained_by_resampler_conditioning, used by tests/test_gpu_frames.py and scripts/parity_record.py), exercised
with the oracle standing in for the device path."""
import os

import numpy as np
import torch

from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc
from tests.util import explained_by_resampler_conditioning, oracle_forward_parallel


def test_cross_feed_rule_accepts_rounding_level_coarse_differences_and_rejects_everything_else(golden_dir):
    g = np.load(os.path.join(golden_dir, "path_blender.npz"))
    rays = torch.from_numpy(g["rays"])[:96]
    sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    ref = oracle_forward_parallel(sd_c, sd_f, rays, True)
    # a second "implementation": the same arithmetic behind coarse weights that differ at rounding level (3e-7 relative)
    gen = torch.Generator().manual_seed(0)
    w2 = ref["coarse_weights"] * (1.0 + 3e-7 * torch.randn(ref["coarse_weights"].shape, generator=gen))
    sf = oc.to_torch_sd(sd_f)
    with torch.no_grad():
        z, _ = oc.sample_coarse(rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8], 64)
        zf, xyzf = oc.resample_fine(rays[:, 0:3], rays[:, 3:6], z, w2, 64)
        rgb, sig = oc.render_points(sf, xyzf, oc.posenc(rays[:, 3:6], 4))
        fine2 = oc.composite(rgb, sig, zf, True)[0]
    impl = {"coarse_weights": w2, "fine_comp_rgbs": fine2}
    idx = torch.arange(rays.shape[0])
    assert bool(explained_by_resampler_conditioning(sd_f, rays, True, impl, ref, idx).all())
    # ... and what the rule must NOT let through: (a) a colour the reference's fine pass does not reproduce from those weights
    bad_rgb = {"coarse_weights": w2, "fine_comp_rgbs": fine2 + 1e-3}
    assert not bool(explained_by_resampler_conditioning(sd_f, rays, True, bad_rgb, ref, idx).any())
    # (b) coarse weights that are off by more than rounding (1e-4): then the coarse network is what differs, not the resampler
    w3 = ref["coarse_weights"] + 1e-4
    with torch.no_grad():
        zf3, xyzf3 = oc.resample_fine(rays[:, 0:3], rays[:, 3:6], z, w3, 64)
        rgb3, sig3 = oc.render_points(sf, xyzf3, oc.posenc(rays[:, 3:6], 4))
        fine3 = oc.composite(rgb3, sig3, zf3, True)[0]
    bad_w = {"coarse_weights": w3, "fine_comp_rgbs": fine3}
    assert not bool(explained_by_resampler_conditioning(sd_f, rays, True, bad_w, ref, idx).any())
    assert explained_by_resampler_conditioning(sd_f, rays, True, impl, ref, torch.zeros(0, dtype=torch.long)).numel() == 0
    # the coarse-weight yardstick with the oracle's fp64 evaluation at hand: twice the oracle's OWN fp32-vs-fp64 distance on
    # the ray (never below w_tol).  Weights 5e-6 away are rejected on this smooth field (own distance ~3e-7) ...
    ref64 = oracle_forward_parallel(sd_c, sd_f, rays, True, dtype=torch.float64)
    own = (ref["coarse_weights"].double() - ref64["coarse_weights"]).abs().max(-1)[0]
    assert float(own.max()) < 1e-6
    w4 = ref["coarse_weights"] + 5e-6
    with torch.no_grad():
        zf4, xyzf4 = oc.resample_fine(rays[:, 0:3], rays[:, 3:6], z, w4, 64)
        rgb4, sig4 = oc.render_points(sf, xyzf4, oc.posenc(rays[:, 3:6], 4))
        fine4 = oc.composite(rgb4, sig4, zf4, True)[0]
    off = {"coarse_weights": w4, "fine_comp_rgbs": fine4}
    assert not bool(explained_by_resampler_conditioning(sd_f, rays, True, off, ref, idx, ref64=ref64).any())
    # ... and accepted where the oracle's own two precisions sit further apart than that (a stand-in fp64 record 4e-6 away)
    far64 = {k: v.clone() for k, v in ref64.items()}
    far64["coarse_weights"] = ref["coarse_weights"].double() + 4e-6
    assert bool(explained_by_resampler_conditioning(sd_f, rays, True, off, ref, idx, ref64=far64).all())
    assert bool(explained_by_resampler_conditioning(sd_f, rays, True, impl, ref, idx, ref64=ref64).all())
