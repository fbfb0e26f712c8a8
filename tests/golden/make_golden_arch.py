#!/usr/bin/env python3
"""Golden vectors for the ARCHITECTURE flags of the path (models/networks.py:124-128 ``--D --W --skips``,
models/nerf_model.py:53-57 ``--deg_pos --deg_dir``): the reference's own ``forward`` and network calls for two
non-default networks on the rays of ``path_llff.npz`` / ``path_blender.npz``.

Runs ONLY in the development container (imports ``/root/reference`` through the shim of ``make_golden.py``); writes
``tests/golden/arch.npz`` (data only).  Weights are not stored: ``nerf_sr_amd.weights.make_state_dict_arch(seed, **arch)``.

    python tests/golden/make_golden_arch.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

# name -> (architecture, rays fixture, white background)
CASES = {
    "small": ({"D": 4, "W": 128, "skips": (2,), "deg_pos": 6, "deg_dir": 2}, "llff", False),
    "odd": ({"D": 6, "W": 192, "skips": (1, 3), "deg_pos": 10, "deg_dir": 4}, "blender", True),   # two skips, width not 2^k
}
N_RAYS = 64


def build(white, arch, seed_c, seed_f):
    from options.test_options import TestOptions
    from models import create_model
    from nerf_sr_amd.weights import make_state_dict_arch
    tmp = tempfile.mkdtemp(prefix="nsr_golden_")
    argv = ["x", "--name", "golden", "--checkpoints_dir", tmp, "--dataset_root", tmp, "--model", "nerf_downX",
            "--dataset_mode", "llff_downX", "--img_wh", "32", "16", "--downscale", "2", "--N_coarse", "64", "--N_importance", "64",
            "--D", str(arch["D"]), "--W", str(arch["W"]), "--skips", *[str(s) for s in arch["skips"]],
            "--deg_pos", str(arch["deg_pos"]), "--deg_dir", str(arch["deg_dir"])] + (["--white_bkgd"] if white else [])
    old, sys.argv = sys.argv, argv
    try:
        opt = TestOptions().parse(None)
    finally:
        sys.argv = old
    opt.white_bkgd = white
    opt.noise_std = 0.0
    model = create_model(opt)
    for net, seed in ((model.netCoarse, seed_c), (model.netFine, seed_f)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict_arch(seed, **arch).items()})
    model.eval()
    return model, opt


def main():
    mg.install_shim()
    torch.set_grad_enabled(False)
    import models.utils as ru
    out = {"n_rays": N_RAYS, "seed_coarse": 21, "seed_fine": 22}
    for case, (arch, tag, white) in CASES.items():
        g = np.load(os.path.join(HERE, f"path_{tag}.npz"))
        model, opt = build(white, arch, 21, 22)
        rays = torch.from_numpy(g["rays"])[:N_RAYS].contiguous()
        model.set_input({"rays": rays[None]})
        model.forward()
        for k in ("coarse_comp_rgbs", "coarse_depth", "coarse_opacity", "coarse_weights",
                  "fine_comp_rgbs", "fine_depth", "fine_opacity", "fine_weights"):
            out[f"{case}_{k}"] = mg.np32(getattr(model, f"out_{k}"))
        o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
        z_c, xyz_c = ru.sample_along_rays(o, d, near, far, 64, False, False)
        x = torch.cat([model.embeddings['pos'](xyz_c.view(-1, 3)), model.embeddings['dir'](d).repeat_interleave(64, dim=0)], -1)
        out[f"{case}_mlp_in_256"] = mg.np32(x[:256])
        out[f"{case}_mlp_out_256"] = mg.np32(model.netCoarse(x[:256]))
        out[f"{case}_mlp_sigma_only_64"] = mg.np32(model.netCoarse(x[:64], sigma_only=True))
    np.savez_compressed(os.path.join(HERE, "arch.npz"), **out)
    print("wrote arch.npz", os.path.getsize(os.path.join(HERE, "arch.npz")), "bytes")


if __name__ == "__main__":
    main()
