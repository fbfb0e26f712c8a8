#!/usr/bin/env python3
"""Golden vectors for ``--gamma_correct`` (models/nerf_downX_model.py:271-276): the reference's own ``forward`` and
``render_rays`` with ``opt.gamma_correct = True`` on the rays of ``path_llff.npz`` / ``path_blender.npz``.

Runs ONLY in the development container (imports ``/root/reference`` through the shim of ``make_golden.py``); writes
``tests/golden/gamma.npz`` (data only: inputs are the rays already held by the path fixtures, outputs the eight
``out_*`` tensors and sixteen per-sample colours).

    python tests/golden/make_golden_gamma.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.install_shim()
    torch.set_grad_enabled(False)
    import models.utils as ru
    out = {}
    for tag, white in (("llff", False), ("blender", True)):
        g = np.load(os.path.join(HERE, f"path_{tag}.npz"))
        model, opt = mg.build_reference_model(white, seed_c=int(g["seed_coarse"]), seed_f=int(g["seed_fine"]),
                                              img_wh=(32, 16), downscale=2)
        opt.gamma_correct = True
        rays = torch.from_numpy(g["rays"])[:64].contiguous()
        model.set_input({"rays": rays[None]})
        model.forward()
        for k in ("coarse_comp_rgbs", "coarse_depth", "coarse_opacity", "coarse_weights",
                  "fine_comp_rgbs", "fine_depth", "fine_opacity", "fine_weights"):
            out[f"{tag}_{k}"] = mg.np32(getattr(model, f"out_{k}"))
        o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
        z_c, xyz_c = ru.sample_along_rays(o, d, near, far, 64, False, False)
        rgbs, sig = model.render_rays(model.netCoarse, xyz_c, model.embeddings['dir'](d))
        out[f"{tag}_coarse_point_rgb"] = mg.np32(rgbs[:16])
        out[f"{tag}_coarse_point_sigma"] = mg.np32(sig[:16])
        out[f"{tag}_n_rays"] = 64
    np.savez_compressed(os.path.join(HERE, "gamma.npz"), **out)
    print("wrote gamma.npz", os.path.getsize(os.path.join(HERE, "gamma.npz")), "bytes")


if __name__ == "__main__":
    main()
