#!/usr/bin/env python3
"""Golden vectors of the reference's refinement network: ``models/networks.py::MaxPoolingModel`` in
eval mode, weights from ``nerf_sr_amd.refine.make_refine_state_dict`` loaded with ``load_state_dict``, on small
random patches.  Development container only; data only.

    python tests/golden/make_golden_refine.py      # rewrites tests/golden/refine.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from nerf_sr_amd.refine import make_refine_state_dict  # noqa: E402


def main():
    mg.install_shim()
    torch.set_grad_enabled(False)
    from models.networks import MaxPoolingModel
    net = MaxPoolingModel(types.SimpleNamespace(not_use_ref=False))
    sd = {k: torch.from_numpy(v) for k, v in make_refine_state_dict(7).items()}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    net.eval()
    gen = torch.Generator().manual_seed(5)
    out = {"seed": 7}
    ref_sd = net.state_dict()
    out["float_keys"] = np.array([k for k, v in ref_sd.items() if v.dtype.is_floating_point])
    out["float_shapes"] = np.array([str(tuple(v.shape)) for v in ref_sd.values() if v.dtype.is_floating_point])
    for tag, (B, R, H, W) in {"a": (2, 3, 16, 24), "b": (1, 8, 32, 32)}.items():
        x = torch.rand(B, 3, H, W, generator=gen) * 2 - 1
        c = torch.rand(B, R, 3, H, W, generator=gen) * 2 - 1
        y = net(x, c)
        feats = net.E(x)
        out[f"x_{tag}"], out[f"c_{tag}"], out[f"y_{tag}"] = mg.np32(x), mg.np32(c), mg.np32(y)
        out[f"f3_{tag}"] = mg.np32(feats[3])
        print(tag, tuple(y.shape), "|y| max", float(y.abs().max()))
    # --not_use_ref: the reference's own module with the no-pooling decoder (networks.py:866-945, 958-969)
    net = MaxPoolingModel(types.SimpleNamespace(not_use_ref=True))
    sd = {k: torch.from_numpy(v) for k, v in make_refine_state_dict(8, not_use_ref=True).items()}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    net.eval()
    out["noref_seed"] = 8
    out["noref_float_shapes"] = np.array([str(tuple(v.shape)) for v in net.state_dict().values() if v.dtype.is_floating_point])
    x = torch.rand(2, 3, 24, 16, generator=gen) * 2 - 1
    y = net(x, None)
    out["x_noref"], out["y_noref"] = mg.np32(x), mg.np32(y)
    print("noref", tuple(y.shape), "|y| max", float(y.abs().max()))
    path = os.path.join(HERE, "refine.npz")
    np.savez_compressed(path, **out)
    print("->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
