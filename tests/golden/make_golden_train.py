#!/usr/bin/env python3
"""Golden vectors of ONE TRAINING ITERATION of the reference.

Runs ONLY in the development container (needs ``/root/reference``).  Imports the reference through the
same ``sys.modules`` shim as ``make_golden.py``, builds ``NeRFDownXModel`` with ``TrainOptions`` (so the
reference's own Adam is constructed), feeds one small batch through ``set_input`` /
``optimize_parameters`` and records: the random draws the reference made (by wrapping ``torch.rand_like``,
``torch.rand``, ``torch.randn_like`` while its forward runs), the eight forward outputs, the LR means, the
losses, a digest (norm, sum, 512-element subsample) of every gradient tensor, and the same digest of the weights
after the Adam step.  Data only; nothing of the reference's source is copied.

    python tests/golden/make_golden_train.py      # rewrites tests/golden/train_*.npz
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (shim + helpers; also puts the repo on sys.path)

from nerf_sr_amd.weights import make_state_dict  # noqa: E402
from nerf_sr_amd import cameras  # noqa: E402

def sample_idx(numel: int, cap: int = 512):
    """Deterministic subsample of a flattened tensor: every element if it is small, else ``cap`` evenly
    spaced ones (odd stride so that rows and columns are both swept)."""
    if numel <= cap:
        return np.arange(numel)
    stride = (numel // cap) | 1
    return np.arange(0, numel, stride)[:cap]


def build_train_model(white_bkgd, seed_c, seed_f, downscale, randomized, noise_std, dataset_mode, extra_argv=()):
    from options.train_options import TrainOptions
    from models import create_model
    tmp = tempfile.mkdtemp(prefix="nsr_golden_train_")
    argv = ["x", "--name", "golden", "--checkpoints_dir", tmp, "--dataset_root", tmp,
            "--model", "nerf_downX", "--dataset_mode", dataset_mode, "--img_wh", "16", "12",
            "--downscale", str(downscale), "--N_coarse", "64", "--N_importance", "64"]
    if white_bkgd:
        argv.append("--white_bkgd")
    argv += list(extra_argv)
    old = sys.argv
    sys.argv = argv
    try:
        opt = TrainOptions().parse(None)
    finally:
        sys.argv = old
    opt.white_bkgd = white_bkgd
    opt.noise_std = noise_std
    opt.randomized = randomized
    model = create_model(opt)
    model.netCoarse.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(seed_c).items()})
    model.netFine.load_state_dict({k: torch.from_numpy(v) for k, v in make_state_dict(seed_f).items()})
    model.train()
    return model, opt


class RecordDraws:
    """Record the tensors returned by the three RNG entry points the reference's train-mode forward uses."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self.orig = (torch.rand_like, torch.rand, torch.randn_like)
        rec = self.draws

        def wrap(fn, tag):
            def f(*a, **k):
                t = fn(*a, **k)
                rec.append((tag, t.detach().clone()))
                return t
            return f
        torch.rand_like = wrap(self.orig[0], "rand_like")
        torch.rand = wrap(self.orig[1], "rand")
        torch.randn_like = wrap(self.orig[2], "randn_like")
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand, torch.randn_like = self.orig


def one_case(tag, white, ndc, near_far, s, n_lr, randomized, noise_std, seed, extra_argv=()):
    torch.manual_seed(seed)
    model, opt = build_train_model(white, 99, 100, s, randomized, noise_std, "llff_downX" if ndc else "blender_downX", extra_argv)
    import models.utils as ru
    import einops
    # rays of a small HR grid, regrouped LR-pixel-major exactly as the datasets do
    H, W = 12 * s // 2, 16 * s // 2
    focal = cameras.llff_focal(W) if ndc else cameras.blender_focal(W)
    c2w = torch.from_numpy(cameras.spiral_pose(0.7) if ndc else cameras.spheric_pose(35.0, -25.0, 4.0)).float()
    dirs = ru.get_ray_directions(H, W, focal)
    o, d = ru.get_rays(dirs, c2w)
    if ndc:
        o, d = ru.get_ndc_rays(H, W, focal, 1.0, o, d)
    near = near_far[0] * torch.ones_like(o[:, :1])
    far = near_far[1] * torch.ones_like(o[:, :1])
    rays = torch.cat([o, d, near, far], 1).view(H, W, 8)
    rays = einops.rearrange(rays, "(h s1) (w s2) c -> (h w) (s1 s2) c", s1=s, s2=s)
    sel = torch.randperm(rays.shape[0])[:n_lr]
    rays = rays[sel].contiguous()                                  # (n_lr, s^2, 8): a batch of LR pixels
    target = torch.rand(n_lr, 3)
    w0_c = {k: v.detach().clone() for k, v in model.netCoarse.state_dict().items()}
    model.set_input({"rays": rays.clone(), "rgbs": target.clone()})
    if getattr(opt, "use_depth_var_loss", False):
        # environment shim, not a change of arithmetic: forward_rays stores self.far as a 0-d NUMPY array (:284) and :351 divides
        # a tensor that requires grad by it -- fine on the reference's pinned torch 1.8.1, a RuntimeError on torch 2.x
        # ("Can't call numpy() on Tensor that requires grad").  Hand the same value over as a Python float.
        inner = model.comp_low_res_output

        def comp_low_res_output_with_float_far():
            model.far = float(model.far)
            return inner()
        model.comp_low_res_output = comp_low_res_output_with_float_far
    with RecordDraws() as rec:
        model.optimize_parameters()
    draws = rec.draws
    out = {"rays": mg.np32(rays.view(-1, 8)), "target_lr": mg.np32(target), "s2": s * s, "white_bkgd": white,
           "randomized": randomized, "noise_std": noise_std, "seed_coarse": 99, "seed_fine": 100,
           "lr": opt.lr, "beta1": opt.beta1, "lambda_coarse": opt.lambda_coarse_mse, "lambda_fine": opt.lambda_fine_mse}
    if randomized:
        kinds = [t for t, _ in draws]
        want = ["rand_like"] + (["randn_like"] if noise_std > 0 else []) + ["rand"] + (["randn_like"] if noise_std > 0 else [])
        assert kinds == want, kinds
        it = iter(draws)
        out["u_coarse"] = mg.np32(next(it)[1])
        if noise_std > 0:
            out["noise_coarse"] = mg.np32(next(it)[1])
        out["u_fine"] = mg.np32(next(it)[1])
        if noise_std > 0:
            out["noise_fine"] = mg.np32(next(it)[1])
    else:
        assert not draws
    # forward outputs as the reference holds them after comp_low_res_output()
    out["lr_coarse"] = mg.np32(model.out_coarse_comp_rgbs)
    out["lr_fine"] = mg.np32(model.out_fine_comp_rgbs)
    out["hr_coarse"] = mg.np32(model.out_coarse_comp_rgbs_ori)
    out["hr_fine"] = mg.np32(model.out_fine_comp_rgbs_ori)
    out["fine_weights"] = mg.np32(model.out_fine_weights)
    out["loss_coarse_mse"] = float(model.loss_coarse_mse)
    out["loss_fine_mse"] = float(model.loss_fine_mse)
    out["loss_tot"] = float(model.loss_tot)
    if opt.use_var_loss or opt.use_depth_var_loss:      # the optional variance losses (nerf_downX_model.py:332-336, 349-353, 374-378)
        out["lambda_var"] = np.array([opt.lambda_coarse_var if opt.use_var_loss else 0.0, opt.lambda_fine_var if opt.use_var_loss else 0.0,
                                      opt.lambda_coarse_depth_var if opt.use_depth_var_loss else 0.0,
                                      opt.lambda_fine_depth_var if opt.use_depth_var_loss else 0.0], np.float64)
        out["var_losses_raw"] = np.array([float(model.loss_out_coarse_var) if opt.use_var_loss else 0.0,
                                          float(model.loss_out_fine_var) if opt.use_var_loss else 0.0,
                                          float(model.loss_coarse_depth_var) if opt.use_depth_var_loss else 0.0,
                                          float(model.loss_fine_depth_var) if opt.use_depth_var_loss else 0.0], np.float64)
        out["far"] = float(model.far)
    for net, name in ((model.netCoarse, "coarse"), (model.netFine, "fine")):
        mod = net.module if hasattr(net, "module") else net
        for k, p in mod.named_parameters():
            # --stop_grad leaves xyz_encoding_final without a gradient (None: torch.optim.Adam skips the tensor); stored as zeros
            g = p.grad.detach() if p.grad is not None else torch.zeros_like(p)
            out[f"gnorm_{name}.{k}"] = float(g.double().norm())
            out[f"gsum_{name}.{k}"] = float(g.double().sum())
            out[f"grad_{name}.{k}"] = mg.np32(g).reshape(-1)[sample_idx(g.numel())]
    # weights after the reference's Adam step (coarse net: a few tensors in full, all as digests)
    for k, v in model.netCoarse.state_dict().items():
        dlt = (v.detach() - w0_c[k]).double()
        out[f"dw_norm_coarse.{k}"] = float(dlt.norm())
        out[f"w1_coarse.{k}"] = mg.np32(v).reshape(-1)[sample_idx(v.numel())]
    path = os.path.join(HERE, f"train_{tag}.npz")
    np.savez_compressed(path, **out)
    print(tag, "loss", out["loss_tot"], "->", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def option_cases():
    # --gamma_correct in training (render_rays, nerf_downX_model.py:271-276); --sigma_activation softplus
    # (models/rendering.py:69-73); --color_activation none (models/networks.py:173-180)
    one_case("llff_gamma", False, True, (0.0, 1.0), 2, 24, True, 1.0, 5, ("--gamma_correct",))
    one_case("blender_softplus", True, False, (2.0, 6.0), 2, 24, True, 0.0, 6, ("--sigma_activation", "softplus"))
    one_case("llff_colornone", False, True, (0.0, 1.0), 2, 24, True, 1.0, 7, ("--color_activation", "none"))
    # --stop_grad true (models/networks.py:127, 218-219)
    one_case("blender_stopgrad", True, False, (2.0, 6.0), 2, 24, True, 0.0, 8, ("--stop_grad", "true"))


def main():
    mg.install_shim()
    torch.set_grad_enabled(True)
    if len(sys.argv) > 1 and sys.argv[1] == "options":      # only the round-5 additions (the others regenerate bit for bit)
        option_cases()
        return
    one_case("llff_det", False, True, (0.0, 1.0), 2, 24, False, 0.0, 1)
    one_case("llff_rand", False, True, (0.0, 1.0), 2, 24, True, 1.0, 2)
    one_case("blender_rand", True, False, (2.0, 6.0), 2, 24, True, 0.0, 3)
    # --use_var_loss --use_depth_var_loss at weights that make the terms matter next to the MSEs (the defaults, 0.01, on a
    # 24-pixel batch are 1e-3 of the gradient)
    one_case("blender_var", True, False, (2.0, 6.0), 2, 24, True, 0.0, 4,
             ("--use_var_loss", "--lambda_coarse_var", "0.05", "--lambda_fine_var", "0.08",
              "--use_depth_var_loss", "--lambda_coarse_depth_var", "0.3", "--lambda_fine_depth_var", "0.2"))
    option_cases()


if __name__ == "__main__":
    main()
