"""Host-side mirror of ``NeRFDownXModel`` for the render path only.

Reproduces the call protocol the reference's loops use
(``set_input -> forward -> out_* -> comp_low_res_output``; train.py:79-80,
test.py:52, models/nerf_downX_model.py:235-353,410-416,621-669) on top of the HIP
path.  Losses, optimisers, checkpoints, visualisers and the GAN / refinement
branches are out of scope.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import ops
from .ops import VanillaMLP, GenericMLP, VolumetricRenderer, PositionalEncoding


def default_options(**kw) -> SimpleNamespace:
    """The option values that define the path (SURVEY §5 'Config / flags')."""
    opt = SimpleNamespace(
        N_coarse=64, N_importance=64, lindisp=False, white_bkgd=False, randomized=True, noise_std=0.0,
        deg_pos=10, deg_dir=4, dim_pos=3, dim_dir=3, dim_rgb=3, downscale=2, img_wh=(504, 378),
        D=8, W=256, skips=[4],                      # models/networks.py:124-128; other values run layer by layer (ops.GenericMLP)
        no_dir=False, color_activation="sigmoid",   # :128, :160-180 ('none'; True): options of VanillaMLP
        sigma_activation="relu", gamma_correct=False,   # models/rendering.py:69-73 ('softplus'); nerf_downX_model.py:271-276
        ray_chunk=4096, point_chunk=262144, precision="fp32",
        check_numerics=True,    # forward() raises on NaN / out-of-range values (the reference: pdb, nerf_downX_model.py:273-274)
    )
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


class NeRFDownXModel:
    """``model.set_input(data); model.forward(); model.comp_low_res_output()`` as in the
    reference; the ``out_*`` attributes carry the same tensors (same shapes, fp32)."""

    def __init__(self, opt: Optional[SimpleNamespace] = None, device="cuda"):
        self.opt = opt or default_options()
        ops.check_mlp_options(self.opt, fused=False)
        if int(self.opt.N_coarse) < 2 or int(self.opt.N_importance) < 0:
            raise ValueError("N_coarse must be >= 2 and N_importance >= 0")
        self.renderer = VolumetricRenderer(self.opt)      # validates sigma_activation before any device is touched
        self.device = torch.device(device)
        # the fused kernels for the architecture every script of the reference uses, nn.Linear by nn.Linear for any other
        self.netCoarse = ops.make_mlp(self.opt, precision=self.opt.precision, device=self.device)
        self.netFine = ops.make_mlp(self.opt, precision=self.opt.precision, device=self.device)
        self.fused = isinstance(self.netCoarse, VanillaMLP)
        self.models = {"coarse": self.netCoarse, "fine": self.netFine}
        self.embeddings = {"pos": PositionalEncoding(3, self.opt.deg_pos), "dir": PositionalEncoding(3, self.opt.deg_dir)}
        self.randomized = False
        self._ws = None
        self._outs: Dict[str, torch.Tensor] = {}

    # -- weights ---------------------------------------------------------------
    def load_networks(self, sd_coarse, sd_fine):
        """Takes the two 24-key state_dicts the reference stores as
        ``{epoch}_net_Coarse.pth`` / ``{epoch}_net_Fine.pth`` (models/base_model.py:181-219)."""
        self.netCoarse.load_state_dict(sd_coarse)
        self.netFine.load_state_dict(sd_fine)
        return self        # (the colour-head options of `opt` -- gamma_correct, color_activation -- are applied by VanillaMLP)

    # -- mode toggles (nerf_downX_model.py:250-258) ------------------------------
    def train(self):
        self.randomized = bool(self.opt.randomized)
        return self

    def eval(self):
        self.randomized = False
        return self

    # -- D1 ------------------------------------------------------------------------
    def set_input(self, input: Dict[str, torch.Tensor]):
        for name, v in input.items():
            v = v.squeeze(0) if (v.ndim > 0 and v.shape[0] == 1) else v
            v = v.reshape(-1, v.shape[-1]) if v.ndim in (3, 4) else v
            setattr(self, f"data_{name}", v.to(self.device))

    # -- D2 ------------------------------------------------------------------------
    def render_rays(self, model: VanillaMLP, xyz: torch.Tensor, dir_embedded: torch.Tensor, **kwargs):
        """Reference signature (xyz (R,N,3), dir_embedded (R,27)) -> (rgbs, sigmas): the
        unfused route through explicit positional encoding and ``VanillaMLP.forward``."""
        R, N = xyz.shape[:2]
        x = torch.cat([self.embeddings["pos"](xyz.reshape(-1, 3)), dir_embedded.repeat_interleave(N, dim=0)], -1)
        out = model(x, **kwargs).view(R, N, -1)
        rgb = out[..., :3]
        if isinstance(model, GenericMLP) and model.gamma_correct and not kwargs.get("sigma_only", False):
            # models/nerf_downX_model.py:271-276: pow(rgb, 1 / 2.2) after the network (VanillaMLP's fused colour head has
            # done it already: nsr_weights_set_gamma); the NaN trap that follows it there is the network's status word here
            rgb = torch.pow(rgb, 1.0 / 2.2)
        return rgb, out[..., 3]

    # -- D3 ------------------------------------------------------------------------
    def forward_rays(self, rays: torch.Tensor) -> Dict[str, torch.Tensor]:
        opt = self.opt
        if not self.randomized and self.fused:
            self._outs = ops.forward_rays(self.netCoarse, self.netFine if opt.N_importance > 0 else None, rays,
                                          opt.N_coarse, opt.N_importance, opt.white_bkgd, opt.lindisp,
                                          check=bool(getattr(opt, "check_numerics", True)),
                                          sigma_activation=self.renderer.sigma_activation)
            return self._outs
        # randomized (training-mode) forward, and every forward of a GenericMLP pair: same kernels, stage by stage, jitter
        # drawn with torch.rand
        o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
        rnd = self.randomized
        z, xyz = ops.sample_along_rays(o, d, near, far, opt.N_coarse, rnd, opt.lindisp)
        if not self.fused:
            dir_emb = self.embeddings["dir"]((rays[:, 8:11] if rays.shape[1] == 11 else d).contiguous())
        rgb, sig = ops.render_rays(self.netCoarse, rays, z) if self.fused else self.render_rays(self.netCoarse, xyz, dir_emb)
        rgb, sig = rgb.contiguous(), sig.contiguous()
        if rnd and opt.noise_std > 0:          # add_gaussian_noise (models/utils.py:199-212): only when randomized
            sig = sig + torch.randn_like(sig) * opt.noise_std
        c = self.renderer(rgb, sig, z, opt.white_bkgd)
        out = dict(zip(ops.OUT_KEYS[:4], c))
        if opt.N_importance > 0:
            z2, xyz2 = ops.resample_along_rays(o, d, z, c[3], opt.N_importance, rnd)
            rgb2, sig2 = ops.render_rays(self.netFine, rays, z2) if self.fused else self.render_rays(self.netFine, xyz2, dir_emb)
            rgb2, sig2 = rgb2.contiguous(), sig2.contiguous()
            if rnd and opt.noise_std > 0:
                sig2 = sig2 + torch.randn_like(sig2) * opt.noise_std
            out.update(zip(ops.OUT_KEYS[4:], self.renderer(rgb2, sig2, z2, opt.white_bkgd)))
        if getattr(opt, "check_numerics", True):
            self.netCoarse.check("coarse network")
            if opt.N_importance > 0:
                self.netFine.check("fine network")
        return out

    def forward(self):
        out = self.forward_rays(self.data_rays)
        for name, v in out.items():
            setattr(self, f"out_{name}", v)

    # -- A1 / A2 ---------------------------------------------------------------------
    def comp_low_res_output(self):
        s2 = self.opt.downscale ** 2
        n_lr = self.data_rgbs.shape[0] if hasattr(self, "data_rgbs") else self.out_coarse_comp_rgbs.shape[0] // s2
        for name in ("coarse_comp_rgbs", "coarse_depth", "fine_comp_rgbs", "fine_depth"):
            if hasattr(self, f"out_{name}"):
                hr = getattr(self, f"out_{name}")
                setattr(self, f"out_{name}_ori", hr)
                setattr(self, f"out_{name}", ops.sr_mean(hr, n_lr, s2))

    def unflatten_reshape(self, input: torch.Tensor) -> torch.Tensor:
        return ops.unflatten_reshape(input, self.opt.img_wh, self.opt.downscale)

    # -- full image (test loop body, nerf_downX_model.py:621-669 without the savers) -----
    @torch.no_grad()
    def render_image(self, c2w, focal: float, ndc: bool, near: float = 0.0, far: float = 1.0):
        """Rays generated on the device -> forward -> LR means + HR image of one pose."""
        rays = ops.subpixel_rays(c2w, self.opt.img_wh, focal, self.opt.downscale, ndc, near, far, self.device)
        self.set_input({"rays": rays})
        self.forward()
        hr = self.unflatten_reshape(self.out_fine_comp_rgbs)
        self.comp_low_res_output()
        return {"hr_rgb": hr, "lr_rgb": self.out_fine_comp_rgbs, "lr_depth": self.out_fine_depth}

    @torch.no_grad()
    def render_image_sharded(self, c2w, focal: float, ndc: bool, near: float = 0.0, far: float = 1.0, group=None,
                             lr_range=None, workspace=None, outs=None, gather: str = "lr", want_weights: bool = False):
        """One frame rendered by all ranks of ``group`` together (BASELINE config #4): the LR-pixel range
        is cut into contiguous blocks (``dist.shard_bounds``; an LR pixel's s*s sub-rays stay on one GPU), every rank
        GENERATES its own ray block on its device (nothing is scattered), runs the eval-mode ``forward_rays`` on it, and
        ONE all-gather assembles the frame on every rank.  Replaces the per-MLP-call scatter / gather of
        nn.DataParallel (models/networks.py:54-69).  No reduction crosses a block boundary, so the result is
        bit-identical to ``render_image`` on one GPU.

        ``gather``: what the collective carries.
          * ``"lr"`` (default): the s*s means, [r, g, b, depth] per LR pixel (16 B / LR pixel) -> ``lr_rgb``, ``lr_depth``.
          * ``"hr"``: the rendered pixels themselves -- ``fine_comp_rgbs``, 12 B per ray (9.1 MB for config
            #4) -- the reference's test-time deliverable (``unflatten_reshape`` of ``out_fine_comp_rgbs_ori``,
            models/nerf_downX_model.py:410-416, feeding calculate_vis :430-450 and test :621-669).  Blocks are
            LR-pixel-major, so the gathered buffer IS the (n_lr * s*s, 3) ray-major tensor ``unflatten_reshape`` takes:
            every rank returns ``hr_rgb`` (H, W, 3), and ``lr_rgb`` as the s*s means of the gathered rays (the same
            arithmetic on the same values as the per-block means).  ``lr_depth`` is not part of this exchange.
        ``want_weights`` (default False): a frame render does not use the per-sample ``weights`` arrays of the reference's
        8-entry dict; leaving them out saves 195 of the 261 MB the fine-pass launch moves through HBM on config #2
        (``forward_rays`` itself, the drop-in for the reference's method, still returns all eight).
        ``lr_range`` overrides this rank's block and skips the collective (single-process tests).  Returns the assembled
        arrays plus this rank's own outputs (``local``) and ``bytes_per_rank`` (payload of the collective)."""
        from . import dist as nsr_dist
        if gather not in ("lr", "hr"):
            raise ValueError("gather must be 'lr' or 'hr'")
        opt = self.opt
        s, s2 = int(opt.downscale), int(opt.downscale) ** 2
        n_lr = (opt.img_wh[1] // s) * (opt.img_wh[0] // s)
        rank, world = nsr_dist._world(group)
        lo, hi = lr_range if lr_range is not None else nsr_dist.shard_bounds(n_lr, world)[rank]
        rays = ops.subpixel_rays(c2w, opt.img_wh, focal, s, ndc, near, far, self.device, lr_range=(lo, hi)).view(-1, 8)
        fine = opt.N_importance > 0
        if self.fused:
            out = ops.forward_rays(self.netCoarse, self.netFine if fine else None, rays, opt.N_coarse, opt.N_importance,
                                   opt.white_bkgd, opt.lindisp, workspace=workspace, outs=outs, want_weights=want_weights,
                                   sigma_activation=self.renderer.sigma_activation)
        else:               # a GenericMLP pair: the eval-mode staged route
            was, self.randomized = self.randomized, False
            try:
                out = self.forward_rays(rays)
            finally:
                self.randomized = was
        tag = "fine" if fine else "coarse"
        cap = (lo, hi) if lr_range is not None else nsr_dist.shard_bounds(n_lr, world)[0]   # the block the payload is sized by
        if gather == "hr":
            # one row per LR pixel holding its s*s rendered rays: the block structure all_gather_pixels assumes
            local = out[f"{tag}_comp_rgbs"].reshape(hi - lo, s2 * 3)
            full = local if lr_range is not None else nsr_dist.all_gather_pixels(local, n_lr, group)
            res = {"lr_range": (lo, hi), "local": out, "bytes_per_rank": (cap[1] - cap[0]) * s2 * 12}
            if lr_range is None:
                hr_rays = full.reshape(n_lr * s2, 3)
                res["hr_rgb"] = self.unflatten_reshape(hr_rays)
                res["lr_rgb"] = ops.sr_mean(hr_rays, n_lr, s2)
            else:
                res["hr_rays"] = full.reshape((hi - lo) * s2, 3)
            return res
        local = torch.empty(hi - lo, 4, dtype=torch.float32, device=self.device)
        if hi > lo:
            local[:, :3] = ops.sr_mean(out[f"{tag}_comp_rgbs"], hi - lo, s2)
            local[:, 3:] = ops.sr_mean(out[f"{tag}_depth"], hi - lo, s2)
        full = local if lr_range is not None else nsr_dist.all_gather_pixels(local, n_lr, group)
        return {"lr_rgb": full[:, :3], "lr_depth": full[:, 3], "lr_range": (lo, hi), "local": out,
                "bytes_per_rank": (cap[1] - cap[0]) * 16}
