"""CPU ORACLE for the NeRF-SR supersampled render hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU restatement of the reference algorithm for the
one hot path this repository accelerates.  It exists so that the
HIP kernels have something to be checked against on a machine where
``/root/reference`` does not exist (the GPU box).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product (``nerf_sr_amd``) never does and fails loudly when its HIP
library is missing.

Pinning: the reference ships no tests and no golden vectors, so the
oracle is pinned against outputs of the reference itself, produced in the
development container by ``tests/golden/make_golden.py`` (which imports the
reference through a ``sys.modules`` shim) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every function below against those
fixtures (bit-exact for fp32 on the same torch build; tolerance 2e-6 otherwise).

Every function cites the reference site it restates (paths relative to the
reference tree).  Arithmetic order is kept identical to the reference so that
fp32 results agree bit-for-bit on the same ATen build; ``dtype=torch.float64``
gives a high-precision ground truth for error budgeting.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------
# R1-R4: sub-pixel ray generation
# ----------------------------------------------------------------------------

def ray_directions(H: int, W: int, focal: float, use_pixel_centers: bool = True) -> torch.Tensor:
    """Camera-space direction of every pixel, (H, W, 3) fp32.

    Restates ``models/utils.py:98-126`` (get_ray_directions): pixel grid built in
    numpy float32 with ``indexing='xy'``, ``((i - W/2)/f, -(j - H/2)/f, -1)``.
    """
    c = 0.5 if use_pixel_centers else 0.0
    xs = np.arange(W, dtype=np.float32) + c
    ys = np.arange(H, dtype=np.float32) + c
    gx, gy = np.meshgrid(xs, ys, indexing="xy")
    gx, gy = torch.from_numpy(gx), torch.from_numpy(gy)
    return torch.stack([(gx - W / 2) / focal, -(gy - H / 2) / focal, -torch.ones_like(gx)], -1)


def rays_from_pose(directions: torch.Tensor, c2w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """World-space origin and *normalised* direction per pixel, each (H*W, 3).

    Restates ``models/utils.py:129-152`` (get_rays).
    """
    d = directions @ c2w[:, :3].T
    d = d / torch.norm(d, dim=-1, keepdim=True)
    o = c2w[:, 3].expand(d.shape)
    return o.reshape(-1, 3), d.reshape(-1, 3)


def ndc_rays(H: int, W: int, focal: float, near: float, o: torch.Tensor, d: torch.Tensor):
    """Forward-facing NDC warp of rays (LLFF only).  Restates ``models/utils.py:155-196``."""
    t = -(near + o[..., 2]) / d[..., 2]
    o = o + t[..., None] * d
    ox_oz = o[..., 0] / o[..., 2]
    oy_oz = o[..., 1] / o[..., 2]
    o0 = -1.0 / (W / (2.0 * focal)) * ox_oz
    o1 = -1.0 / (H / (2.0 * focal)) * oy_oz
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = -1.0 / (W / (2.0 * focal)) * (d[..., 0] / d[..., 2] - ox_oz)
    d1 = -1.0 / (H / (2.0 * focal)) * (d[..., 1] / d[..., 2] - oy_oz)
    d2 = 1 - o2
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def subpixel_ray_grid(c2w: torch.Tensor, H: int, W: int, focal: float, s: int, ndc: bool,
                      near: float, far: float) -> torch.Tensor:
    """Full R1-R4 chain: the (H/s * W/s, s*s, 8) ray tensor one test pose yields.

    Restates the ray construction of ``data/llff_downX_dataset.py:473-494`` (NDC:
    ``get_rays`` -> ``get_ndc_rays(near=1.0)`` -> near/far := 0/1) and
    ``data/blender_downX_dataset.py:207-215`` (no NDC, near/far given), followed by
    the regroup ``'(h s1) (w s2) c -> (h w) (s1 s2) c'`` that makes every LR pixel
    own its s*s HR sub-pixel rays (sub-pixel index = dy*s + dx).
    H, W are the HR image size; focal the HR focal length.
    """
    dirs = ray_directions(H, W, focal)
    o, d = rays_from_pose(dirs, c2w)
    if ndc:
        o, d = ndc_rays(H, W, focal, 1.0, o, d)
        nr, fr = 0.0, 1.0
    else:
        nr, fr = near, far
    nr_t = nr * torch.ones_like(o[:, :1])
    fr_t = fr * torch.ones_like(o[:, :1])
    rays = torch.cat([o, d, nr_t, fr_t], 1).view(H, W, 8)
    h, w = H // s, W // s
    rays = rays.view(h, s, w, s, 8).permute(0, 2, 1, 3, 4).reshape(h * w, s * s, 8)
    return rays


# ----------------------------------------------------------------------------
# E1: positional encoding
# ----------------------------------------------------------------------------

def posenc(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """``[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]`` -> (B, 3 + 6*n_freqs).

    Restates ``models/embedding.py:39-40,44-62`` (log-scale bands, xyz included).
    """
    bands = 2 ** torch.linspace(0, n_freqs - 1, n_freqs)
    parts = [x]
    for f in bands:
        parts.append(torch.sin(f * x))
        parts.append(torch.cos(f * x))
    return torch.cat(parts, -1)


# ----------------------------------------------------------------------------
# S1 / S2: stratified sampling and inverse-CDF resampling
# ----------------------------------------------------------------------------

def points_on_rays(o: torch.Tensor, d: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """``o + z*d`` -> (R, N, 3).  Restates ``models/utils.py:5-14`` (cast_rays)."""
    return o[..., None, :] + z[..., None] * d[..., None, :]


def sample_coarse(o, d, near, far, n: int, lindisp: bool = False, u: torch.Tensor = None):
    """Coarse depths z (R, n) and points (R, n, 3).

    Restates ``models/utils.py:17-44`` (sample_along_rays).  ``u`` replaces the
    reference's ``torch.rand_like`` when the randomized (training) branch is wanted;
    ``u=None`` is the deterministic eval path.
    """
    t = torch.linspace(0, 1, n, dtype=o.dtype)
    if lindisp:
        z = 1.0 / (1.0 / near * (1 - t) + 1.0 / far * t)
    else:
        z = near * (1 - t) + far * t
    if u is not None:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        hi = torch.cat([mid, z[:, -1:]], -1)
        lo = torch.cat([z[:, :1], mid], -1)
        z = lo + u * (hi - lo)
    return z, points_on_rays(o, d, z)


def resample_fine(o, d, z, weights, n_imp: int, u: torch.Tensor = None):
    """Hierarchical inverse-CDF resampling: merged, sorted z (R, N+n_imp) and points.

    Restates ``models/utils.py:47-95`` (resample_along_rays): bins = interval
    midpoints, pdf from ``weights[:, 1:-1] + 1e-5``, ``searchsorted(right=True)``,
    ``denom < 1e-5 -> 1``, then ``sort(cat([z, z_new]))``.  ``u=None`` is the eval
    path (``linspace(0, 1, n_imp)``); pass ``u`` (R, n_imp) for the randomized one.
    """
    eps = 1e-5
    bins = 0.5 * (z[:, :-1] + z[:, 1:])
    w = weights[:, 1:-1]
    R, nb = w.shape
    w = w + eps
    pdf = w / w.sum(dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    if u is None:
        u = torch.linspace(0, 1, n_imp, dtype=z.dtype).expand(R, n_imp)
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp_min(idx - 1, 0)
    hi = torch.clamp_max(idx, nb)
    pair = torch.stack([lo, hi], -1).view(R, -1)
    cdf_g = torch.gather(cdf, 1, pair).view(R, -1, 2)
    bins_g = torch.gather(bins, 1, pair).view(R, -1, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom[denom < eps] = 1
    z_new = bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])
    z_all = torch.sort(torch.cat([z, z_new], -1), -1)[0]
    return z_all, points_on_rays(o, d, z_all)


def resample_conditioning(z, weights, n_imp: int, u: torch.Tensor = None):
    """Test helper (not part of the reference): how ill-conditioned is S2 on these rays?

    The inverse CDF divides by ``denom = cdf[above] - cdf[below]`` and the reference snaps
    ``denom < 1e-5`` to 1 (``models/utils.py:87-88``).  cdf entries carry ~1e-7 of fp32
    rounding noise (sum order of ``weights.sum``), so a new sample moves by
    ``~1e-7 * (bins_a - bins_b) / denom`` and, when ``denom`` sits within that noise of the
    1e-5 threshold, jumps by up to a whole bin between two correct fp32 evaluations
    (e.g. the reference on CPU vs on a GPU).  Returns per ray, computed in float64:
    ``amp`` = max over new samples of ``(bins_a - bins_b) / denom``, ``margin`` = min over
    new samples of ``|denom_raw - 1e-5|``, and ``bin_width`` = max interval between bins.
    """
    z, weights = z.double(), weights.double()
    eps = 1e-5
    bins = 0.5 * (z[:, :-1] + z[:, 1:])
    w = weights[:, 1:-1] + eps
    R, nb = w.shape
    cdf = torch.cumsum(w / w.sum(-1, keepdim=True), -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    if u is None:
        u = torch.linspace(0, 1, n_imp, dtype=torch.float64).expand(R, n_imp)
    u = u.double().contiguous()
    # per-bin slope of the inverse CDF and distance of its denom from the snap threshold
    denom_raw = cdf[:, 1:] - cdf[:, :-1]                       # (R, nb)
    width = bins[:, 1:] - bins[:, :-1]
    slope = width / torch.where(denom_raw < eps, torch.ones_like(denom_raw), denom_raw)
    dist = (denom_raw - eps).abs()
    # bin each sample falls in; a sample within 1e-6 of a bin edge may be evaluated in the
    # neighbouring bin by another correct fp32 implementation (incl. u = 1 at the clamped end)
    j = torch.clamp(torch.searchsorted(cdf, u, right=True) - 1, 0, nb - 1)
    amp = torch.gather(slope, 1, j)
    margin = torch.gather(dist, 1, j)
    for dj in (-1, 1):
        jn = torch.clamp(j + dj, 0, nb - 1)
        edge = torch.gather(cdf, 1, j + (1 if dj == 1 else 0))
        near = (u - edge).abs() < 1e-6
        amp = torch.where(near, torch.maximum(amp, torch.gather(slope, 1, jn)), amp)
        margin = torch.where(near, torch.minimum(margin, torch.gather(dist, 1, jn)), margin)
    return amp.max(-1)[0], margin.min(-1)[0], width.max(-1)[0]


# ----------------------------------------------------------------------------
# M1: the NeRF MLP
# ----------------------------------------------------------------------------

def mlp_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, sigma_only: bool = False,
                color_activation: str = "sigmoid", stop_grad: bool = False) -> torch.Tensor:
    """``VanillaMLP.forward`` on embedded rows x (B, 90) -> (B, 4) = [rgb, sigma_raw].

    Restates ``models/networks.py:182-226`` (D=8, W=256, skips=[4] in every script; any other ``--D --W --skips`` is
    read off the tensors' shapes): a skip layer sees ``cat([pe, h])`` (input first); sigma is the raw Linear output (the
    ReLU is applied by the renderer); colour goes through sigmoid, or through nothing
    with ``color_activation='none'`` (``:173-180``).  A ``--no_dir`` network
    (``:160-169, 213-216``: ``dir_encoding.0.weight`` of shape (128, 256)) feeds
    ``xyz_encoding_final`` alone to ``dir_encoding``.  ``stop_grad`` (``:218-219``): the input of ``dir_encoding`` is detached
    (matters under autograd only: oracle/train_oracle.py).
    ``sd`` is the 24-key state_dict (torch tensors).
    """
    lin = torch.nn.functional.linear
    # the architecture flags (--D --W --skips, encoding degrees: models/networks.py:124-157) are read off the tensors: the
    # first layer's fan-in is the encoded position's width, a trunk layer whose fan-in is W + that width is a skip layer
    n_xyz = sd["xyz_encoding_1.0.weight"].shape[1]
    width = sd["xyz_encoding_1.0.weight"].shape[0]
    depth = sum(1 for k in sd if k.startswith("xyz_encoding_") and k.endswith(".0.weight"))
    pe, de = x[:, :n_xyz], x[:, n_xyz:]
    h = pe
    for i in range(depth):
        if i > 0 and sd[f"xyz_encoding_{i + 1}.0.weight"].shape[1] == width + n_xyz:
            h = torch.cat([pe, h], -1)
        h = torch.relu(lin(h, sd[f"xyz_encoding_{i + 1}.0.weight"], sd[f"xyz_encoding_{i + 1}.0.bias"]))
    sigma = lin(h, sd["sigma.weight"], sd["sigma.bias"])
    if sigma_only:
        return sigma
    g = lin(h, sd["xyz_encoding_final.weight"], sd["xyz_encoding_final.bias"])
    no_dir = sd["dir_encoding.0.weight"].shape[1] == g.shape[1]
    dir_in = g if no_dir else torch.cat([g, de], -1)
    c = torch.relu(lin(dir_in.detach() if stop_grad else dir_in, sd["dir_encoding.0.weight"], sd["dir_encoding.0.bias"]))
    rgb = lin(c, sd["rgb.0.weight"], sd["rgb.0.bias"])
    if color_activation == "sigmoid":
        rgb = torch.sigmoid(rgb)
    elif color_activation != "none":
        raise ValueError(color_activation)
    return torch.cat([rgb, sigma], -1)


def render_points(sd, xyz: torch.Tensor, dir_embedded: torch.Tensor, point_chunk: int = 262144,
                  gamma_correct: bool = False, color_activation: str = "sigmoid", stop_grad: bool = False, deg_pos: int = 10):
    """(R, N, 3) points + (R, 27) dir embedding -> rgb (R, N, 3), sigma (R, N).

    Restates ``models/nerf_downX_model.py:260-278`` (render_rays): PE of the points,
    ``repeat_interleave`` of the per-ray dir embedding, concat to (P, 90), MLP in
    ``point_chunk`` slices (``utils/utils.py:130-152``); ``gamma_correct``: ``out_rgbs = pow(out_rgbs, 1/2.2)``
    (``:271-276``, ``--gamma_correct``).
    """
    R, N = xyz.shape[:2]
    pts = xyz.reshape(-1, 3)
    x = torch.cat([posenc(pts, deg_pos), dir_embedded.repeat_interleave(N, dim=0)], -1)
    outs = [mlp_forward(sd, x[i:i + point_chunk], color_activation=color_activation, stop_grad=stop_grad)
            for i in range(0, x.shape[0], point_chunk)]
    out = torch.cat(outs, 0).view(R, N, -1)
    rgb = out[..., :-1]
    if gamma_correct:
        rgb = torch.pow(rgb, 1 / 2.2)
    return rgb, out[..., -1]


# ----------------------------------------------------------------------------
# V1: volumetric compositing
# ----------------------------------------------------------------------------

def composite(rgb: torch.Tensor, sigma: torch.Tensor, z: torch.Tensor, white_bkgd: bool, sigma_activation: str = "relu"):
    """sigma -> alpha -> transmittance -> weights -> (rgb, depth, opacity, weights).

    Restates ``models/rendering.py:75-111`` (VolumetricRenderer.forward): last
    delta is 1e10, ``T_k = prod_{j<k}(1 - alpha_j + 1e-10)``, deltas are not scaled
    by |d|, relu on sigma -- or, ``sigma_activation='softplus'`` (``:69-73``), ``log(1 + exp(sigma - 1))``.
    """
    delta = z[:, 1:] - z[:, :-1]
    delta = torch.cat([delta, 1e10 * torch.ones_like(delta[:, :1])], -1)
    if sigma_activation == "relu":
        dens = torch.relu(sigma)
    elif sigma_activation == "softplus":
        dens = torch.log(1 + torch.exp(sigma - 1))
    else:
        raise ValueError(sigma_activation)
    alpha = 1 - torch.exp(-delta * dens)
    trans = torch.cat([torch.ones_like(alpha[:, :1]), torch.cumprod(1 - alpha[:, :-1] + 1e-10, -1)], -1)
    w = alpha * trans
    comp = (w[..., None] * rgb).sum(-2)
    depth = (w * z).sum(-1)
    opacity = w.sum(-1)
    if white_bkgd:
        comp = comp + (1 - opacity[..., None])
    return comp, depth, opacity, w


# ----------------------------------------------------------------------------
# D3: forward_rays, A1: s^2 mean, A2: unflatten
# ----------------------------------------------------------------------------

def forward_rays(sd_coarse, sd_fine, rays: torch.Tensor, n_coarse: int = 64, n_importance: int = 64,
                 white_bkgd: bool = False, lindisp: bool = False, ray_chunk: int = 4096, gamma_correct: bool = False,
                 sigma_activation: str = "relu", color_activation: str = "sigmoid", deg_pos: int = 10, deg_dir: int = 4):
    """Eval-mode ``forward_rays`` over (R, 8) rays -> dict of the 8 reference outputs.

    Restates ``models/nerf_downX_model.py:280-313`` chunked as ``:316-324``
    (``ray_chunk`` = 4096, ``options/base_options.py:69``).  Ray columns:
    ``[o(0:3), d(3:6), near(6), far(7)]``; 11-wide rows are the vanilla model's
    (``models/nerf_model.py:207-242``): the encoded view direction is columns 8:11.
    """
    outs = []
    for i in range(0, rays.shape[0], ray_chunk):
        r = rays[i:i + ray_chunk]
        o, d, near, far = r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8]
        de = posenc(r[:, 8:11] if r.shape[1] == 11 else d, deg_dir)
        z, xyz = sample_coarse(o, d, near, far, n_coarse, lindisp)
        rgb, sig = render_points(sd_coarse, xyz, de, gamma_correct=gamma_correct, color_activation=color_activation, deg_pos=deg_pos)
        c_rgb, c_depth, c_op, c_w = composite(rgb, sig, z, white_bkgd, sigma_activation)
        res = {"coarse_comp_rgbs": c_rgb, "coarse_depth": c_depth, "coarse_opacity": c_op,
               "coarse_weights": c_w}
        if n_importance > 0:
            z2, xyz2 = resample_fine(o, d, z, c_w, n_importance)
            rgb2, sig2 = render_points(sd_fine, xyz2, de, gamma_correct=gamma_correct, color_activation=color_activation, deg_pos=deg_pos)
            f_rgb, f_depth, f_op, f_w = composite(rgb2, sig2, z2, white_bkgd, sigma_activation)
            res.update({"fine_comp_rgbs": f_rgb, "fine_depth": f_depth, "fine_opacity": f_op,
                        "fine_weights": f_w})
        outs.append(res)
    return {k: torch.cat([o_[k] for o_ in outs], 0) for k in outs[0]}


def sr_mean(hr: torch.Tensor, n_lr: int, s2: int) -> torch.Tensor:
    """LR value = mean over the s*s sub-pixel rays of each LR pixel.

    Restates ``models/nerf_downX_model.py:337-348`` (``reshape(N_lr, s^2, -1).mean(1)``).
    """
    return torch.mean(torch.reshape(hr, (n_lr, s2, -1)), dim=1)


def unflatten_hr(x: torch.Tensor, H: int, W: int, s: int) -> torch.Tensor:
    """(N_lr*s^2, c) in LR-pixel-major / sub-pixel order -> HR image (H, W, c).

    Restates ``models/nerf_downX_model.py:410-416`` (the inverse of the R4 regroup).
    """
    h, w = H // s, W // s
    return x.reshape(h, w, s, s, -1).permute(0, 2, 1, 3, 4).reshape(H, W, -1)


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """``-10 log10(mean((a-b)^2))``.  Restates ``models/criterions.py:27-36``."""
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)


# ----------------------------------------------------------------------------
# helpers shared by tests / bench (not part of the reference restatement)
# ----------------------------------------------------------------------------

def to_torch_sd(sd_np, dtype=torch.float32):
    """numpy state_dict -> torch CPU tensors of ``dtype``."""
    return {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in sd_np.items()}
