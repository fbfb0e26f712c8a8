"""Weight interface of the render hot path.

The reference hands the NeRF MLP to the path as the 24-tensor ``state_dict()``
of ``VanillaMLP`` (reference ``models/networks.py:131-180``; checkpoint writer
``models/base_model.py:181-196``).  nn.Linear layout: ``weight`` is
``(out, in)`` row-major, ``y = x @ W.T + b``.

This module owns
  * the key/shape table of that state_dict (``STATE_DICT_SPEC``),
  * a deterministic, seedable generator for synthetic weights (no checkpoints
    can be downloaded), used by tests, the golden-vector script and
    ``bench.py`` so that no weight blob has to be committed.

Everything here is numpy; nothing touches the GPU.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

# Architecture constants of the path (reference defaults):
#   D=8, W=256, skips=[4]  (models/networks.py:124-126)
#   deg_pos=10, deg_dir=4  (models/nerf_model.py:56-57)
D_LAYERS = 8
WIDTH = 256
SKIP_LAYER = 4            # 0-based index of the trunk layer whose input is cat([pe, h])
DEG_POS = 10
DEG_DIR = 4
POS_CH = 3 + 3 * 2 * DEG_POS   # 63
DIR_CH = 3 + 3 * 2 * DEG_DIR   # 27
IN_CH = POS_CH + DIR_CH        # 90
MACS_PER_POINT = 593_408
FLOP_PER_POINT = 2 * MACS_PER_POINT


def _spec():
    spec = OrderedDict()
    for i in range(D_LAYERS):
        if i == 0:
            fan_in = POS_CH
        elif i == SKIP_LAYER:
            fan_in = WIDTH + POS_CH
        else:
            fan_in = WIDTH
        spec[f"xyz_encoding_{i + 1}.0.weight"] = (WIDTH, fan_in)
        spec[f"xyz_encoding_{i + 1}.0.bias"] = (WIDTH,)
    spec["xyz_encoding_final.weight"] = (WIDTH, WIDTH)
    spec["xyz_encoding_final.bias"] = (WIDTH,)
    spec["dir_encoding.0.weight"] = (WIDTH // 2, WIDTH + DIR_CH)
    spec["dir_encoding.0.bias"] = (WIDTH // 2,)
    spec["sigma.weight"] = (1, WIDTH)
    spec["sigma.bias"] = (1,)
    spec["rgb.0.weight"] = (3, WIDTH // 2)
    spec["rgb.0.bias"] = (3,)
    return spec


#: key -> shape, in the order ``VanillaMLP.state_dict()`` yields them
STATE_DICT_SPEC = _spec()
N_PARAMS = sum(int(np.prod(s)) for s in STATE_DICT_SPEC.values())
assert N_PARAMS == 595_844
assert sum(int(np.prod(s)) for k, s in STATE_DICT_SPEC.items() if k.endswith("weight")) == MACS_PER_POINT


#: synthetic density-field presets: (spectral decay of the PE input columns, sigma.weight scale, sigma.bias)
FIELDS = {
    "plain": (0.0, 1.0, None),     # kaiming only: fog everywhere, every ray saturates
    "smooth": (1.0, 10.0, -3.0),   # default: 1/f spectrum, ~45 % empty samples, opacity varies ray to ray
    "sharp": (0.0, 30.0, -40.0),   # stress: white spectrum x30 density head, chaotic under fp32 rounding
}


def make_state_dict(seed: int = 99, field: str = "smooth", bias_scale: float = 0.05):
    """Deterministic synthetic ``VanillaMLP`` weights as ``{key: float32 ndarray}``.

    ``W ~ N(0, 2/fan_in)`` mirrors ``init_type=kaiming`` of the reference
    (``models/networks.py:31-38``: kaiming_normal_, fan_in, a=0).  The reference
    zero-inits biases; a small non-zero bias (``bias_scale``) is used here so
    that bias handling of the kernels is actually exercised by the parity tests.

    ``field`` shapes the random density field so that renders are non-trivial
    (no dataset or checkpoint is reachable offline):

    * ``"smooth"`` (default) scales the positional-encoding columns of octave k in
      the two layers that read them (``xyz_encoding_1``, ``xyz_encoding_5``) by
      ``2^-k`` — a 1/f spectrum, band-limited at the sample spacing like a trained
      scene — and sets ``sigma.weight *= 10, sigma.bias = -3``: ~45 % of the
      samples are empty, opacity mean 0.3-0.9 with std ~0.3 across rays.  On this
      field the reference algorithm itself is well conditioned (fp32 vs fp64
      oracle differ by < 1e-5 RGB), so an all-rays 1e-4 parity bound is meaningful.
    * ``"sharp"`` (white spectrum, ``sigma.weight *= 30, sigma.bias = -40``) is a
      stress field: the reference's ``denom < 1e-5 -> 1`` snap
      (``models/utils.py:87-88``) and the 1/pdf amplification of the inverse CDF
      make its own fp32 and fp64 results differ by > 1e-2 on ~5 % of the rays, so
      only statistical parity can be asked of it.
    * ``"plain"``: kaiming weights only.
    """
    if field not in FIELDS:
        raise ValueError(f"field must be one of {list(FIELDS)}")
    decay, sigma_scale, sigma_bias = FIELDS[field]
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for key, shape in STATE_DICT_SPEC.items():
        if key.endswith("weight"):
            fan_in = shape[1]
            w = rng.standard_normal(shape, dtype=np.float64) * np.sqrt(2.0 / fan_in)
            sd[key] = w.astype(np.float32)
        else:
            sd[key] = (rng.standard_normal(shape, dtype=np.float64) * bias_scale).astype(np.float32)
    if decay > 0:
        for key in ("xyz_encoding_1.0.weight", f"xyz_encoding_{SKIP_LAYER + 1}.0.weight"):
            w = sd[key].copy()
            for k in range(DEG_POS):
                w[:, 3 + 6 * k: 9 + 6 * k] *= np.float32(2.0 ** (-decay * k))
            sd[key] = w
    sd["sigma.weight"] = (sd["sigma.weight"] * np.float32(sigma_scale)).astype(np.float32)
    if sigma_bias is not None:
        sd["sigma.bias"] = np.full((1,), sigma_bias, dtype=np.float32)
    return sd


DIR_W = "dir_encoding.0.weight"          # (128, 256 + 27); (128, 256) in a --no_dir network (models/networks.py:160-169)


def check_state_dict(sd, no_dir: bool = False) -> None:
    """Raise ``ValueError`` unless ``sd`` has exactly the 24 keys/shapes of the path (``no_dir``: of the ``--no_dir``
    variant, whose ``dir_encoding`` layer sees ``xyz_encoding_final`` alone)."""
    missing = [k for k in STATE_DICT_SPEC if k not in sd]
    if missing:
        raise ValueError(f"state_dict is missing keys: {missing}")
    for k, shape in STATE_DICT_SPEC.items():
        got = tuple(sd[k].shape)
        want = (shape[0], shape[1] - 27) if (no_dir and k == DIR_W) else tuple(shape)
        if got != want:
            raise ValueError(f"state_dict[{k!r}] has shape {got}, expected {want}" + (" (no_dir network)" if no_dir else ""))


def pad_no_dir(w):
    """``--no_dir`` (models/networks.py:160-169, 213-216): ``dir_encoding`` is Linear(256, 128) on ``xyz_encoding_final``
    alone.  The kernels are laid out for the (128, 283) layer over ``cat([final, dir_pe])``: the same function with 27
    ZERO columns behind the 256 -- every product with them is an exact 0, so the outputs are those of the narrow layer
    bit for bit in any accumulation order.  Works on numpy arrays and torch tensors."""
    if isinstance(w, np.ndarray):
        return np.concatenate([w, np.zeros((w.shape[0], 27), dtype=w.dtype)], 1)
    import torch
    return torch.cat([w, torch.zeros(w.shape[0], 27, dtype=w.dtype, device=w.device)], 1)


# ---------------------------------------------------------------------------------------------------------------------
# The architecture flags of VanillaMLP (models/networks.py:124-128 --D --W --skips --no_dir; models/nerf_model.py:53-57
# --dim_rgb --deg_pos --deg_dir).  The fused kernels are laid out for the defaults above; any other value runs layer by
# layer (ops.GenericMLP).
# ---------------------------------------------------------------------------------------------------------------------
def arch_of(opt=None) -> dict:
    """The architecture values of an options object (absent ones = the reference's defaults)."""
    g = (lambda k, d: getattr(opt, k, d)) if opt is not None else (lambda k, d: d)
    return {"D": int(g("D", D_LAYERS)), "W": int(g("W", WIDTH)), "skips": tuple(int(s) for s in g("skips", (SKIP_LAYER,))),
            "deg_pos": int(g("deg_pos", DEG_POS)), "deg_dir": int(g("deg_dir", DEG_DIR)), "dim_rgb": int(g("dim_rgb", 3)),
            "no_dir": bool(g("no_dir", False))}


def is_default_arch(arch: dict) -> bool:
    """True when the fused kernels cover it (``no_dir`` rides them too: weights.pad_no_dir)."""
    return all(arch[k] == v for k, v in (("D", D_LAYERS), ("W", WIDTH), ("skips", (SKIP_LAYER,)), ("deg_pos", DEG_POS),
                                          ("deg_dir", DEG_DIR), ("dim_rgb", 3)))


def arch_spec(D=D_LAYERS, W=WIDTH, skips=(SKIP_LAYER,), deg_pos=DEG_POS, deg_dir=DEG_DIR, dim_rgb=3, no_dir=False):
    """key -> shape of ``VanillaMLP(opt).state_dict()`` for these flags, in the module's own order (models/networks.py:131-180)."""
    if D < 1 or W < 2 or W % 2 or deg_pos < 0 or deg_dir < 0 or dim_rgb < 1 or any(s < 1 or s >= D for s in skips):
        raise ValueError("D >= 1, even W >= 2, degrees >= 0, dim_rgb >= 1 and skip layers inside 1 .. D - 1 "
                         "(a skip at layer 0 would double the input: the reference's constructor gives layer 0 the plain width)")
    in_xyz, in_dir = 3 + 6 * deg_pos, 3 + 6 * deg_dir
    spec = OrderedDict()
    for i in range(D):
        fan_in = in_xyz if i == 0 else (W + in_xyz if i in skips else W)
        spec[f"xyz_encoding_{i + 1}.0.weight"] = (W, fan_in)
        spec[f"xyz_encoding_{i + 1}.0.bias"] = (W,)
    spec["xyz_encoding_final.weight"] = (W, W)
    spec["xyz_encoding_final.bias"] = (W,)
    spec["dir_encoding.0.weight"] = (W // 2, W + (0 if no_dir else in_dir))
    spec["dir_encoding.0.bias"] = (W // 2,)
    spec["sigma.weight"] = (1, W)
    spec["sigma.bias"] = (1,)
    spec["rgb.0.weight"] = (dim_rgb, W // 2)
    spec["rgb.0.bias"] = (dim_rgb,)
    return spec


def make_state_dict_arch(seed: int, bias_scale: float = 0.05, **arch):
    """Deterministic synthetic weights for any architecture ``arch_spec`` describes: kaiming weights, small biases, the
    "smooth" field's density head (``sigma.weight`` x 10, ``sigma.bias`` -3) and 1/f positional-encoding columns."""
    spec = arch_spec(**arch)
    deg_pos = arch.get("deg_pos", DEG_POS)
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    for key, shape in spec.items():
        if key.endswith("weight"):
            sd[key] = (rng.standard_normal(shape, dtype=np.float64) * np.sqrt(2.0 / shape[1])).astype(np.float32)
        else:
            sd[key] = (rng.standard_normal(shape, dtype=np.float64) * bias_scale).astype(np.float32)
    in_xyz = 3 + 6 * deg_pos
    for key, shape in spec.items():
        if key.startswith("xyz_encoding_") and key.endswith("0.weight") and shape[1] in (in_xyz, arch.get("W", WIDTH) + in_xyz):
            for k in range(deg_pos):
                sd[key][:, 3 + 6 * k: 9 + 6 * k] *= np.float32(2.0 ** (-k))
    sd["sigma.weight"] = (sd["sigma.weight"] * np.float32(10.0)).astype(np.float32)
    sd["sigma.bias"] = np.full((1,), -3.0, dtype=np.float32)
    return sd


def check_state_dict_arch(sd, **arch) -> None:
    spec = arch_spec(**arch)
    missing = [k for k in spec if k not in sd]
    if missing:
        raise ValueError(f"state_dict is missing keys: {missing}")
    for k, shape in spec.items():
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"state_dict[{k!r}] has shape {tuple(sd[k].shape)}, expected {tuple(shape)} for {arch}")
