"""Host-side mirror of the reference's operator interface for the render hot path.

Same names, argument meaning and return shapes as the reference functions they
replace (cited per function), implemented as calls into libnsr.so through the C
ABI.  PyTorch is plumbing only: it owns the device buffers and the stream.
All tensors are fp32, contiguous, on the current HIP device.
"""
from __future__ import annotations

import ctypes
from ctypes import c_float, c_void_p  # noqa: F401
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .weights import (STATE_DICT_SPEC, check_state_dict, pad_no_dir, DIR_W, IN_CH, arch_of, arch_spec, is_default_arch,
                      check_state_dict_arch)

__all__ = [
    "subpixel_rays", "PositionalEncoding", "sample_along_rays", "resample_along_rays", "cast_rays",
    "VanillaMLP", "VolumetricRenderer", "check_mlp_options", "render_rays", "render_rays_composited", "forward_rays", "sr_mean", "unflatten_reshape",
]


def _stream() -> c_void_p:
    """torch's current stream on the CURRENT device: libnsr enqueues on the caller's current HIP device
    (include/nsr.h), so every tensor handed to an op must live there (``_check_device``)."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_device(dev, name: str) -> None:
    """Kernels run on torch's current device against raw pointers: a tensor on another GPU would fault or be
    reached over the fabric silently.  Fail loudly instead (``torch.cuda.set_device`` / ``with torch.cuda.device``)."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise ValueError(f"{name} must live on the GPU (nerf_sr_amd has no CPU path)")
    cur = torch.cuda.current_device()
    if dev.index is not None and dev.index != cur:
        raise ValueError(f"{name} lives on cuda:{dev.index} but the current device is cuda:{cur}; "
                         f"call torch.cuda.set_device({dev.index}) (one process per GPU) before using nerf_sr_amd")


def _p(t: Optional[torch.Tensor]) -> c_void_p:
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (nerf_sr_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    _check_device(t.device, name)
    return t.contiguous()


def _ray_stride(rays: torch.Tensor) -> int:
    """8: nerf_downX rows [o, d, near, far]; 11: vanilla rows with the view direction in columns 8:11."""
    if rays.ndim != 2 or rays.shape[1] not in (8, 11):
        raise ValueError(f"rays must be (R, 8) or (R, 11), got {tuple(rays.shape)}")
    return int(rays.shape[1])


def _pack_rays(ori, dir, near, far) -> torch.Tensor:
    R = ori.shape[0]
    near = near.reshape(R, 1) if isinstance(near, torch.Tensor) else torch.full((R, 1), float(near), device=ori.device)
    far = far.reshape(R, 1) if isinstance(far, torch.Tensor) else torch.full((R, 1), float(far), device=ori.device)
    return torch.cat([ori, dir, near, far], 1).contiguous()


# ----------------------------------------------------------------------------- R1-R4
def subpixel_rays(c2w, img_wh: Tuple[int, int], focal: float, downscale: int, ndc: bool,
                  near: float = 0.0, far: float = 1.0, device="cuda",
                  lr_range: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """(H/s*W/s, s*s, 8) sub-pixel ray tensor of one pose, generated on the device.

    Replaces get_ray_directions + get_rays (+ get_ndc_rays) + the einops regroup
    (models/utils.py:98-196, data/llff_downX_dataset.py:473-490,
    data/blender_downX_dataset.py:207-215).  ``img_wh`` is the HR size (W, H).
    ``lr_range = (lo, hi)`` generates only the LR pixels [lo, hi) (row-major LR index): the ray shard of one
    GPU when the frame is cut into contiguous LR-pixel blocks -> (hi - lo, s*s, 8).
    """
    lib = _lib.load()
    W, H = int(img_wh[0]), int(img_wh[1])
    s = int(downscale)
    c = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(12))
    lo, hi = (0, (H // s) * (W // s)) if lr_range is None else (int(lr_range[0]), int(lr_range[1]))
    _check_device(device, "device")
    rays = torch.empty(max(hi - lo, 0), s * s, 8, dtype=torch.float32, device=device)
    _lib.check(lib.nsr_gen_rays_range(c.ctypes.data_as(ctypes.POINTER(c_float)), H, W, float(focal), s, int(bool(ndc)),
                                      float(near), float(far), lo, hi, _p(rays), _stream()), "nsr_gen_rays_range")
    return rays


# ----------------------------------------------------------------------------- E1
class PositionalEncoding:
    """``x -> [x, sin(2^k x), cos(2^k x)]_k``; mirrors models/embedding.py:14-62
    (log-scale bands, xyz included; 3 input channels)."""

    def __init__(self, in_channels: int = 3, N_freqs: int = 10, opt=None):
        if in_channels != 3:
            raise ValueError("the built path encodes 3-channel inputs only")
        self.in_channels, self.N_freqs = in_channels, int(N_freqs)

    @property
    def out_channels(self) -> int:
        return self.in_channels * (2 * self.N_freqs + 1)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = _f32(x, "x")
        if x.ndim != 2 or x.shape[1] != 3:
            raise ValueError(f"x must be (B, 3), got {tuple(x.shape)}")
        out = torch.empty(x.shape[0], self.out_channels, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().nsr_posenc(_p(x), x.shape[0], self.N_freqs, _p(out), _stream()), "nsr_posenc")
        return out


# ----------------------------------------------------------------------------- S1 / S2
def cast_rays(ori, dir, z_vals):
    """``ori + z*dir`` (models/utils.py:5-14) — kept for API parity; torch elementwise."""
    return ori[..., None, :] + z_vals[..., None] * dir[..., None, :]


def sample_along_rays(ori, dir, near, far, num_samples: int, randomized: bool, lindisp: bool,
                      u: Optional[torch.Tensor] = None):
    """Stratified depths and points; mirrors models/utils.py:17-44.

    ``randomized=True`` draws the per-bin jitter with ``torch.rand`` on the device
    unless ``u`` (R, num_samples) is supplied (used by the parity tests).
    Returns ``(z_vals (R,N), points (R,N,3))``.
    """
    rays = _pack_rays(_f32(ori, "ori"), _f32(dir, "dir"), near, far)
    R = rays.shape[0]
    if randomized and u is None:
        u = torch.rand(R, num_samples, device=rays.device)
    if u is not None:
        u = _f32(u, "u")
    z = torch.empty(R, num_samples, dtype=torch.float32, device=rays.device)
    pts = torch.empty(R, num_samples, 3, dtype=torch.float32, device=rays.device)
    _lib.check(_lib.load().nsr_sample_along_rays(_p(rays), 8, R, num_samples, int(bool(lindisp)), _p(u), _p(z), _p(pts),
                                                 _stream()), "nsr_sample_along_rays")
    return z, pts


def resample_along_rays(ori, dir, z_vals, weights, num_samples: int, randomized: bool,
                        u: Optional[torch.Tensor] = None):
    """Inverse-CDF importance resampling + merge; mirrors models/utils.py:47-95.
    Returns ``(z_vals (R, N+num_samples) sorted, points)``."""
    ori, dir = _f32(ori, "ori"), _f32(dir, "dir")
    z_vals, weights = _f32(z_vals, "z_vals"), _f32(weights, "weights")
    R, Nc = z_vals.shape
    rays = _pack_rays(ori, dir, 0.0, 1.0)
    if randomized and u is None:
        u = torch.rand(R, num_samples, device=z_vals.device)
    if u is not None:
        u = _f32(u, "u")
    z_out = torch.empty(R, Nc + num_samples, dtype=torch.float32, device=z_vals.device)
    pts = torch.empty(R, Nc + num_samples, 3, dtype=torch.float32, device=z_vals.device)
    _lib.check(_lib.load().nsr_resample_along_rays(_p(rays), 8, _p(z_vals), _p(weights), R, Nc, num_samples, _p(u),
                                                   _p(z_out), _p(pts), _stream()), "nsr_resample_along_rays")
    return z_out, pts


# ----------------------------------------------------------------------------- M1
# the one architecture the HIP kernels are built for: the values every script of the reference uses
_MLP_FIXED = {"D": 8, "W": 256, "skips": [4], "deg_pos": 10, "deg_dir": 4, "dim_pos": 3, "dim_dir": 3, "dim_rgb": 3}
_GENERIC_FIXED = {"dim_pos": 3, "dim_dir": 3, "dim_rgb": 3}     # what the ray kernels and the compositor cannot vary either
#                                                                   (GenericMLP on its own takes any dim_rgb)
# built since round 5; stop_grad (a detach: models/networks.py:218-219) changes nothing in a forward pass and is an option of
# train.Trainer
_MLP_CHOICES = {"color_activation": ("sigmoid", "none"), "no_dir": (False, True), "stop_grad": (False, True)}


def check_mlp_options(opt, fused: bool = True) -> None:
    """Reject every VanillaMLP option value the kernels do not implement (models/networks.py:124-128 ``--D --W --skips``;
    models/embedding.py degrees) instead of silently computing the default architecture.  ``no_dir`` and
    ``color_activation`` (:160-180) are options of ``VanillaMLP`` since round 5, ``stop_grad`` of ``train.Trainer``.
    Options that are absent from ``opt`` count as the reference's defaults.  ``fused=False``: the check of ``make_mlp`` /
    ``NeRFDownXModel``, which serve other ``--D --W --skips`` / degrees / ``dim_rgb`` layer by layer (``GenericMLP``)."""
    if opt is None:
        return
    bad = []
    for name, want in (_MLP_FIXED if fused else _GENERIC_FIXED).items():
        if not hasattr(opt, name):
            continue
        got = getattr(opt, name)
        got = list(got) if isinstance(got, (list, tuple)) else got
        if got != want:
            bad.append(f"{name}={got!r} (built: {want!r})")
    for name, choices in _MLP_CHOICES.items():
        if hasattr(opt, name) and getattr(opt, name) not in choices:
            bad.append(f"{name}={getattr(opt, name)!r} (built: {list(choices)!r})")
    if bad:
        raise ValueError("VanillaMLP option(s) outside the built path (the MFMA kernels are laid out for the 8 x 256 network with "
                         "a skip at layer 5, models/networks.py:131-180): "
                         + ", ".join(bad))


class VanillaMLP:
    """The 8x256 NeRF MLP; mirrors models/networks.py:121-226.

    Weights enter as the reference's 24-key ``state_dict`` (``load_state_dict``) and
    are re-laid-out once into the MFMA fragment stream the kernel consumes
    (``nsr_pack_weights``).  ``forward(x, sigma_only=False)``: x (B, 90) embedded rows
    -> (B, 4) = [rgb, sigma] (or (B, 1)).
    """

    def __init__(self, opt=None, precision: str = "fp32", device="cuda"):
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {list(_lib.PRECISIONS)}")
        check_mlp_options(opt)
        self.precision = precision
        self._prec = _lib.PRECISIONS[precision]
        self.device = torch.device(device)
        _check_device(self.device, "VanillaMLP device")
        nbytes = _lib.load().nsr_packed_weights_bytes(self._prec)
        if nbytes == 0:
            raise _lib.NsrError(f"precision {precision!r} is not built into libnsr.so")
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._sd = None
        self._gamma = False
        self._want_gamma = bool(getattr(opt, "gamma_correct", False)) if opt is not None else False
        # --color_activation none (models/networks.py:173-180): the rgb head ends in nn.Identity; --no_dir (:160-169):
        # dir_encoding sees xyz_encoding_final alone (weights.pad_no_dir)
        self.color_activation = getattr(opt, "color_activation", "sigmoid") if opt is not None else "sigmoid"
        self.no_dir = bool(getattr(opt, "no_dir", False)) if opt is not None else False

    def _options_word(self) -> int:
        return (_lib.NSR_OPT_GAMMA if self._want_gamma else 0) | (_lib.NSR_OPT_COLOR_NONE if self.color_activation == "none" else 0)

    def load_state_dict(self, sd: Dict[str, "np.ndarray | torch.Tensor"]):
        check_state_dict(sd, self.no_dir)
        dev, packed_from = {}, {}
        for k in STATE_DICT_SPEC:
            v = sd[k]
            v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach()
            dev[k] = v.to(device=self.device, dtype=torch.float32).contiguous()
            packed_from[k] = pad_no_dir(dev[k]).contiguous() if (self.no_dir and k == DIR_W) else dev[k]
        ptrs = (c_void_p * len(dev))(*[c_void_p(t.data_ptr()) for t in packed_from.values()])
        rc = _lib.load().nsr_pack_weights(ptrs, _p(self.packed), self._prec, _stream())
        if rc == _lib.NSR_ERR_RANGE:
            # the blob HAS been overwritten with the out-of-range network (include/nsr.h): this object no longer holds a
            # usable one -- forward() / state_dict() must not go on serving the previous tensors against the new blob
            self._sd = None
            self._gamma = False
            raise _lib.NsrNumericsError(
                f"state_dict cannot be carried at precision {self.precision!r}: a weight or bias is non-finite"
                + (" or |w| >= 1023.75 (the split-fp16 stream holds 2^6 w in fp16)" if self.precision == "f16x3" else
                   " or |w| >= 65520 (fp16 operands)" if self.precision == "f16" else "")
                + "; load it with precision='fp32'", 1)
        _lib.check(rc, "nsr_pack_weights")
        self._sd = dev     # keep the fp32 originals alive (state_dict() round trip)
        self._gamma = False
        if self._options_word():   # packing clears the blob's option word: the requested colour-head options are re-applied
            self._write_options()
        return self

    def _write_options(self):
        _lib.check(_lib.load().nsr_weights_set_options(_p(self.packed), self._prec, self._options_word(), _stream()),
                   "nsr_weights_set_options")
        self._gamma = self._want_gamma

    def set_gamma_correct(self, enable: bool = True):
        """``--gamma_correct`` (models/nerf_downX_model.py:271-276): the colour head returns ``rgb ** (1 / 2.2)``.
        An option of the PACKED network: call it after ``load_state_dict`` (re-packing clears it)."""
        if self._sd is None:
            raise RuntimeError("VanillaMLP.set_gamma_correct called before load_state_dict")
        self._want_gamma = bool(enable)       # remembered: every later load_state_dict re-applies it
        self._write_options()
        return self

    def status(self, clear: bool = False) -> int:
        """The network's sticky numerics status word (``NSR_FLAG_*`` bits, include/nsr.h): 0 = every launch through this
        network so far saw finite inputs, in-range activations and finite outputs.  Waits for the current stream."""
        if self._sd is None:
            raise RuntimeError("VanillaMLP.status called before load_state_dict (the blob, and its status word, are not initialised)")
        flags = ctypes.c_uint(0)
        _lib.check(_lib.load().nsr_weights_status(_p(self.packed), self._prec, int(bool(clear)), ctypes.byref(flags), _stream()),
                   "nsr_weights_status")
        return int(flags.value)

    def check(self, what: str = "network"):
        """Raise ``NsrNumericsError`` if the status word is set (and clear it): the reference drops into pdb on NaN
        colours (models/nerf_downX_model.py:273-274); this is the error a replacement reports instead."""
        flags = self.status(clear=True)
        if flags:
            names = _lib.flag_names(flags)
            hint = " -- values left the split-fp16 operand range: re-run with precision='fp32'" \
                if (flags & 4 or (flags & 2 and self.precision in ("f16x3", "f16"))) else ""
            raise _lib.NsrNumericsError(f"{what} ({self.precision}): numerics status {flags:#x} = {' | '.join(names)}{hint}", flags)
        return self

    def state_dict(self):
        if self._sd is None:
            raise RuntimeError("no weights loaded")
        return {k: v.clone() for k, v in self._sd.items()}

    def forward(self, x: torch.Tensor, sigma_only: bool = False) -> torch.Tensor:
        if self._sd is None:
            raise RuntimeError("VanillaMLP.forward called before load_state_dict")
        x = _f32(x, "x")
        if x.ndim != 2 or x.shape[1] != IN_CH:
            raise ValueError(f"x must be (B, {IN_CH}), got {tuple(x.shape)}")
        out = torch.empty(x.shape[0], 1 if sigma_only else 4, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().nsr_mlp_forward(_p(self.packed), self._prec, _p(x), x.shape[0], int(bool(sigma_only)),
                                               _p(out), _stream()), "nsr_mlp_forward")
        return out

    __call__ = forward


class GenericMLP:
    """``VanillaMLP`` (models/networks.py:121-226) for the architecture flags the fused kernels are not laid out for:
    ``--D --W --skips`` (:124-126), ``--deg_pos --deg_dir --dim_rgb`` (models/nerf_model.py:53-57), with ``--no_dir`` and
    ``--color_activation none`` as there.  Every ``nn.Linear`` (+ activation) is one launch of the training step's fp32-MFMA
    GEMM (``nsr_linear``, include/nsr_train.h) over zero-padded operands -- K and N rounded up to multiples of 32 with
    zero weight columns / rows, whose products are exact zeros -- writing straight into the next layer's input buffer, so
    ``cat([input_xyz, h])`` of a skip layer and ``cat([final, dir])`` are column ranges of one buffer, never copies of
    activations.  Same interface as ``VanillaMLP`` (``forward(x, sigma_only)``, ``state_dict``, ``check``); inference only
    (the training step is built for the default architecture)."""

    POINT_CHUNK = 262144          # rows per pass (the reference's point_chunk, options/base_options.py:70): bounds the buffers

    def __init__(self, opt=None, device="cuda", precision: str = "fp32", **arch):
        self.arch = {**arch_of(opt), **arch}
        self.spec = arch_spec(**self.arch)                  # validates
        self.color_activation = getattr(opt, "color_activation", "sigmoid") if opt is not None else "sigmoid"
        if self.color_activation not in ("sigmoid", "none"):
            raise ValueError(f"color_activation={self.color_activation!r}: 'sigmoid' or 'none' (models/networks.py:173-180)")
        # --gamma_correct is not a property of the network in the reference either: render_rays raises the colours to 1 / 2.2
        # after the network call (models/nerf_downX_model.py:271-276).  The fused kernels carry it as an option of their
        # colour head; on this route NeRFDownXModel.render_rays applies it to what forward() returns.
        self.gamma_correct = bool(getattr(opt, "gamma_correct", False)) if opt is not None else False
        self.device = torch.device(device)
        _check_device(self.device, "GenericMLP device")
        a = self.arch
        self.in_xyz, self.in_dir = 3 + 6 * a["deg_pos"], 3 + 6 * a["deg_dir"]
        r32 = lambda n: (n + 31) // 32 * 32
        self.Kx, self.Wp, self.Hp, self.Dp, self.Rp = r32(self.in_xyz), r32(a["W"]), r32(a["W"] // 2), r32(self.in_dir), r32(a["dim_rgb"])
        # "fp32": every nn.Linear on the fp32 MFMA (nsr_linear); "f16x3" (round 6): on the split-fp16 MFMA, three terms per
        # product, products exact to ~2^-21 (nsr_linear_f16x3: the weights are split once at load time) -- the arithmetic of
        # the fused kernels' default precision, layer by layer.  The single-operand fast precisions have no layer-by-layer form.
        if precision not in ("fp32", "f16x3"):
            raise ValueError(f"GenericMLP precision {precision!r}: 'fp32' or 'f16x3' (the layer-by-layer route has no single-operand fast path)")
        self.precision = precision
        self._sd = None
        self._bad = None
        self._split = {}

    # -- weights ------------------------------------------------------------------------------------------------------
    def _padded(self, w, rows, col_map, cols):
        """(rows, cols) zero matrix with w's column ranges placed by col_map = [(src0, n, dst0), ...]."""
        out = torch.zeros(rows, cols, dtype=torch.float32, device=self.device)
        for s0, n, d0 in col_map:
            out[: w.shape[0], d0:d0 + n] = w[:, s0:s0 + n]
        return out

    def _padded_bias(self, b, n):
        out = torch.zeros(n, dtype=torch.float32, device=self.device)
        out[: b.shape[0]] = b
        return out

    def load_state_dict(self, sd):
        check_state_dict_arch(sd, **self.arch)
        a, dev = self.arch, {}
        for k in self.spec:
            v = sd[k]
            v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach()
            dev[k] = v.to(device=self.device, dtype=torch.float32).contiguous()
        W, ix, Kx, Wp = a["W"], self.in_xyz, self.Kx, self.Wp
        self._layers = []                       # (weight, bias, first input column, K): the trunk, inputs = slots [pe | h]
        for i in range(a["D"]):
            w, b = dev[f"xyz_encoding_{i + 1}.0.weight"], self._padded_bias(dev[f"xyz_encoding_{i + 1}.0.bias"], Wp)
            if i == 0:
                self._layers.append((self._padded(w, Wp, [(0, ix, 0)], Kx), b, 0, Kx))
            elif i in a["skips"]:               # cat([input_xyz, h]): pe at columns 0.., h at columns Kx..
                self._layers.append((self._padded(w, Wp, [(0, ix, 0), (ix, W, Kx)], Kx + Wp), b, 0, Kx + Wp))
            else:
                self._layers.append((self._padded(w, Wp, [(0, W, 0)], Wp), b, Kx, Wp))
        self._sigma = (self._padded(dev["sigma.weight"], 32, [(0, W, 0)], Wp), self._padded_bias(dev["sigma.bias"], 32))
        self._final = (self._padded(dev["xyz_encoding_final.weight"], Wp, [(0, W, 0)], Wp), self._padded_bias(dev["xyz_encoding_final.bias"], Wp))
        dmap = [(0, W, 0)] + ([] if a["no_dir"] else [(W, self.in_dir, Wp)])
        self._dir = (self._padded(dev["dir_encoding.0.weight"], self.Hp, dmap, Wp + (0 if a["no_dir"] else self.Dp)),
                     self._padded_bias(dev["dir_encoding.0.bias"], self.Hp))
        self._rgb = (self._padded(dev["rgb.0.weight"], self.Rp, [(0, W // 2, 0)], self.Hp), self._padded_bias(dev["rgb.0.bias"], self.Rp))
        self._sd = dev
        self._bad = torch.zeros((), dtype=torch.bool, device=self.device)
        self._split = {}
        if self.precision == "f16x3":       # (hi, lo) fp16 halves of 64 w for every padded weight, made once
            lib = _lib.load()
            for w, *_ in list(self._layers) + [self._sigma, self._final, self._dir, self._rgb]:
                hl = torch.empty(2, w.numel(), dtype=torch.int16, device=self.device)
                _lib.check(lib.nsr_split_weights(_p(w), w.numel(), c_void_p(hl[0].data_ptr()), c_void_p(hl[1].data_ptr()), _stream()),
                           "nsr_split_weights")
                self._split[w.data_ptr()] = hl
        return self

    def state_dict(self):
        if self._sd is None:
            raise RuntimeError("no weights loaded")
        return {k: v.clone() for k, v in self._sd.items()}

    # -- numerics status: the reference drops into pdb on NaN colours (models/nerf_downX_model.py:273-274) ---------------------------
    def status(self, clear: bool = False) -> int:
        flags = _lib.NSR_FLAG_OUTPUT_NONFINITE if (self._bad is not None and bool(self._bad)) else 0
        if clear and self._bad is not None:
            self._bad.zero_()
        return flags

    def check(self, what: str = "network"):
        flags = self.status(clear=True)
        if flags:
            raise _lib.NsrNumericsError(f"{what} (layer-by-layer fp32): non-finite network output", flags)
        return self

    # -- forward ------------------------------------------------------------------------------------------------------
    def _gemm(self, x, col0, ldx, K, w, b, act, y, ycol0, ldy, P):
        lib, esz = _lib.load(), 4
        hl = self._split.get(w.data_ptr())
        if hl is not None:
            _lib.check(lib.nsr_linear_f16x3(c_void_p(x.data_ptr() + col0 * esz), ldx, c_void_p(hl[0].data_ptr()), c_void_p(hl[1].data_ptr()),
                                            w.shape[1], _p(b), act, c_void_p(y.data_ptr() + ycol0 * esz), ldy, P, K, w.shape[0], _stream()),
                       "nsr_linear_f16x3")
            return
        _lib.check(lib.nsr_linear(c_void_p(x.data_ptr() + col0 * esz), ldx, _p(w), w.shape[1], _p(b), act,
                                  c_void_p(y.data_ptr() + ycol0 * esz), ldy, c_void_p(0), 0, P, K, w.shape[0], _stream()), "nsr_linear")

    def forward(self, x: torch.Tensor, sigma_only: bool = False) -> torch.Tensor:
        if self._sd is None:
            raise RuntimeError("GenericMLP.forward called before load_state_dict")
        x = _f32(x, "x")
        n_in = self.in_xyz + (0 if sigma_only else self.in_dir)
        if x.ndim != 2 or x.shape[1] < n_in:
            raise ValueError(f"x must be (B, {n_in}), got {tuple(x.shape)}")
        a, B = self.arch, x.shape[0]
        out = torch.empty(B, 1 if sigma_only else a["dim_rgb"] + 1, dtype=torch.float32, device=x.device)
        Kx, Wp, Ls = self.Kx, self.Wp, self.Kx + self.Wp
        for r0 in range(0, B, self.POINT_CHUNK):
            xb = x[r0:r0 + self.POINT_CHUNK]
            P = xb.shape[0]
            slots = [torch.zeros(P, Ls, dtype=torch.float32, device=x.device) for _ in range(2)]     # [pe | pad | h | pad] twice
            for s in slots:
                s[:, : self.in_xyz] = xb[:, : self.in_xyz]
            for i, (w, b, c0, K) in enumerate(self._layers):        # layer i reads slot (i - 1) % 2, writes h of slot i % 2
                self._gemm(slots[(i - 1) % 2], c0, Ls, K, w, b, 1, slots[i % 2], Kx, Ls, P)
            h = slots[(a["D"] - 1) % 2]
            small = torch.empty(P, 32, dtype=torch.float32, device=x.device)
            self._gemm(h, Kx, Ls, Wp, self._sigma[0], self._sigma[1], 0, small, 0, 32, P)
            if sigma_only:
                out[r0:r0 + P, 0] = small[:, 0]
                continue
            Ld = Wp + (0 if a["no_dir"] else self.Dp)
            dbuf = torch.zeros(P, Ld, dtype=torch.float32, device=x.device)                             # [final | pad | dir pe | pad]
            self._gemm(h, Kx, Ls, Wp, self._final[0], self._final[1], 0, dbuf, 0, Ld, P)
            if not a["no_dir"]:
                dbuf[:, Wp:Wp + self.in_dir] = xb[:, self.in_xyz:self.in_xyz + self.in_dir]
            c = torch.empty(P, self.Hp, dtype=torch.float32, device=x.device)
            self._gemm(dbuf, 0, Ld, Ld, self._dir[0], self._dir[1], 1, c, 0, self.Hp, P)
            rgb = torch.empty(P, self.Rp, dtype=torch.float32, device=x.device)
            self._gemm(c, 0, self.Hp, self.Hp, self._rgb[0], self._rgb[1], 2 if self.color_activation == "sigmoid" else 0, rgb, 0, self.Rp, P)
            out[r0:r0 + P, : a["dim_rgb"]] = rgb[:, : a["dim_rgb"]]
            out[r0:r0 + P, a["dim_rgb"]] = small[:, 0]
        self._bad |= ~torch.isfinite(out).all()
        return out

    __call__ = forward


def make_mlp(opt=None, precision: str = "fp32", device="cuda"):
    """The network class for an options object: the fused-kernel ``VanillaMLP`` for the architecture every script of the
    reference uses, ``GenericMLP`` for other ``--D --W --skips`` / degrees / ``dim_rgb``."""
    if is_default_arch(arch_of(opt)):
        return VanillaMLP(opt, precision=precision, device=device)
    warn_generic_mlp(arch_of(opt), precision)
    return GenericMLP(opt, device=device, precision="f16x3" if precision == "f16x3" else "fp32")


_GENERIC_WARNED = set()


def warn_generic_mlp(arch: dict, precision: str = "fp32") -> None:
    """One warning per architecture and process: a non-default ``--D --W --skips`` / encoding degree leaves the fused MFMA
    kernels (whose register, LDS and weight-stream layouts ARE the 8 x 256 network with a skip at layer 5) for the
    layer-by-layer fp32 GEMM route -- correct (tests/golden/arch.npz, made by the reference's own forward), but every
    activation matrix travels through HBM: `bench.py --arch D,W,skips` times it (README: ~an order of magnitude slower per
    sample point than the fused f16x3 kernel at the default size) -- on the fp32 MFMA, or under precision 'f16x3' on the split-fp16 MFMA (three terms per product, nsr_linear_f16x3)."""
    import warnings
    key = (arch.get("D"), arch.get("W"), tuple(arch.get("skips", ())), arch.get("deg_pos"), arch.get("deg_dir"))
    if key in _GENERIC_WARNED:
        return
    _GENERIC_WARNED.add(key)
    note = "" if precision in ("fp32", "f16x3") else f"; precision={precision!r} has no layer-by-layer form: running fp32"
    how = "split-fp16 (three-term)" if precision == "f16x3" else "fp32"
    warnings.warn(f"NeRF-SR network D={key[0]} W={key[1]} skips={list(key[2])} deg_pos={key[3]} deg_dir={key[4]} is not the "
                  f"architecture of the fused kernels (8 x 256, skip at 4, degrees 10 / 4): running the layer-by-layer {how} GEMM "
                  f"path (ops.GenericMLP), several times slower per sample point{note}", RuntimeWarning, stacklevel=3)


# ----------------------------------------------------------------------------- V1
def renderer_flags(white_bkgd: bool, sigma_activation: str = "relu") -> int:
    """The renderer option word of the compositing entry points (include/nsr.h): white background | softplus density."""
    if sigma_activation not in ("relu", "softplus"):
        raise ValueError(f"sigma_activation={sigma_activation!r}: 'relu' or 'softplus' (models/rendering.py:69-73)")
    return (_lib.NSR_WHITE_BKGD if white_bkgd else 0) | (_lib.NSR_SIGMA_SOFTPLUS if sigma_activation == "softplus" else 0)


class VolumetricRenderer:
    """Alpha compositing; mirrors models/rendering.py:66-111 (``opt.sigma_activation``: 'relu' or 'softplus' =
    ``log(1 + exp(sigma - 1))``, :69-73)."""

    def __init__(self, opt=None):
        self.sigma_activation = getattr(opt, "sigma_activation", "relu") if opt is not None else "relu"
        renderer_flags(False, self.sigma_activation)     # validates

    def forward(self, rgb, sigma, z_vals, white_bkgd: bool):
        rgb, sigma, z_vals = _f32(rgb, "rgb"), _f32(sigma, "sigma"), _f32(z_vals, "z_vals")
        R, N = z_vals.shape
        if rgb.shape != (R, N, 3) or sigma.shape != (R, N):
            raise ValueError("rgb must be (R,N,3) and sigma (R,N) matching z_vals (R,N)")
        dev = z_vals.device
        comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(R, dtype=torch.float32, device=dev)
        opac = torch.empty(R, dtype=torch.float32, device=dev)
        w = torch.empty(R, N, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().nsr_composite(_p(rgb), 3, _p(sigma), 1, _p(z_vals), R, N, renderer_flags(white_bkgd, self.sigma_activation),
                                             _p(comp), _p(depth), _p(opac), _p(w), _stream()), "nsr_composite")
        return comp, depth, opac, w

    __call__ = forward


# ----------------------------------------------------------------------------- D2 / D3
def render_rays(model: VanillaMLP, rays: torch.Tensor, z_vals: torch.Tensor):
    """Fused render_rays (models/nerf_downX_model.py:260-278): MLP at every sample of
    every ray, positional encodings computed in-kernel.  rays (R,8), z (R,N) ->
    ``(rgbs (R,N,3), sigmas (R,N))`` as views of one (R,N,4) buffer."""
    rays, z_vals = _f32(rays, "rays"), _f32(z_vals, "z_vals")
    R, N = z_vals.shape
    stride = _ray_stride(rays)
    raw = torch.empty(R, N, 4, dtype=torch.float32, device=rays.device)
    _lib.check(_lib.load().nsr_render_rays(_p(model.packed), model._prec, _p(rays), stride, _p(z_vals), R, N, _p(raw),
                                           _stream()), "nsr_render_rays")
    return raw[..., :3], raw[..., 3]


def render_rays_composited(model: VanillaMLP, rays: torch.Tensor, z_vals: torch.Tensor, white_bkgd: bool,
                           want_raw: bool = False, sigma_activation: str = "relu"):
    """``render_rays`` + ``VolumetricRenderer.forward`` in one launch (models/nerf_downX_model.py:289-291): 64 or 128
    samples per ray, fp32 / f16x3.  Returns ``(comp_rgb (R,3), depth (R), opacity (R), weights (R,N))`` and, if
    ``want_raw``, the (R, N, 4) network output as a fifth element.  Bit-identical to the two-call route."""
    rays, z_vals = _f32(rays, "rays"), _f32(z_vals, "z_vals")
    R, N = z_vals.shape
    dev = rays.device
    comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    opac = torch.empty(R, dtype=torch.float32, device=dev)
    w = torch.empty(R, N, dtype=torch.float32, device=dev)
    raw = torch.empty(R, N, 4, dtype=torch.float32, device=dev) if want_raw else None
    _lib.check(_lib.load().nsr_render_rays_composited(_p(model.packed), model._prec, _p(rays), _ray_stride(rays), _p(z_vals), R, N,
                                                      renderer_flags(white_bkgd, sigma_activation), _p(raw), _p(comp), _p(depth), _p(opac), _p(w),
                                                      _stream()), "nsr_render_rays_composited")
    return (comp, depth, opac, w, raw) if want_raw else (comp, depth, opac, w)


OUT_KEYS = ("coarse_comp_rgbs", "coarse_depth", "coarse_opacity", "coarse_weights",
            "fine_comp_rgbs", "fine_depth", "fine_opacity", "fine_weights")


def forward_rays(coarse: VanillaMLP, fine: Optional[VanillaMLP], rays: torch.Tensor, N_coarse: int = 64,
                 N_importance: int = 64, white_bkgd: bool = False, lindisp: bool = False,
                 workspace: Optional[torch.Tensor] = None, outs: Optional[Dict[str, torch.Tensor]] = None,
                 want_weights: bool = True, events=None, check: bool = False,
                 sigma_activation: str = "relu") -> Dict[str, torch.Tensor]:
    """Eval-mode forward_rays for the WHOLE batch in one enqueue sequence
    (models/nerf_downX_model.py:280-324; with 11-wide rays: the vanilla model's models/nerf_model.py:207-242):
    returns the reference's 8-entry dict.
    ``workspace`` / ``outs`` let a caller reuse buffers across frames; ``events`` (4 raw
    hipEvent_t handles from ``HipEvents``) brackets the coarse / fine MLP launches.
    ``check=True`` reads both networks' numerics status words after the enqueue (one stream wait) and raises
    ``NsrNumericsError`` on non-finite inputs / outputs or out-of-range activations -- the reference's NaN trap
    (models/nerf_downX_model.py:273-274) as an exception; ``False`` leaves the sticky flags for a later
    ``net.check()`` / ``net.status()`` (nothing is synchronised)."""
    lib = _lib.load()
    rays = _f32(rays, "rays")
    rays = rays.reshape(-1, rays.shape[-1])
    stride = _ray_stride(rays)
    R = rays.shape[0]
    if N_importance > 0 and fine is None:
        raise ValueError("N_importance > 0 needs the fine network")
    if fine is not None and fine._prec != coarse._prec:
        raise ValueError("coarse and fine networks must use the same precision")
    dev = rays.device
    need = lib.nsr_forward_rays_workspace_bytes_for(coarse._prec, R, N_coarse, N_importance)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    Nf = N_coarse + N_importance
    shapes = {"coarse_comp_rgbs": (R, 3), "coarse_depth": (R,), "coarse_opacity": (R,), "coarse_weights": (R, N_coarse),
              "fine_comp_rgbs": (R, 3), "fine_depth": (R,), "fine_opacity": (R,), "fine_weights": (R, Nf)}
    if outs is None:
        outs = {}
    keys = OUT_KEYS if N_importance > 0 else OUT_KEYS[:4]
    for k in keys:
        if k.endswith("weights") and not want_weights:
            continue
        if k not in outs or tuple(outs[k].shape) != shapes[k]:
            outs[k] = torch.empty(shapes[k], dtype=torch.float32, device=dev)
    ptrs = (c_void_p * 8)(*[_p(outs.get(k)) for k in OUT_KEYS])
    ev = (c_void_p * 4)(*events) if events is not None else None
    _lib.check(lib.nsr_forward_rays_profiled(_p(coarse.packed), _p(fine.packed) if fine is not None else c_void_p(0),
                                             coarse._prec, _p(rays), stride, R, N_coarse, N_importance,
                                             renderer_flags(white_bkgd, sigma_activation),
                                             int(bool(lindisp)), ptrs, _p(workspace), workspace.numel(), _stream(), ev),
               "nsr_forward_rays")
    if check:
        coarse.check("coarse network")
        if fine is not None and N_importance > 0:
            fine.check("fine network")
    return outs


class HipEvents:
    """n raw hipEvent_t handles (created through libnsr so they belong to its HIP runtime)."""

    def __init__(self, n: int):
        lib = _lib.load()
        self.handles = []
        for _ in range(n):
            h = c_void_p()
            _lib.check(lib.nsr_event_create(ctypes.byref(h)), "nsr_event_create")
            self.handles.append(h)

    def elapsed_ms(self, i: int, j: int) -> float:
        ms = c_float()
        _lib.check(_lib.load().nsr_event_elapsed_ms(self.handles[i], self.handles[j], ctypes.byref(ms)),
                   "nsr_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            lib = _lib.load()
            for h in self.handles:
                lib.nsr_event_destroy(h)
        except Exception:
            pass


# ----------------------------------------------------------------------------- A1 / A2
def sr_mean(hr: torch.Tensor, n_lr: int, s2: int) -> torch.Tensor:
    """``reshape(N_lr, s^2, c).mean(1)`` (models/nerf_downX_model.py:337-348)."""
    hr = _f32(hr, "hr")
    c = hr.numel() // (n_lr * s2)
    lr = torch.empty(n_lr, c, dtype=torch.float32, device=hr.device)
    _lib.check(_lib.load().nsr_sr_mean(_p(hr), n_lr, s2, c, _p(lr), _stream()), "nsr_sr_mean")
    return lr


def unflatten_reshape(x: torch.Tensor, img_wh: Tuple[int, int], downscale: int) -> torch.Tensor:
    """(N_lr*s^2, c) -> HR image (H, W, c) (models/nerf_downX_model.py:410-416)."""
    x = _f32(x, "x")
    W, H = int(img_wh[0]), int(img_wh[1])
    c = x.numel() // (H * W)
    out = torch.empty(H, W, c, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().nsr_unflatten(_p(x), H, W, int(downscale), c, _p(out), _stream()), "nsr_unflatten")
    return out
