"""Host mirror of the reference's depth warp (``warp.py:100-176``) over ``include/nsr_warp.h``.

The reference's script loops over a scene's images on the host and runs a Python double loop over the pixels; here
one kernel launch per image does the per-pixel stage on the device.  Pose handling (COLMAP -> centred, rescaled
poses, ``warp.py:35-92``) stays with the caller: pass the view's float32 ``c2w`` and the reference view's float64
``ref_w2c`` exactly as the script computes them (``:104-107``)."""
from __future__ import annotations

from ctypes import c_double, c_float
from typing import Optional

import numpy as np
import torch

from . import _lib
from .ops import _f32, _p, _stream


DEPTH_KINDS = {"metric": 0, "ndc": 1, "ray": 2}


def depth_warp(depth: torch.Tensor, c2w, ref_w2c, focal: float, ndc=True,
               ref_rgb: Optional[torch.Tensor] = None):
    """depth (H, W) fp32 on the GPU (``{i}-fine-depth-ori``); c2w (3, 4); ref_w2c (3, 4).  ``ndc`` says what the depth
    map holds: True / ``'ndc'`` = NDC depth of an LLFF scene (``warp.py:118``), False / ``'metric'`` = depth along the
    camera axis, used as it is (the reference's ``spheric_poses`` branch, ``warp.py:120-126``), ``'ray'`` = distance
    along a unit-norm ray (what the render path yields for Blender scenes: config #5; defined by this build, see
    include/nsr_warp.h).  Returns ``locs`` (H, W, 3) float64 -- the content of ``{i}_locs.npz`` -- and, if
    ``ref_rgb`` (3, H, W) is given, the warped image (3, H, W) (``{i}-wrapped.png`` before quantisation)."""
    kind = DEPTH_KINDS[ndc] if isinstance(ndc, str) else int(bool(ndc))
    depth = _f32(depth, "depth")
    if depth.ndim != 2:
        raise ValueError("depth must be (H, W)")
    H, W = depth.shape
    c = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(12))
    r = np.ascontiguousarray(np.asarray(ref_w2c, dtype=np.float64).reshape(12))
    locs = torch.empty(H, W, 3, dtype=torch.float64, device=depth.device)
    warped = None
    if ref_rgb is not None:
        ref_rgb = _f32(ref_rgb, "ref_rgb")
        if tuple(ref_rgb.shape) != (3, H, W):
            raise ValueError("ref_rgb must be (3, H, W)")
        warped = torch.empty(3, H, W, dtype=torch.float32, device=depth.device)
    _lib.check(_lib.load().nsr_depth_warp(_p(depth), H, W, float(focal), (c_float * 12)(*c.tolist()),
                                          (c_double * 12)(*r.tolist()), kind, _p(ref_rgb), _p(locs), _p(warped),
                                          _stream()), "nsr_depth_warp")
    return (locs, warped) if ref_rgb is not None else locs
