"""BASELINE config #5 composed on the device: render -> depth -> warp -> refinement network.

In the reference this is three separate programs glued by files:

1. ``test.py:52`` (``NeRFDownXModel.test``, models/nerf_downX_model.py:621-669) renders every view and writes the HR
   image and ``{i}-fine-depth-ori.npz`` = ``unflatten_reshape(out_fine_depth_ori)`` (``:447-448``,
   utils/visualizer.py:94-99);
2. ``warp.py:100-176`` reads those depth maps, warps every view into the reference view and writes ``{i}_locs.npz``;
3. ``RefineModel.test`` (models/refine_model.py:199-232) over ``LLFFRefineDataset`` (data/llff_refine_dataset.py:257-354)
   tiles the rendered view, gathers up to 8 reference patches per tile at the warped locations, runs
   ``MaxPoolingModel`` and stitches.

Here the same three stages run back to back on one stream with every intermediate resident in HBM (no files, no host
copies): ``NeRFDownXModel.render_image`` -> ``warp.depth_warp`` -> ``refine.refine_image``.  What the depth map holds
decides the warp's first line (include/nsr_warp.h): NDC depth for forward-facing scenes (the reference's only case),
distance along the unit-norm ray for Blender scenes (config #5; defined by this build).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import ops, refine, warp
from .model import NeRFDownXModel


def world_to_camera(c2w) -> np.ndarray:
    """float64 (3, 4) world-to-camera of a (3, 4) pose, as warp.py:107 builds ``ref_w2c`` from the float32 pose."""
    m = np.concatenate([np.asarray(c2w, dtype=np.float32), np.array([0, 0, 0, 1]).reshape(1, 4)], 0)
    return np.linalg.inv(m)[:3]


@torch.no_grad()
def render_warp_refine(model: NeRFDownXModel, net: refine.MaxPoolingModel, c2w, ref_c2w, ref_img: torch.Tensor,
                       focal: float, ndc: bool, near: float = 0.0, far: float = 1.0, patch_len: int = 64,
                       num_ref_patches: int = 8, batch: int = 32, events: Optional[list] = None,
                       depth_kind: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """One synthesised view through the whole of config #5.

    ``model``: the render path with its two networks loaded; ``net``: the refinement network; ``c2w``: pose of the
    view to synthesise; ``ref_c2w`` / ``ref_img`` (3, H, W) in [0, 1]: pose and HR image of the reference view (view 0
    of the scene in the reference).  ``events``: optional list of 4 ``torch.cuda.Event`` recorded before the render,
    after it, after the warp and after the refinement pass.
    ``depth_kind`` says what the rendered depth map holds (``warp.DEPTH_KINDS``): default ``'ndc'`` for NDC scenes and
    ``'ray'`` (distance along the unit-norm ray: this build's definition for Blender scenes) otherwise; pass ``'metric'``
    to use the depth as it is, which is what the reference's ``warp.py:120-126`` does for ``spheric_poses`` LLFF scenes
    (needed to reproduce its ``{i}_locs.npz`` there).
    Returns the HR render (H, W, 3), the HR depth map (H, W), ``locs`` (H, W, 3) float64 and the refined image
    (3, H, W) in [0, 1]."""
    def mark(i):
        if events is not None:
            events[i].record()
    mark(0)
    res = model.render_image(c2w, focal, ndc, near, far)
    depth_hw = model.unflatten_reshape(model.out_fine_depth_ori.reshape(-1, 1))[..., 0].contiguous()   # {i}-fine-depth-ori
    mark(1)
    if depth_kind is None:
        depth_kind = "ndc" if ndc else "ray"
    if depth_kind not in warp.DEPTH_KINDS:
        raise ValueError(f"depth_kind must be one of {list(warp.DEPTH_KINDS)}")
    locs = warp.depth_warp(depth_hw, c2w, world_to_camera(ref_c2w), focal, depth_kind)
    mark(2)
    # the refine dataset normalises images to [-1, 1] (T.Normalize(0.5, 0.5), data/llff_refine_dataset.py:252-255)
    sr = (res["hr_rgb"].permute(2, 0, 1) * 2.0 - 1.0).contiguous()
    refined = refine.refine_image(net, sr, (ops._f32(ref_img, "ref_img") * 2.0 - 1.0).contiguous(), locs, patch_len,
                                  num_ref_patches, batch)
    mark(3)
    return {"hr_rgb": res["hr_rgb"], "depth": depth_hw, "locs": locs, "refined": (refined + 1.0) * 0.5,
            "lr_rgb": res["lr_rgb"]}
