"""Host mirror of the reference's refinement network over ``include/nsr_refine.h``.

``MaxPoolingModel`` mirrors ``models/networks.py:945-990`` in eval mode: ``forward(x_synth, list_x_candi)`` with
``x_synth`` (B, 3, H, W) and ``list_x_candi`` (B, R, 3, H, W) in [-1, 1] -> refined patch (B, 3, H, W) (tanh).
Weights enter as the reference's ``state_dict`` (``load_state_dict``; the int64 ``num_batches_tracked`` entries are
ignored).  All arithmetic runs in libnsr.so; there is no CPU path.
"""
from __future__ import annotations

from collections import OrderedDict
from ctypes import c_void_p
from typing import Dict

import numpy as np
import torch

from . import _lib
from .ops import _f32, _p, _stream

# (name, cin, cout, stride, batch-norm name or None) in forward = state_dict order
LAYERS = [
    ("E.conv1", 3, 128, None), ("E.conv2", 128, 128, "E.conv2_bnorm"), ("E.conv3", 128, 256, "E.conv3_bnorm"),
    ("E.conv4", 256, 256, "E.conv4_bnorm"), ("E.conv5", 256, 512, "E.conv5_bnorm"), ("E.conv6", 512, 512, "E.conv6_bnorm"),
    ("E.conv7", 512, 512, "E.conv7_bnorm"),
    ("D.conv1", 1024, 512, "D.conv1_bnorm"), ("D.conv2", 512, 512, "D.conv2_bnorm"), ("D.conv2_up", 512, 512, "D.conv2_up_bnorm"),
    ("D.conv3", 1536, 512, "D.conv3_bnorm"), ("D.conv4", 512, 512, "D.conv4_bnorm"), ("D.conv4_up", 512, 256, "D.conv4_up_bnorm"),
    ("D.conv5", 768, 256, "D.conv5_bnorm"), ("D.conv6", 256, 256, "D.conv6_bnorm"), ("D.conv6_up", 256, 128, "D.conv6_up_bnorm"),
    ("D.conv7", 384, 128, "D.conv7_bnorm"), ("D.conv8", 128, 128, "D.conv8_bnorm"), ("D.conv9", 128, 3, None),
]


# --not_use_ref (Model_VNPCAT_Decoder_NoPooling, networks.py:866-945): the decoder's concatenations carry no F_max_i
# channels, so four convolutions are narrower
NOREF_CIN = {"D.conv1": 512, "D.conv3": 1024, "D.conv5": 512, "D.conv7": 256}


def _spec(not_use_ref: bool = False) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    for name, cin, cout, bn in LAYERS:
        if not_use_ref:
            cin = NOREF_CIN.get(name, cin)
        s[f"{name}.weight"] = (cout, cin, 3, 3)
        s[f"{name}.bias"] = (cout,)
        if bn:
            for k in ("weight", "bias", "running_mean", "running_var"):
                s[f"{bn}.{k}"] = (cout,)
    return s


#: the 106 float tensors of MaxPoolingModel.state_dict(), in order (and with --not_use_ref)
REFINE_SPEC = _spec()
REFINE_SPEC_NOREF = _spec(True)
assert len(REFINE_SPEC) == 106 and len(REFINE_SPEC_NOREF) == 106


def make_refine_state_dict(seed: int, not_use_ref: bool = False) -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights (no checkpoint can be downloaded): xavier-normal convolutions
    (``initialize_weight``, networks.py:776-783), BatchNorm weight ~ N(1, 0.02), and NON-trivial running statistics
    (mean ~ N(0, 0.1), var ~ U(0.5, 1.5)) and biases so that the folded affine map is exercised."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, shape in (REFINE_SPEC_NOREF if not_use_ref else REFINE_SPEC).items():
        if k.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif k.endswith("running_mean"):
            v = rng.normal(0.0, 0.1, shape)
        elif len(shape) == 4:
            fan_in, fan_out = shape[1] * 9, shape[0] * 9
            v = rng.normal(0.0, np.sqrt(2.0 / (fan_in + fan_out)), shape)
        elif "bnorm.weight" in k:
            v = rng.normal(1.0, 0.02, shape)
        else:
            v = rng.normal(0.0, 0.05, shape)
        sd[k] = v.astype(np.float32)
    return sd


def refine_macs(H: int = 64, W: int = 64, R: int = 8) -> int:
    """True convolution MACs of one forward on a (H, W) patch with R reference patches (networks.py:735-990):
    the encoder runs on 1 + R images, the decoder once; no padding or im2col overhead counted."""
    px = [H * W, H * W // 4, H * W // 16, H * W // 64]
    enc_level = [0, 0, 1, 1, 2, 2, 3]
    dec_level = [3, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0, 0]
    m = 0
    for i, (_, cin, cout, _bn) in enumerate(LAYERS[:7]):
        m += (1 + R) * px[enc_level[i]] * 9 * cin * cout
    for i, (_, cin, cout, _bn) in enumerate(LAYERS[7:]):
        m += px[dec_level[i]] * 9 * cin * cout
    return m


class MaxPoolingModel:
    """Encoder + max over the reference patches + decoder, eval mode (BatchNorm uses its running statistics).
    ``opt.not_use_ref`` (or ``not_use_ref=True``) selects the reference's no-pooling decoder: the encoder runs on the
    synthesised patch only and ``forward`` ignores ``list_x_candi`` (networks.py:958-969)."""

    def __init__(self, opt=None, precision: str = "f16x3", device="cuda", not_use_ref: bool = False):
        self.not_use_ref = bool(not_use_ref or (opt is not None and getattr(opt, "not_use_ref", False)))
        if precision not in ("fp32", "f16x3"):
            raise ValueError("precision must be 'fp32' or 'f16x3'")
        self.precision, self._prec = precision, _lib.PRECISIONS[precision]
        self.device = torch.device(device)
        self.spec = REFINE_SPEC_NOREF if self.not_use_ref else REFINE_SPEC
        lib = _lib.load()
        nbytes = (lib.nsr_refine_packed_bytes_noref if self.not_use_ref else lib.nsr_refine_packed_bytes)(self._prec)
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._loaded = False
        self._ws = None

    def load_state_dict(self, sd):
        missing = [k for k in self.spec if k not in sd]
        if missing:
            raise KeyError(f"state_dict lacks {missing[:3]}{'...' if len(missing) > 3 else ''}")
        dev = []
        for k, shape in self.spec.items():
            v = sd[k]
            v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach()
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"{k}: expected shape {shape}, got {tuple(v.shape)}")
            dev.append(v.to(device=self.device, dtype=torch.float32).contiguous())
        ptrs = (c_void_p * len(dev))(*[c_void_p(t.data_ptr()) for t in dev])
        pack = _lib.load().nsr_refine_pack_weights_noref if self.not_use_ref else _lib.load().nsr_refine_pack_weights
        _lib.check(pack(ptrs, _p(self.packed), self._prec, _stream()), "nsr_refine_pack_weights")
        torch.cuda.current_stream().synchronize()       # `dev` may be freed once the pack kernels have run
        self._loaded = True
        return self

    def eval(self):
        return self

    def forward(self, x_synth: torch.Tensor, list_x_candi: torch.Tensor = None) -> torch.Tensor:
        if not self._loaded:
            raise RuntimeError("MaxPoolingModel.forward called before load_state_dict")
        if self.not_use_ref:
            return self._forward_noref(_f32(x_synth, "x_synth"))
        x, c = _f32(x_synth, "x_synth"), _f32(list_x_candi, "list_x_candi")
        if x.ndim != 4 or x.shape[1] != 3 or c.ndim != 5 or c.shape[0] != x.shape[0] or tuple(c.shape[2:]) != tuple(x.shape[1:]):
            raise ValueError("expected x_synth (B, 3, H, W) and list_x_candi (B, R, 3, H, W)")
        B, _, H, W = x.shape
        R = c.shape[1]
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        lib = _lib.load()
        need = lib.nsr_refine_workspace_bytes_for(self._prec, B, R, H, W)
        if need == 0:
            raise ValueError("H and W must be positive multiples of 8 and R >= 1")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _lib.check(lib.nsr_refine_forward(_p(self.packed), self._prec, _p(x), _p(c), B, R, H, W, _p(out), _p(self._ws), self._ws.numel(),
                                          _stream()), "nsr_refine_forward")
        return out

    def _forward_noref(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim != 4 or x.shape[1] != 3:
            raise ValueError("expected x_synth (B, 3, H, W)")
        B, _, H, W = x.shape
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        lib = _lib.load()
        need = lib.nsr_refine_workspace_bytes_for(self._prec, B, 1, H, W)
        if need == 0:
            raise ValueError("H and W must be positive multiples of 8")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        _lib.check(lib.nsr_refine_forward_noref(_p(self.packed), self._prec, _p(x), B, H, W, _p(out), _p(self._ws), self._ws.numel(),
                                                _stream()), "nsr_refine_forward_noref")
        return out

    __call__ = forward


# ------------------------------------------------------------------------------------------------ tiler / stitcher
def tile_refs(locs: torch.Tensor, patch_len: int = 64, num_ref_patches: int = 8):
    """locs (H, W, 3) float64 on the GPU (``nerf_sr_amd.warp.depth_warp`` / ``{i}_locs.npz``) -> ``starts`` (n, 2)
    int32 [x, y] of the SR tiles and ``ref_starts`` (n, num_ref_patches, 2) int32 of their reference patches (-1: use
    the SR tile), as ``LLFFRefineDataset.__getitem__`` picks them (data/llff_refine_dataset.py:303-329)."""
    if not locs.is_cuda or locs.dtype != torch.float64 or locs.ndim != 3 or locs.shape[2] != 3:
        raise ValueError("locs must be a (H, W, 3) float64 tensor on the GPU")
    locs = locs.contiguous()
    H, W = locs.shape[:2]
    n = -(-W // patch_len) * -(-H // patch_len)
    starts = torch.empty(n, 2, dtype=torch.int32, device=locs.device)
    refs = torch.empty(n, num_ref_patches, 2, dtype=torch.int32, device=locs.device)
    _lib.check(_lib.load().nsr_refine_tile(_p(locs), H, W, patch_len, num_ref_patches, _p(starts), _p(refs), _stream()),
               "nsr_refine_tile")
    return starts, refs


def gather_patches(sr_img: torch.Tensor, ref_img: torch.Tensor, starts: torch.Tensor, ref_starts: torch.Tensor, patch_len: int = 64):
    """sr_img, ref_img (3, H, W) -> sr_patch (n, 3, p, p), ref_patches (n, R, 3, p, p) (the dataset's sample)."""
    sr_img, ref_img = _f32(sr_img, "sr_img"), _f32(ref_img, "ref_img")
    _, H, W = sr_img.shape
    n, R = ref_starts.shape[:2]
    sr = torch.empty(n, 3, patch_len, patch_len, dtype=torch.float32, device=sr_img.device)
    ref = torch.empty(n, R, 3, patch_len, patch_len, dtype=torch.float32, device=sr_img.device)
    _lib.check(_lib.load().nsr_refine_gather(_p(sr_img), _p(ref_img), H, W, patch_len, R, _p(starts.contiguous()),
                                             _p(ref_starts.contiguous()), n, _p(sr), _p(ref), _stream()), "nsr_refine_gather")
    return sr, ref


def stitch_patches(patches: torch.Tensor, starts: torch.Tensor, img_wh) -> torch.Tensor:
    """Paste predictions back in tile order, later tiles overwrite earlier ones (models/refine_model.py:211-214)."""
    patches = _f32(patches, "patches")
    W, H = int(img_wh[0]), int(img_wh[1])
    img = torch.empty(3, H, W, dtype=torch.float32, device=patches.device)
    _lib.check(_lib.load().nsr_refine_stitch(_p(patches), _p(starts.contiguous()), patches.shape[0], patches.shape[-1], H, W,
                                             _p(img), _stream()), "nsr_refine_stitch")
    return img


def refine_image(net: MaxPoolingModel, sr_img: torch.Tensor, ref_img: torch.Tensor, locs: torch.Tensor, patch_len: int = 64,
                 num_ref_patches: int = 8, batch: int = 256) -> torch.Tensor:
    """The refinement pass over one synthesised view (config #5 tail): tile -> network on batches of tiles -> stitch.
    Images are (3, H, W) in [-1, 1] (the dataset's Normalize(0.5, 0.5)); returns the refined (3, H, W) image.
    ``batch``: tiles per network call.  The default takes the 169 tiles of an 800 x 800 frame in ONE call (3.5 GB of
    activations): the 8 x 8-pixel decoder layers of a 32-tile batch fill an eighth of the chip (57 vs 45 ms per frame)."""
    starts, refs = tile_refs(locs, patch_len, num_ref_patches)
    sr, ref = gather_patches(sr_img, ref_img, starts, refs, patch_len)
    pred = torch.cat([net(sr[i:i + batch], ref[i:i + batch]) for i in range(0, sr.shape[0], batch)], 0)
    return stitch_patches(pred, starts, (sr_img.shape[2], sr_img.shape[1]))
