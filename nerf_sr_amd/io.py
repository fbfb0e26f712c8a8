"""On-disk formats the reference reads: network checkpoints, COLMAP binary reconstructions (LLFF
scenes) and Blender ``transforms_*.json`` -- so that real checkpoints and scenes can be fed to the HIP path.

Host-side, numpy only (the reference's versions are host-side Python too):
  * checkpoints  ``{epoch}_net_{Coarse,Fine}.pth`` = ``torch.save(net.state_dict())`` (models/base_model.py:181-219)
  * COLMAP ``sparse/0/{cameras,images,points3D}.bin`` (utils/colmap.py:108-258; format: COLMAP's
    src/base/reconstruction.cc) and the LLFF pose pipeline built on them (data/llff_downX_dataset.py read_meta,
    identical in warp.py:35-92): focal rescale, camera-to-world, near/far bounds from the visible points,
    'right down front' -> 'right up back', centring on the average pose, rescaling so the nearest depth is 1 / 0.75
  * Blender ``transforms_{split}.json`` (data/blender_downX_dataset.py:60-95): focal from ``camera_angle_x``, 4x4 poses
  * LR training targets (data/llff_downX_dataset.py:312-329): Pillow's 8-bit LANCZOS resize to the HR and LR sizes,
    ``/ 255`` and the sub-pixel regroup -- these run ON THE DEVICE (include/nsr_image.h), bit-identical to Pillow;
    decoding the image file itself (PNG / JPEG -> uint8 array) stays with the caller (PIL, like the reference)
"""
from __future__ import annotations

import json
import os
import struct
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from .weights import STATE_DICT_SPEC

# ------------------------------------------------------------------------------------------------ checkpoints


def checkpoint_paths(checkpoints_dir: str, exp_name: str, epoch) -> Tuple[str, str]:
    """``'%s_net_%s.pth' % (epoch, name)`` for name in ('Coarse', 'Fine') (models/base_model.py:189, 206)."""
    d = os.path.join(checkpoints_dir, exp_name)
    return os.path.join(d, f"{epoch}_net_Coarse.pth"), os.path.join(d, f"{epoch}_net_Fine.pth")


def load_network_state(path: str, spec=None) -> "OrderedDict[str, np.ndarray]":
    """One ``.pth`` of the reference -> state dict of float32 numpy arrays in ``spec`` order (default: the NeRF MLP's
    24 tensors).  A ``module.`` prefix (a checkpoint saved from a DataParallel wrapper) is stripped; integer buffers
    (``num_batches_tracked``) are dropped."""
    import torch
    raw = torch.load(path, map_location="cpu", weights_only=True)
    sd = {}
    for k, v in raw.items():
        k = k[7:] if k.startswith("module.") else k
        if v.dtype.is_floating_point:
            sd[k] = v.detach().to(torch.float32).numpy()
    spec = STATE_DICT_SPEC if spec is None else spec
    missing = [k for k in spec if k not in sd]
    if missing:
        raise KeyError(f"{path}: missing {missing[:3]}{'...' if len(missing) > 3 else ''}")
    out = OrderedDict((k, np.ascontiguousarray(sd[k])) for k in spec)
    for k, shape in spec.items():
        if tuple(out[k].shape) != tuple(shape):
            raise ValueError(f"{path}: {k} has shape {out[k].shape}, expected {tuple(shape)}")
    return out


def save_network_state(sd: Dict[str, np.ndarray], path: str) -> None:
    """Write a state dict the way the reference does (``torch.save(net.cpu().state_dict(), path)``)."""
    import torch
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items()), path)


# ------------------------------------------------------------------------------------------------ COLMAP binaries

#: COLMAP camera models: id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


@dataclass
class Camera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


@dataclass
class ImageRec:
    id: int
    qvec: np.ndarray      # (w, x, y, z), world-to-camera rotation
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray       # (n, 2)
    point3D_ids: np.ndarray

    def qvec2rotmat(self) -> np.ndarray:
        w, x, y, z = self.qvec
        return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                         [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                         [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


@dataclass
class Point3D:
    id: int
    xyz: np.ndarray
    rgb: np.ndarray
    error: float
    image_ids: np.ndarray
    point2D_idxs: np.ndarray


def _rd(f, fmt: str):
    return struct.unpack("<" + fmt, f.read(struct.calcsize("<" + fmt)))


def read_cameras_binary(path: str) -> Dict[int, Camera]:
    cams = {}
    with open(path, "rb") as f:
        for _ in range(_rd(f, "Q")[0]):
            cid, model_id, w, h = _rd(f, "iiQQ")
            name, n = CAMERA_MODELS[model_id]
            cams[cid] = Camera(cid, name, w, h, np.array(_rd(f, "d" * n)))
    return cams


def read_images_binary(path: str) -> Dict[int, ImageRec]:
    imgs = {}
    with open(path, "rb") as f:
        for _ in range(_rd(f, "Q")[0]):
            v = _rd(f, "idddddddi")
            name = b""
            while True:
                ch = f.read(1)
                if ch == b"\x00":
                    break
                name += ch
            n2d = _rd(f, "Q")[0]
            pts = np.array(_rd(f, "ddq" * n2d)).reshape(-1, 3) if n2d else np.zeros((0, 3))
            imgs[v[0]] = ImageRec(v[0], np.array(v[1:5]), np.array(v[5:8]), v[8], name.decode("utf-8"),
                                  pts[:, :2].copy(), pts[:, 2].astype(np.int64))
    return imgs


def read_points3d_binary(path: str) -> Dict[int, Point3D]:
    pts = {}
    with open(path, "rb") as f:
        for _ in range(_rd(f, "Q")[0]):
            v = _rd(f, "QdddBBBd")
            n = _rd(f, "Q")[0]
            tr = np.array(_rd(f, "ii" * n), dtype=np.int64).reshape(-1, 2) if n else np.zeros((0, 2), np.int64)
            pts[v[0]] = Point3D(v[0], np.array(v[1:4]), np.array(v[4:7]), v[7], tr[:, 0].copy(), tr[:, 1].copy())
    return pts


# ------------------------------------------------------------------------------------------------ LLFF pose pipeline


def _unit(v: np.ndarray) -> np.ndarray:
    return v / np.linalg.norm(v)


def average_pose(poses: np.ndarray) -> np.ndarray:
    """The (3, 4) frame the scene is re-centred on (data/llff_dataset.py:21-56): origin at the mean camera centre,
    z along the mean viewing axis, x perpendicular to z and to the mean up vector, y completing the right-handed
    frame -- an orthonormal basis by construction."""
    mean_up, mean_back = poses[:, :, 1].mean(0), poses[:, :, 2].mean(0)
    z = _unit(mean_back)
    x = _unit(np.cross(mean_up, z))
    return np.column_stack([x, np.cross(z, x), z, poses[:, :, 3].mean(0)])


def center_poses(poses: np.ndarray):
    """Express every pose in the average pose's frame (data/llff_dataset.py:59-83).  The frame is a rigid transform
    [R | t] with orthonormal R, so its inverse is [R^T | -R^T t]: no 4 x 4 homogeneous matrices are built or
    inverted (the reference's np.linalg.inv route agrees to ~1e-16)."""
    frame = average_pose(poses)
    rot_t, origin = frame[:, :3].T, frame[:, 3]
    centred = np.empty_like(poses, dtype=np.float64)
    centred[:, :, :3] = rot_t @ poses[:, :, :3]
    centred[:, :, 3] = (poses[:, :, 3] - origin) @ rot_t.T
    return centred, frame


def _visible_depth_bounds(centres: np.ndarray, view_axes: np.ndarray, pts3d: Dict[int, Point3D]) -> np.ndarray:
    """Per camera: the 0.1 / 99.9 percentiles of the depth (along the camera's z axis) of the scene points COLMAP
    saw in that image (read_meta, "Step 2: read bounds").  Like the reference, row ``image_id - 1`` of the visibility
    table is taken to be the camera at position ``image_id - 1`` of the images file."""
    ids = list(pts3d)
    xyz = np.stack([pts3d[k].xyz for k in ids], 0)                                  # (P, 3)
    counts = np.array([len(pts3d[k].image_ids) for k in ids])
    seen = np.zeros((len(centres), len(ids)), dtype=bool)
    seen[np.concatenate([pts3d[k].image_ids for k in ids]).astype(np.int64) - 1, np.repeat(np.arange(len(ids)), counts)] = True
    depth = np.einsum("npc,nc->np", xyz[None] - centres[:, None], view_axes)        # (N, P)
    return np.array([np.percentile(depth[i, seen[i]], (0.1, 99.9)) for i in range(len(centres))])


def llff_scene_from_colmap(sparse_dir: str, img_w: int) -> Dict[str, object]:
    """``read_meta`` steps 1-3 of the LLFF datasets (data/llff_downX_dataset.py:196-249; the same code is in
    warp.py:35-92): returns focal (at width ``img_w``), image names (sorted), centred + rescaled camera-to-world
    poses (N, 3, 4) float64, per-image near/far bounds (N, 2), the index of the validation view and the scale factor."""
    cam = read_cameras_binary(os.path.join(sparse_dir, "cameras.bin"))[1]
    images = list(read_images_binary(os.path.join(sparse_dir, "images.bin")).values())      # file order
    # world-to-camera [R | t] of every image, inverted as a rigid transform: camera-to-world [R^T | -R^T t]
    rot = np.stack([im.qvec2rotmat() for im in images], 0)
    c2w = np.concatenate([rot.transpose(0, 2, 1), -np.einsum("nji,nj->ni", rot, np.stack([im.tvec for im in images]))[..., None]], 2)
    bounds = _visible_depth_bounds(c2w[:, :, 3], c2w[:, :, 2], read_points3d_binary(os.path.join(sparse_dir, "points3D.bin")))
    order = np.argsort([im.name for im in images])                                           # images sorted by file name
    c2w, bounds = c2w[order], bounds[order]
    c2w = c2w * np.array([1.0, -1.0, -1.0, 1.0])           # COLMAP "right down front" -> "right up back" (bmild/nerf#34)
    poses, _ = center_poses(c2w)
    scale = bounds.min() * 0.75                            # nearest depth ends up at 1 / 0.75 (kwea123/nerf_pl#50)
    poses[:, :, 3] /= scale
    return {"focal": float(cam.params[0] * img_w / cam.width), "names": [images[i].name for i in order], "poses": poses,
            "bounds": bounds / scale, "val_idx": int(np.argmin(np.linalg.norm(poses[:, :, 3], axis=1))),
            "scale_factor": float(scale)}


# ------------------------------------------------------------------------------------------------ Blender scenes


def load_blender_transforms(path: str, img_w: int) -> Dict[str, object]:
    """``transforms_{split}.json``: focal = 0.5 * 800 / tan(0.5 * camera_angle_x) * img_w / 800
    (data/blender_downX_dataset.py:77-80), poses = the first three rows of every frame's ``transform_matrix``."""
    with open(path) as f:
        meta = json.load(f)
    focal = 0.5 * 800 / np.tan(0.5 * meta["camera_angle_x"]) * img_w / 800
    poses = np.stack([np.array(fr["transform_matrix"], dtype=np.float64)[:3, :4] for fr in meta["frames"]], 0)
    return {"focal": float(focal), "poses": poses, "files": [fr["file_path"] for fr in meta["frames"]], "near": 2.0, "far": 6.0}


# ------------------------------------------------------------------------------------------------ LR targets (device)


def lanczos_tables(in_size: int, out_size: int):
    """Pillow's fixed-point LANCZOS weights for resampling ``in_size`` -> ``out_size`` samples, computed on the host in
    double precision by libnsr (``nsr_lanczos_coeffs``): ``bounds`` (out_size, 2) int32, ``kk`` (out_size, ksize) int32."""
    from . import _lib
    lib = _lib.load()
    ksize = lib.nsr_lanczos_ksize(int(in_size), int(out_size))
    if ksize <= 0:
        raise ValueError("sizes must be positive")
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    _lib.check(lib.nsr_lanczos_coeffs(int(in_size), int(out_size), bounds.ctypes.data, kk.ctypes.data), "nsr_lanczos_coeffs")
    return bounds, kk


def resize_lanczos_u8(img, out_wh):
    """``PIL.Image.resize(out_wh, Image.LANCZOS)`` of an (H, W, C) uint8 image ON THE DEVICE, bit-identical to Pillow
    (horizontal pass, then vertical pass; include/nsr_image.h).  ``img``: uint8 torch tensor on the GPU.
    C = 4 is an RGBA image: resampled premultiplied, like Pillow does (``convert("RGBa")`` -> resample ->
    ``convert("RGBA")``); C = 1 / 3: plain channels.  Other channel counts have no Pillow mode behind them."""
    import torch
    from . import _lib
    from .ops import _check_device, _p, _stream
    if img.dtype != torch.uint8 or img.ndim != 3:
        raise TypeError("img must be an (H, W, C) uint8 tensor")
    _check_device(img.device, "img")
    if img.shape[2] not in (1, 3, 4):
        raise ValueError(f"img must have 1 (L), 3 (RGB) or 4 (RGBA) channels, got {img.shape[2]}")
    img = img.contiguous()
    lib = _lib.load()
    w, h = int(out_wh[0]), int(out_wh[1])
    rgba = img.shape[2] == 4 and (w, h) != (img.shape[1], img.shape[0])
    if rgba:
        pre = torch.empty_like(img)
        _lib.check(lib.nsr_rgba_premultiply_u8(_p(img), img.shape[0] * img.shape[1], 0, _p(pre), _stream()), "nsr_rgba_premultiply_u8")
        img = pre
    for axis, out_size in ((1, w), (0, h)):
        H, W, C = img.shape
        in_size = W if axis == 1 else H
        if out_size == in_size:
            continue
        bounds, kk = lanczos_tables(in_size, out_size)
        b_dev, k_dev = torch.from_numpy(bounds).to(img.device), torch.from_numpy(kk).to(img.device)
        dst = torch.empty((H, out_size, C) if axis == 1 else (out_size, W, C), dtype=torch.uint8, device=img.device)
        _lib.check(lib.nsr_resample_pass_u8(_p(img), H, W, C, axis, out_size, _p(b_dev), _p(k_dev), kk.shape[1], _p(dst),
                                            _stream()), "nsr_resample_pass_u8")
        img = dst
    if rgba:
        _lib.check(lib.nsr_rgba_premultiply_u8(_p(img), img.shape[0] * img.shape[1], 1, _p(img), _stream()), "nsr_rgba_premultiply_u8")
    return img


def lr_targets(img_u8, img_wh, downscale: int):
    """What the downX datasets keep per training image (data/llff_downX_dataset.py:312-329, ``ds_method='lanc'``):
    the scene image -> HR ``img_wh`` -> LR ``img_wh / s`` (LANCZOS, 8 bit), then ``rgbs`` (N_lr, 3) = LR / 255 and
    ``rgbs_ori`` (N_lr, s*s, 3) = HR / 255 in the ray tensor's LR-pixel-major order.  Everything on the device.
    A 4-channel image is the Blender datasets' RGBA case (data/blender_downX_dataset.py:104-120): both resizes run on
    the RGBA image (premultiplied resampling) and the targets are blended onto white, ``rgb * a + (1 - a)``."""
    import torch
    from . import _lib
    from .ops import _p, _stream
    W, H = int(img_wh[0]), int(img_wh[1])
    s = int(downscale)
    if img_u8.ndim != 3 or img_u8.shape[2] not in (3, 4):
        raise ValueError(f"img_u8 must be (H, W, 3) RGB or (H, W, 4) RGBA, got {tuple(img_u8.shape)}")
    hr = resize_lanczos_u8(img_u8, (W, H))
    lr = resize_lanczos_u8(hr, (W // s, H // s))
    lib = _lib.load()
    rgbs = torch.empty((H // s) * (W // s), 3, dtype=torch.float32, device=hr.device)
    ori = torch.empty((H // s) * (W // s), s * s, 3, dtype=torch.float32, device=hr.device)
    fn, name = (lib.nsr_image_to_targets_rgba, "nsr_image_to_targets_rgba") if hr.shape[2] == 4 else \
        (lib.nsr_image_to_targets, "nsr_image_to_targets")
    _lib.check(fn(_p(lr), H // s, W // s, 1, _p(rgbs), _stream()), name)
    _lib.check(fn(_p(hr), H, W, s, _p(ori), _stream()), name)
    return rgbs, ori
