"""On-disk formats the reference reads: network checkpoints, COLMAP binary reconstructions (LLFF
scenes) and Blender ``transforms_*.json`` -- so that real checkpoints and scenes can be fed to the HIP path.

Host-side, numpy only (the reference's versions are host-side Python too):
  * checkpoints  ``{epoch}_net_{Coarse,Fine}.pth`` = ``torch.save(net.state_dict())`` (models/base_model.py:181-219)
  * COLMAP ``sparse/0/{cameras,images,points3D}.bin`` (utils/colmap.py:108-258; format: COLMAP's
    src/base/reconstruction.cc) and the LLFF pose pipeline built on them (data/llff_downX_dataset.py read_meta,
    identical in warp.py:35-92): focal rescale, camera-to-world, near/far bounds from the visible points,
    'right down front' -> 'right up back', centring on the average pose, rescaling so the nearest depth is 1 / 0.75
  * Blender ``transforms_{split}.json`` (data/blender_downX_dataset.py:60-95): focal from ``camera_angle_x``, 4x4 poses
"""
from __future__ import annotations

import json
import os
import struct
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .weights import STATE_DICT_SPEC, check_state_dict

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


def _normalize(v):
    return v / np.linalg.norm(v)


def average_pose(poses: np.ndarray) -> np.ndarray:
    """(3, 4) average pose: mean centre, z = mean z, x = y' cross z, y = z cross x (data/llff_dataset.py:21-56)."""
    center = poses[..., 3].mean(0)
    z = _normalize(poses[..., 2].mean(0))
    y_ = poses[..., 1].mean(0)
    x = _normalize(np.cross(y_, z))
    y = np.cross(z, x)
    return np.stack([x, y, z, center], 1)


def center_poses(poses: np.ndarray):
    """inv(average pose) @ poses (data/llff_dataset.py:59-83)."""
    avg = average_pose(poses)
    avg_h = np.eye(4)
    avg_h[:3] = avg
    last = np.tile(np.array([0, 0, 0, 1]), (len(poses), 1, 1))
    poses_h = np.concatenate([poses, last], 1)
    return (np.linalg.inv(avg_h) @ poses_h)[:, :3], avg


def llff_scene_from_colmap(sparse_dir: str, img_w: int) -> Dict[str, object]:
    """``read_meta`` steps 1-3 of the LLFF datasets (data/llff_downX_dataset.py; same code in warp.py:35-92):
    returns focal (at width ``img_w``), image names (sorted), centred + rescaled camera-to-world poses (N, 3, 4)
    float64, per-image near/far bounds (N, 2), the index of the validation view and the scale factor."""
    cam = read_cameras_binary(os.path.join(sparse_dir, "cameras.bin"))[1]
    focal = cam.params[0] * img_w / cam.width
    imdata = read_images_binary(os.path.join(sparse_dir, "images.bin"))
    perm = np.argsort([imdata[k].name for k in imdata])
    names = sorted(imdata[k].name for k in imdata)
    bottom = np.array([0, 0, 0, 1.0]).reshape(1, 4)
    w2c = np.stack([np.concatenate([np.concatenate([imdata[k].qvec2rotmat(), imdata[k].tvec.reshape(3, 1)], 1), bottom], 0)
                    for k in imdata], 0)
    poses = np.linalg.inv(w2c)[:, :3]
    pts3d = read_points3d_binary(os.path.join(sparse_dir, "points3D.bin"))
    pts_world = np.zeros((1, 3, len(pts3d)))
    vis = np.zeros((len(poses), len(pts3d)))
    for i, k in enumerate(pts3d):
        pts_world[0, :, i] = pts3d[k].xyz
        for j in pts3d[k].image_ids:
            vis[j - 1, i] = 1
    depths = ((pts_world - poses[..., 3:4]) * poses[..., 2:3]).sum(1)
    bounds = np.zeros((len(poses), 2))
    for i in range(len(poses)):
        zs = depths[i][vis[i] == 1]
        bounds[i] = [np.percentile(zs, 0.1), np.percentile(zs, 99.9)]
    poses, bounds = poses[perm], bounds[perm]
    poses = np.concatenate([poses[..., 0:1], -poses[..., 1:3], poses[..., 3:4]], -1)   # right down front -> right up back
    poses, _ = center_poses(poses)
    val_idx = int(np.argmin(np.linalg.norm(poses[..., 3], axis=1)))
    scale = bounds.min() * 0.75
    bounds = bounds / scale
    poses[..., 3] /= scale
    return {"focal": float(focal), "names": names, "poses": poses, "bounds": bounds, "val_idx": val_idx, "scale_factor": float(scale)}


# ------------------------------------------------------------------------------------------------ Blender scenes


def load_blender_transforms(path: str, img_w: int) -> Dict[str, object]:
    """``transforms_{split}.json``: focal = 0.5 * 800 / tan(0.5 * camera_angle_x) * img_w / 800
    (data/blender_downX_dataset.py:77-80), poses = the first three rows of every frame's ``transform_matrix``."""
    with open(path) as f:
        meta = json.load(f)
    focal = 0.5 * 800 / np.tan(0.5 * meta["camera_angle_x"]) * img_w / 800
    poses = np.stack([np.array(fr["transform_matrix"], dtype=np.float64)[:3, :4] for fr in meta["frames"]], 0)
    return {"focal": float(focal), "poses": poses, "files": [fr["file_path"] for fr in meta["frames"]], "near": 2.0, "far": 6.0}
