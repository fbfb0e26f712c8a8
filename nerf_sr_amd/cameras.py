"""Synthetic camera poses for tests and benchmarks (numpy, host side).

No dataset can be loaded offline, so the configs of BASELINE.json are driven by
synthetic poses of the same *shape* as the reference's test trajectories:

* forward-facing "LLFF-like": a pose on the render spiral the reference builds
  with ``create_spiral_poses(radii, focus_depth)`` (``data/llff_downX_dataset.py:86-118``):
  camera centre ``(cos t, -sin t, -sin t/2) * radii`` looking at ``(0, 0, -focus)``.
* inward-facing "Blender-like": a pose on a sphere of radius 4 looking at the
  origin (``data/blender_downX_dataset.py`` uses the dataset's transforms; the
  360 trajectory of ``create_spheric_poses`` has the same geometry).

Both return ``c2w`` as a (3, 4) float32 matrix ``[x | y | z | centre]`` with the
camera looking down ``-z`` (OpenGL / NeRF convention used by ``get_rays``).
"""
from __future__ import annotations

import numpy as np


def _unit(v):
    return v / np.linalg.norm(v)


def look_at_pose(centre, target, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """c2w (3,4) for a camera at ``centre`` whose -z axis points at ``target``."""
    centre = np.asarray(centre, dtype=np.float64)
    z = _unit(centre - np.asarray(target, dtype=np.float64))   # camera +z points away from the target
    x = _unit(np.cross(np.asarray(up, dtype=np.float64), z))
    y = np.cross(z, x)
    return np.stack([x, y, z, centre], 1).astype(np.float32)


def spiral_pose(t: float, radii=(0.3, 0.3, 0.1), focus_depth: float = 3.5) -> np.ndarray:
    """One pose of the forward-facing render spiral at angle ``t`` (radians)."""
    centre = np.array([np.cos(t), -np.sin(t), -np.sin(0.5 * t)]) * np.asarray(radii, dtype=np.float64)
    return look_at_pose(centre, (0.0, 0.0, -focus_depth))


def spheric_pose(theta_deg: float, phi_deg: float = -30.0, radius: float = 4.0) -> np.ndarray:
    """Inward-facing pose on a sphere: azimuth ``theta``, elevation ``-phi``."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    centre = radius * np.array([np.cos(ph) * np.sin(th), -np.sin(ph), np.cos(ph) * np.cos(th)])
    return look_at_pose(centre, (0.0, 0.0, 0.0))


def llff_focal(width_px: int) -> float:
    """HR focal in pixels for a fern-like capture (COLMAP f ~ 3260.5 px at 4032 px wide)."""
    return 3260.5263 * width_px / 4032.0


def blender_focal(width_px: int, camera_angle_x: float = 0.6911112070083618) -> float:
    """``0.5*800/tan(0.5*angle) * W/800`` (``data/blender_downX_dataset.py:77-80``)."""
    return 0.5 * 800.0 / np.tan(0.5 * camera_angle_x) * width_px / 800.0
