#!/usr/bin/env python3
"""Golden vectors of the reference's depth warp (``warp.py:100-176``).

Runs ONLY in the development container.  ``warp.py`` is a script: it defines ``LLFFDataset`` whose constructor
reads a COLMAP reconstruction, the scene's images and the ``{i}-fine-depth-ori.npz`` files the test loop wrote,
warps every image into the first one's camera and saves ``{i}_locs.npz`` / ``{i}-wrapped.png``; then it runs that
on hard-coded absolute paths.  This harness executes the reference's OWN class definition (the text of the file up
to its driver lines is compiled at run time, never stored) on a small synthetic scene:

  * ``utils.colmap.read_*_binary`` are replaced by in-memory fakes describing 3 cameras + 64 scene points, so the
    reference's pose pipeline (COLMAP -> centre -> rescale, ``warp.py:35-92``) runs unchanged on them;
  * images are random PNGs in a temp dir, depths are synthetic NDC depth maps saved the way
    ``utils/visualizer.py:94-99`` saves them;
  * ``torchvision.transforms.ToTensor / ToPILImage`` (absent here) are minimal stand-ins; ToPILImage records the
    warped tensor it is handed, which is what the fixture stores.

A second pass takes the reference's OTHER depth branch (``warp.py:120-126``, ``spheric_poses = True``: the rendered
depth is used as it is, no NDC -> metric conversion) by flipping that attribute in a subclass hook that runs between
the constructor's ``self.spheric_poses = False`` and ``read_meta()``; its depth maps are metric (camera-axis depth
2.5 .. 5).  This is the arithmetic BASELINE config #5 (Blender, near / far 2 / 6) goes through -> ``warp_spheric.npz``.

Fixtures ``warp_llff.npz`` / ``warp_spheric.npz``: per image the inputs of the per-pixel stage (depth map, the float32 pose, the float64
reference world-to-camera matrix, focal) and its outputs (``locs`` (H, W, 3) float64, warped image).  Note the
arithmetic types are those NumPy >= 2 promotion gives ``warp.py:128-131`` (float32 grid / float64 focal -> float64).
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


class ToTensor:
    def __call__(self, img):
        a = np.asarray(img, dtype=np.uint8)
        return torch.from_numpy(a.astype(np.float32) / 255.0).permute(2, 0, 1).contiguous()


class ToPILImage:
    recorded = []

    def __call__(self, t):
        ToPILImage.recorded.append(t.detach().clone())
        a = (t.detach().clamp(0, 1) * 255).byte().permute(1, 2, 0).numpy()
        return Image.fromarray(a)


def look_at_colmap(centre, target):
    """camera-to-world in COLMAP's 'right down front' convention."""
    f = target - centre
    f = f / np.linalg.norm(f)
    r = np.cross(f, np.array([0.0, 1.0, 0.0]))
    r = r / np.linalg.norm(r)
    d = np.cross(f, r)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, d, f, centre
    return c2w


def main():
    mg.install_shim()
    import torchvision.transforms as T
    T.ToTensor, T.ToPILImage = ToTensor, ToPILImage
    run(False, "warp_llff.npz", 7)
    run(True, "warp_spheric.npz", 11)


def run(spheric: bool, fixture: str, seed: int):
    ToPILImage.recorded = []
    rng = np.random.default_rng(seed)
    W, H, n_img = 24, 18, 3
    root = tempfile.mkdtemp(prefix="nsr_warp_root_")
    result = tempfile.mkdtemp(prefix="nsr_warp_res_")
    os.makedirs(os.path.join(root, "images"))
    names = [f"img_{i:02d}.png" for i in range(n_img)]
    for n in names:
        Image.fromarray(rng.integers(0, 256, (H * 2, W * 2, 3), dtype=np.uint8)).save(os.path.join(root, "images", n))
    # scene: points around the origin, cameras on a small arc in front of it
    pts = rng.normal(0, 0.6, (64, 3)) + np.array([0.0, 0.0, 4.0])
    centres = [np.array([0.35 * (i - 1), 0.1 * i, -0.2 * i]) for i in range(n_img)]
    c2ws = [look_at_colmap(c, np.array([0.0, 0.0, 4.0])) for c in centres]

    class FakeImage:
        def __init__(self, name, c2w):
            w2c = np.linalg.inv(c2w)
            self.name, self._R, self.tvec = name, w2c[:3, :3], w2c[:3, 3]

        def qvec2rotmat(self):
            return self._R

    cam = types.SimpleNamespace(height=H * 2, width=W * 2, params=np.array([2.1 * W * 2, W, H, 0.0]))
    imdata = {i + 1: FakeImage(names[i], c2ws[i]) for i in range(n_img)}
    pts3d = {k: types.SimpleNamespace(xyz=pts[k], image_ids=[1, 2, 3]) for k in range(len(pts))}
    import utils.colmap as uc
    uc.read_cameras_binary = lambda p: {1: cam}
    uc.read_images_binary = lambda p: imdata
    uc.read_points3d_binary = lambda p: pts3d
    depths = []
    for i in range(n_img):
        yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
        d = 0.45 + 0.3 * np.sin(3 * xx + i) * np.cos(2 * yy) + 0.05 * rng.random((H, W))
        if spheric:
            d = 2.5 + 3.0 * d                                    # metric depth along the camera axis
        d = d.astype(np.float32)[..., None]                     # (H, W, 1) like out_fine_depth_ori reshaped by the visualiser
        np.savez(os.path.join(result, f"{i}-fine-depth-ori.npz"), d)
        depths.append(d[..., 0])
    # the reference's own class, compiled from its file at run time (nothing of it is stored)
    src = open(os.path.join(mg.REF, "warp.py")).read()
    cut = src.index("\nwidth = ")
    ns = {"__name__": "reference_warp"}
    cwd = os.getcwd()
    os.chdir(mg.REF)
    try:
        exec(compile(src[:cut], os.path.join(mg.REF, "warp.py"), "exec"), ns)
    finally:
        os.chdir(cwd)
    cls = ns["LLFFDataset"]
    if spheric:
        class SphericPoses(cls):
            def define_transforms(self):      # runs after `self.spheric_poses = False`, before read_meta()
                super().define_transforms()
                self.spheric_poses = True
        cls = SphericPoses
    ds = cls(root, result, W, H)
    assert bool(ds.spheric_poses) == spheric
    out = {"W": W, "H": H, "n_img": n_img, "focal": np.float64(ds.focal), "ref_w2c": np.asarray(ds.ref_w2c, np.float64),
           "ref_rgbs": mg.np32(ds.ref_rgbs)}
    assert len(ToPILImage.recorded) == n_img
    for i in range(n_img):
        out[f"depth_{i}"] = depths[i]
        out[f"c2w_{i}"] = torch.FloatTensor(ds.poses[i]).numpy()
        out[f"locs_{i}"] = np.load(os.path.join(result, f"{i}_locs.npz"))["arr_0"]
        out[f"warped_{i}"] = mg.np32(ToPILImage.recorded[i])
        assert out[f"locs_{i}"].dtype == np.float64 and out[f"locs_{i}"].shape == (H, W, 3)
    inside = sum(int(((out[f"locs_{i}"][..., 0] >= 0) & (out[f"locs_{i}"][..., 0] < W) & (out[f"locs_{i}"][..., 1] >= 0)
                      & (out[f"locs_{i}"][..., 1] < H)).sum()) for i in range(n_img))
    path = os.path.join(HERE, fixture)
    np.savez_compressed(path, **out)
    print("warp fixture ->", path, f"{os.path.getsize(path) / 1024:.0f} KiB; pixels landing inside the reference view: {inside} of {n_img * H * W}")


if __name__ == "__main__":
    main()
