#!/usr/bin/env python3
"""Golden vectors for nerf_sr_amd/io.py.  Development container only; data only.

A small synthetic COLMAP reconstruction is WRITTEN as real binary files (tests/colmap_writer.py), parsed with the
reference's own readers (utils/colmap.py) and pushed through the reference's LLFF pose pipeline -- the read_meta of
warp.py's LLFFDataset (same code as data/llff_downX_dataset.py), executed from the reference file as in
make_golden_warp.py.  The fixture holds the three binary files (bytes) and what the reference made of them.
"""
import os
import sys
import tempfile

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg  # noqa: E402
import make_golden_warp as mw  # noqa: E402
from colmap_writer import write_reconstruction  # noqa: E402


def main():
    mg.install_shim()
    import torchvision.transforms as T
    T.ToTensor, T.ToPILImage = mw.ToTensor, mw.ToPILImage
    rng = np.random.default_rng(11)
    W, H, n_img = 16, 12, 5
    root = tempfile.mkdtemp(prefix="nsr_io_root_")
    result = tempfile.mkdtemp(prefix="nsr_io_res_")
    os.makedirs(os.path.join(root, "images"))
    names = [f"view_{c}.png" for c in "dbeac"]                      # deliberately NOT in sorted order
    for n in names:
        Image.fromarray(rng.integers(0, 256, (H * 4, W * 4, 3), dtype=np.uint8)).save(os.path.join(root, "images", n))
    pts = rng.normal(0, 0.8, (40, 3)) + np.array([0.0, 0.0, 5.0])
    c2ws = [mw.look_at_colmap(np.array([0.4 * (i - 2), 0.15 * (i % 3), -0.1 * i]), np.array([0.0, 0.1, 5.0])) for i in range(n_img)]
    tracks = [sorted(rng.choice(np.arange(1, n_img + 1), size=rng.integers(2, n_img + 1), replace=False).tolist()) for _ in pts]
    sparse = os.path.join(root, "sparse", "0")
    write_reconstruction(sparse, W * 4, H * 4, 2.3 * W * 4, names, c2ws, pts, tracks)
    for i in range(n_img):
        np.savez(os.path.join(result, f"{i}-fine-depth-ori.npz"), (0.3 + 0.5 * rng.random((H, W, 1))).astype(np.float32))
    import utils.colmap as uc
    cams, imgs, p3d = (uc.read_cameras_binary(os.path.join(sparse, "cameras.bin")),
                       uc.read_images_binary(os.path.join(sparse, "images.bin")),
                       uc.read_points3d_binary(os.path.join(sparse, "points3D.bin")))
    src = open(os.path.join(mg.REF, "warp.py")).read()
    ns = {"__name__": "reference_warp"}
    cwd = os.getcwd()
    os.chdir(mg.REF)
    try:
        exec(compile(src[:src.index("\nwidth = ")], os.path.join(mg.REF, "warp.py"), "exec"), ns)
    finally:
        os.chdir(cwd)
    ds = ns["LLFFDataset"](root, result, W, H)
    out = {"W": W, "n_img": n_img}
    for fn in ("cameras.bin", "images.bin", "points3D.bin"):
        out["file_" + fn] = np.frombuffer(open(os.path.join(sparse, fn), "rb").read(), dtype=np.uint8)
    out["cam_params"], out["cam_wh"] = np.asarray(cams[1].params), np.array([cams[1].width, cams[1].height])
    ids = sorted(imgs)
    out["img_ids"] = np.array(ids)
    out["img_names"] = np.array([imgs[k].name for k in ids])
    out["img_qvec"] = np.stack([imgs[k].qvec for k in ids])
    out["img_tvec"] = np.stack([imgs[k].tvec for k in ids])
    out["img_rot"] = np.stack([imgs[k].qvec2rotmat() for k in ids])
    out["img_npts"] = np.array([len(imgs[k].point3D_ids) for k in ids])
    pids = sorted(p3d)
    out["pt_xyz"] = np.stack([p3d[k].xyz for k in pids])
    out["pt_track_len"] = np.array([len(p3d[k].image_ids) for k in pids])
    out["pt_first_image"] = np.array([p3d[k].image_ids[0] for k in pids])
    out["focal"], out["poses"], out["bounds"] = np.float64(ds.focal), np.asarray(ds.poses), np.asarray(ds.bounds)
    out["image_paths"] = np.array([os.path.basename(p) for p in ds.image_paths])
    path = os.path.join(HERE, "io_colmap.npz")
    np.savez_compressed(path, **out)
    print("->", path, f"{os.path.getsize(path) / 1024:.0f} KiB; focal {ds.focal:.4f}; bounds {ds.bounds.min():.3f}..{ds.bounds.max():.3f}")


if __name__ == "__main__":
    main()
