"""N4 LR-target construction on the device (include/nsr_image.h) against the Pillow-generated fixture and the oracle:
8-bit LANCZOS resampling and the ToTensor + regroup step.  Integer / byte work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import image_oracle as io_oracle

pytestmark = pytest.mark.gpu
SIZES = ((64, 48), (32, 24), (40, 30), (100, 70), (83, 20), (200, 61))


@pytest.fixture(scope="module")
def nio():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from nerf_sr_amd import io as _io
    return _io


def test_resize_vs_pillow_fixture(nio, golden_dir):
    g = np.load(os.path.join(golden_dir, "lanczos.npz"))
    for name in ("smooth", "noise"):
        src = torch.from_numpy(g[name]).cuda()
        for w, h in SIZES:
            got = nio.resize_lanczos_u8(src, (w, h))
            assert got.dtype == torch.uint8 and np.array_equal(got.cpu().numpy(), g[f"{name}_{w}x{h}"]), (name, w, h)
        assert torch.equal(nio.resize_lanczos_u8(src, (src.shape[1], src.shape[0])), src)     # identity size: no pass


def test_lr_targets_vs_dataset_fixture(nio, golden_dir):
    g = np.load(os.path.join(golden_dir, "lanczos.npz"))
    for s in (2, 4):
        rgbs, ori = nio.lr_targets(torch.from_numpy(g["smooth"]).cuda(), (64, 48), s)
        assert np.array_equal(rgbs.cpu().numpy(), g[f"rgbs_s{s}"]) and np.array_equal(ori.cpu().numpy(), g[f"rgbs_ori_s{s}"])


def test_fern_sized_image_vs_oracle(nio):
    """A 4032 x 3024 capture (the LLFF scenes' native size) -> HR 504 x 378 -> LR 252 x 189 (BASELINE config #2),
    against the oracle: every byte and every target value identical; a 1-channel image and an upscale as well."""
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:3024, 0:4032]
    img = np.stack([127 + 100 * np.sin(xx / 31.0) * np.cos(yy / 57.0), (xx + yy) % 256, 255 - (xx // 16 + yy // 12) % 256], -1)
    img = np.clip(img + rng.integers(-20, 21, img.shape), 0, 255).astype(np.uint8)
    rgbs, ori = nio.lr_targets(torch.from_numpy(img).cuda(), (504, 378), 2)
    want_rgbs, want_ori = io_oracle.lr_targets(img, (504, 378), 2)
    assert np.array_equal(rgbs.cpu().numpy(), want_rgbs) and np.array_equal(ori.cpu().numpy(), want_ori)
    gray = img[:400, :300, :1].copy()
    assert np.array_equal(nio.resize_lanczos_u8(torch.from_numpy(gray).cuda(), (450, 123)).cpu().numpy(),
                          io_oracle.resize_lanczos_u8(gray, (450, 123)))


def test_rgba_resize_and_targets_vs_pillow_fixture(nio, golden_dir):
    """Blender scenes are RGBA (data/blender_downX_dataset.py:104-120): premultiplied resampling, blend onto white."""
    g = np.load(os.path.join(golden_dir, "lanczos_rgba.npz"))
    for name in ("obj", "noise"):
        src = torch.from_numpy(g[name]).cuda()
        for w, h in ((64, 64), (32, 32), (40, 30), (120, 70)):
            assert np.array_equal(nio.resize_lanczos_u8(src, (w, h)).cpu().numpy(), g[f"{name}_{w}x{h}"]), (name, w, h)
        assert torch.equal(nio.resize_lanczos_u8(src, (src.shape[1], src.shape[0])), src)     # identity size: untouched
    for s in (2, 4):
        rgbs, ori = nio.lr_targets(torch.from_numpy(g["obj"]).cuda(), (64, 64), s)
        assert np.array_equal(rgbs.cpu().numpy(), g[f"rgbs_s{s}"]) and np.array_equal(ori.cpu().numpy(), g[f"rgbs_ori_s{s}"])
    # an 800 x 800 RGBA render (the Blender scenes' native size) -> HR 400 x 400 -> LR 200 x 200 (BASELINE config #3)
    rng = np.random.default_rng(9)
    yy, xx = np.mgrid[0:800, 0:800]
    a = np.clip((300.0 - np.hypot(xx - 400.0, yy - 380.0)) * 25.0, 0, 255)
    img = np.stack([127 + 100 * np.sin(xx / 23.0), (xx + yy) % 256, 255 - (xx // 16 + yy // 12) % 256, a], -1)
    img = np.clip(img + rng.integers(-10, 11, img.shape) * (img[..., 3:4] > 0), 0, 255).astype(np.uint8)
    rgbs, ori = nio.lr_targets(torch.from_numpy(img).cuda(), (400, 400), 2)
    want_rgbs, want_ori = io_oracle.lr_targets_rgba(img, (400, 400), 2)
    assert np.array_equal(rgbs.cpu().numpy(), want_rgbs) and np.array_equal(ori.cpu().numpy(), want_ori)
    with pytest.raises(ValueError):
        nio.lr_targets(torch.zeros(8, 8, 2, dtype=torch.uint8, device="cuda"), (4, 4), 2)
