"""Pins oracle/image_oracle.py (restatement of Pillow's 8-bit LANCZOS resampler and of the downX datasets' LR-target
lines) to tests/golden/lanczos.npz, produced by Pillow itself (tests/golden/make_golden_image.py).  Bit-exact.  CPU only."""
import os

import numpy as np

from oracle import image_oracle as io

SIZES = ((64, 48), (32, 24), (40, 30), (100, 70), (83, 20), (200, 61))


def test_resize_matches_pillow(golden_dir):
    g = np.load(os.path.join(golden_dir, "lanczos.npz"))
    for name in ("smooth", "noise"):
        for w, h in SIZES:
            got = io.resize_lanczos_u8(g[name], (w, h))
            assert got.dtype == np.uint8 and np.array_equal(got, g[f"{name}_{w}x{h}"]), (name, w, h)


def test_lr_targets_match_the_dataset_lines(golden_dir):
    g = np.load(os.path.join(golden_dir, "lanczos.npz"))
    for s in (2, 4):
        rgbs, ori = io.lr_targets(g["smooth"], (64, 48), s)
        assert np.array_equal(rgbs, g[f"rgbs_s{s}"]) and np.array_equal(ori, g[f"rgbs_ori_s{s}"])
        assert rgbs.shape == (64 * 48 // (s * s), 3) and ori.shape == (64 * 48 // (s * s), s * s, 3)


def test_coefficient_tables():
    b, kk = io.lanczos_coeffs(504, 252)
    assert b.shape == (252, 2) and kk.shape == (252, 13)          # support 6 -> 2 * 6 + 1 taps
    assert abs(int(kk[100].sum()) - (1 << 22)) <= 13             # normalised, up to one rounding per tap
    b, kk = io.lanczos_coeffs(100, 250)                          # upscaling: support 3
    assert kk.shape == (250, 7) and (b[:, 1] <= 7).all() and b[0, 0] == 0


def test_libnsr_host_tables_equal_the_oracle():
    """nsr_lanczos_coeffs is plain host C (no GPU needed): its fixed-point tables must equal the restatement's."""
    from nerf_sr_amd import io as nsr_io
    for n_in, n_out in ((504, 252), (378, 189), (4032, 504), (100, 250), (61, 61), (83, 20), (7, 3)):
        b, kk = nsr_io.lanczos_tables(n_in, n_out)
        wb, wk = io.lanczos_coeffs(n_in, n_out)
        assert np.array_equal(b, wb) and np.array_equal(kk, wk), (n_in, n_out)


def test_rgba_resize_and_targets_match_pillow(golden_dir):
    """Blender scenes are RGBA: premultiplied resampling and the blend onto white (data/blender_downX_dataset.py:104-118)."""
    g = np.load(os.path.join(golden_dir, "lanczos_rgba.npz"))
    for name in ("obj", "noise"):
        assert np.array_equal(io.premultiply_rgba(g[name]), g[f"{name}_RGBa"])
        assert np.array_equal(io.unpremultiply_rgba(g[f"{name}_RGBa"]), g[f"{name}_RGBa_back"])
        for w, h in ((64, 64), (32, 32), (40, 30), (120, 70)):
            assert np.array_equal(io.resize_lanczos_rgba_u8(g[name], (w, h)), g[f"{name}_{w}x{h}"]), (name, w, h)
    for s in (2, 4):
        rgbs, ori = io.lr_targets_rgba(g["obj"], (64, 64), s)
        assert np.array_equal(rgbs, g[f"rgbs_s{s}"]) and np.array_equal(ori, g[f"rgbs_ori_s{s}"])
