"""CPU oracle of the LR-target construction of the downX datasets -- test infrastructure only.

The reference builds its low-resolution training targets with Pillow: the scene image is resized to the HR size and
then to the LR size with ``Image.LANCZOS`` on 8-bit RGB, converted with ``ToTensor`` (``/ 255``) and the HR image is
regrouped so that every LR pixel owns its s x s HR pixels (data/llff_downX_dataset.py:312-329,
data/blender_downX_dataset.py:117-135).  The arithmetic lives in a third-party dependency that is NOT in the
reference tree: ``pillow`` (unpinned in requirements.txt; the resampler has been stable since Pillow 3.x), file
``src/libImaging/Resample.c``, functions ``precompute_coeffs``, ``normalize_coeffs_8bpc``,
``ImagingResampleHorizontal_8bpc`` / ``ImagingResampleVertical_8bpc``.  Its published algorithm, restated here:

  * filter: ``lanczos(x) = sinc(x) sinc(x / 3)`` on [-3, 3), ``sinc(0) = 1``;
  * per output sample ``xx``: ``scale = in / out``, ``filterscale = max(scale, 1)``, ``support = 3 filterscale``,
    ``center = (xx + 0.5) scale``, taps ``xmin = int(center - support + 0.5)`` (clamped to 0) ..
    ``xmax = int(center + support + 0.5)`` (clamped to ``in``), weights
    ``w = lanczos((x + xmin - center + 0.5) / filterscale)`` normalised to sum 1 in double;
  * weights -> 22-bit fixed point: ``int(w 2^22 +- 0.5)`` (truncation toward zero of the shifted value);
  * a pass computes ``clip8((2^21 + sum_k pixel_k w_k) >> 22)`` in int32; the horizontal pass runs first (into an
    8-bit intermediate image), the vertical pass second.

RGBA images (the Blender scenes; data/blender_downX_dataset.py:104-118) are resampled premultiplied: Pillow's ``resize``
converts RGBA -> "RGBa" (``rgbA2rgba``: ``MULDIV255``), resamples the four channels as above, and converts back
(``rgba2rgbA``: ``clip8(255 c / a)``); the dataset then blends onto white in fp32.

Pinned by ``tests/golden/lanczos.npz`` and ``lanczos_rgba.npz`` (made by running Pillow itself in the development
container, ``make_golden_image.py`` / ``make_golden_image_rgba.py``): bit-exact.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        def sinc(t):
            if t == 0.0:
                return 1.0
            t = t * math.pi
            return math.sin(t) / t
        return sinc(x) * sinc(x / 3.0)
    return 0.0


def lanczos_coeffs(in_size: int, out_size: int):
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the whole-image box: returns
    ``bounds`` (out_size, 2) int32 = (first tap, number of taps) and ``kk`` (out_size, ksize) int32 fixed-point
    weights (zero beyond the tap count)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)         # same left-to-right double accumulation as the C loop
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One resampling pass over `axis` of an (H, W, C) uint8 image, int32 arithmetic (two's-complement wrap-around
    cannot occur: 255 * sum|w| < 2^31)."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_lanczos_u8(img: np.ndarray, out_wh) -> np.ndarray:
    """``Image.fromarray(img).resize(out_wh, Image.LANCZOS)`` for an (H, W, C) uint8 image -> (h, w, C) uint8."""
    H, W = img.shape[:2]
    w, h = int(out_wh[0]), int(out_wh[1])
    out = img
    if w != W:
        out = _pass(out, *lanczos_coeffs(W, w), axis=1)
    if h != H:
        out = _pass(out, *lanczos_coeffs(H, h), axis=0)
    return out


def premultiply_rgba(img: np.ndarray) -> np.ndarray:
    """Pillow's RGBA -> "RGBa" conversion (src/libImaging/Convert.c ``rgbA2rgba``): colour bytes become
    ``MULDIV255(c, a) = (t = c a + 128, ((t >> 8) + t) >> 8)``, alpha is kept."""
    c = img[..., :3].astype(np.uint32)
    a = img[..., 3:4].astype(np.uint32)
    t = c * a + 128
    out = img.copy()
    out[..., :3] = (((t >> 8) + t) >> 8).astype(np.uint8)
    return out


def unpremultiply_rgba(img: np.ndarray) -> np.ndarray:
    """Pillow's "RGBa" -> RGBA conversion (``rgba2rgbA``): ``clip8(255 c / a)`` (integer division) unless alpha is 0 or
    255, where the colour bytes are copied."""
    c = img[..., :3].astype(np.uint32)
    a = img[..., 3:4].astype(np.uint32)
    q = np.minimum((255 * c) // np.maximum(a, 1), 255).astype(np.uint8)
    out = img.copy()
    out[..., :3] = np.where((a == 0) | (a == 255), img[..., :3], q)
    return out


def resize_lanczos_rgba_u8(img: np.ndarray, out_wh) -> np.ndarray:
    """``Image.fromarray(img, "RGBA").resize(out_wh, Image.LANCZOS)``: Pillow resamples RGBA images premultiplied
    (PIL/Image.py ``resize``: ``convert("RGBa")`` -> resize -> ``convert("RGBA")``)."""
    return unpremultiply_rgba(resize_lanczos_u8(premultiply_rgba(img), out_wh))


def lr_targets_rgba(img_u8: np.ndarray, img_wh, downscale: int):
    """The Blender datasets' variant (data/blender_downX_dataset.py:104-118): RGBA scene image -> HR -> LR (both resized
    as RGBA), ``ToTensor``, then ``rgb * a + (1 - a)`` (blend onto white, fp32) -> ``rgbs`` (N_lr, 3), ``rgbs_ori``
    (N_lr, s*s, 3)."""
    W, H = int(img_wh[0]), int(img_wh[1])
    s = int(downscale)
    hr = resize_lanczos_rgba_u8(img_u8, (W, H))
    lr = resize_lanczos_rgba_u8(hr, (W // s, H // s))

    def blend(x):
        f = x.astype(np.float32) / np.float32(255.0)
        return f[..., :3] * f[..., 3:4] + (np.float32(1.0) - f[..., 3:4])
    hr_f, lr_f = blend(hr), blend(lr)
    h, w = H // s, W // s
    ori = hr_f.reshape(h, s, w, s, 3).transpose(0, 2, 1, 3, 4).reshape(h * w, s * s, 3)
    return lr_f.reshape(-1, 3), ori


def lr_targets(img_u8: np.ndarray, img_wh, downscale: int):
    """The two tensors the downX datasets keep per training image (data/llff_downX_dataset.py:312-329, ``ds_method =
    'lanc'``): ``rgbs`` (N_lr, 3) = the LANCZOS-downscaled image / 255 and ``rgbs_ori`` (N_lr, s*s, 3) = the HR image
    / 255 regrouped ``'(h s1) (w s2) c -> (h w) (s1 s2) c'``; the scene image is first resized to ``img_wh``."""
    W, H = int(img_wh[0]), int(img_wh[1])
    s = int(downscale)
    hr = resize_lanczos_u8(img_u8, (W, H))
    lr = resize_lanczos_u8(hr, (W // s, H // s))
    hr_f = hr.astype(np.float32) / np.float32(255.0)
    lr_f = lr.astype(np.float32) / np.float32(255.0)
    h, w = H // s, W // s
    ori = hr_f.reshape(h, s, w, s, 3).transpose(0, 2, 1, 3, 4).reshape(h * w, s * s, 3)
    return lr_f.reshape(-1, 3), ori
