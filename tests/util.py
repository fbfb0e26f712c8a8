"""Shared assertions of the parity tests."""
import numpy as np
import torch

from oracle import nerf_oracle as oc


def to_np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def assert_close(got, want, atol):
    got, want = to_np(got), to_np(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if got.size else 0.0
    assert err <= atol, f"max abs err {err:.3e} > {atol:.1e}"
    return err


def assert_resample_close(got, want, z, weights, n_imp, u=None, base_tol=2e-6):
    """z_fine parity with a conditioning-aware bound (see oracle.resample_conditioning).

    Well-conditioned rays: |err| <= base_tol + 5e-7 * amp (fp32 rounding noise of the cdf times
    the inverse-CDF slope).  Rays whose ``denom`` lies within 5e-7 of the reference's 1e-5 snap
    threshold are genuinely two-valued in fp32: there a new sample may sit anywhere in its bin.
    """
    got, want = to_np(got).astype(np.float64), to_np(want).astype(np.float64)
    assert got.shape == want.shape
    zc = torch.as_tensor(to_np(z))
    wc = torch.as_tensor(to_np(weights))
    uc = None if u is None else torch.as_tensor(to_np(u))
    amp, margin, width = [t.numpy() for t in oc.resample_conditioning(zc, wc, n_imp, uc)]
    err = np.abs(got - want).max(-1)
    tol = base_tol + 5e-7 * amp
    chaotic = margin < 5e-7
    tol = np.where(chaotic, np.maximum(tol, width * 1.0001), tol)
    bad = np.nonzero(err > tol)[0]
    assert bad.size == 0, (f"{bad.size} rays outside the conditioning bound; worst ray {bad[0]}: err {err[bad[0]]:.3e} "
                           f"tol {tol[bad[0]]:.3e} amp {amp[bad[0]]:.3e} margin {margin[bad[0]]:.3e}")
    assert (np.diff(got, axis=-1) >= 0).all(), "z_fine not sorted"
    return float(err.max()), int(chaotic.sum())


def sample_idx(numel: int, cap: int = 512):
    """Subsample of a flattened tensor used by the training fixtures (must match
    tests/golden/make_golden_train.py::sample_idx)."""
    if numel <= cap:
        return np.arange(numel)
    stride = (numel // cap) | 1
    return np.arange(0, numel, stride)[:cap]


def train_draws(g):
    """The random draws recorded in a training fixture as keyword arguments of the oracle / HIP step."""
    d = {"noise_std": float(g["noise_std"])}
    for k in ("u_coarse", "noise_coarse", "u_fine", "noise_fine"):
        if k in g:
            d[k] = g[k]
    return d


def oracle_fp32_and_fp64(sd_c, sd_f, rays, white_bkgd, n_coarse=64, n_importance=64, threads=32):
    """The oracle's fp32 and fp64 evaluations of the same rays, run CONCURRENTLY in two Python threads with `threads` ATen
    threads each (torch releases the GIL inside its operators; OpenMP gives each calling thread its own team).  The GPU
    boxes have 128 hardware threads and the oracle stops scaling at ~32 (bench.py's policy), so running the two
    evaluations side by side costs about max(fp32, fp64) instead of their sum.  The fp32 results do not depend on the
    thread count (checked: bit-identical); the fp64 ones move by one ulp (2e-16) in the final colour sum, far below anything
    the parity statistics resolve."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = max(1, min(int(threads), (os.cpu_count() or 2) // 2))       # two evaluations share the host

    def run(dtype):
        torch.set_num_threads(n)                                    # per calling thread (OpenMP ICV); MKL: the same n for both
        with torch.no_grad():
            return oc.forward_rays(oc.to_torch_sd(sd_c, dtype), oc.to_torch_sd(sd_f, dtype), rays.to(dtype), n_coarse,
                                   n_importance, white_bkgd)
    with ThreadPoolExecutor(2) as ex:
        f32, f64 = ex.submit(run, torch.float32), ex.submit(run, torch.float64)
        return f32.result(), f64.result()
