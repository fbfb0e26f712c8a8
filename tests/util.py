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
    """The oracle's fp32 and fp64 evaluations of the same rays, the way bench.py's `cpu_baseline` runs the oracle fastest
    (profiles/r4_cpu_sweep.json; VERDICT r5 "next" #5): the port is bound by per-operator overhead on small matrices, so ONE
    evaluation stops scaling at ~32 threads while MANY evaluations side by side over disjoint ray chunks keep scaling (torch
    releases the GIL inside its operators, OpenMP gives every calling thread its own team).  Both precisions' chunks share one
    pool: `workers` Python threads x 2 ATen threads per precision (64 + 64 workers on the GPU boxes' 256 hardware threads,
    1 + 1 on a small host).  Rays are independent, so the chunking changes nothing but the shapes the host BLAS sees (fp32
    results move by an ulp or two of their matrix products between chunkings -- far inside what the parity statistics
    resolve, and the contract is stated against THIS evaluation's own fp32-vs-fp64 gap).  `threads`: kept for callers of the
    round-2..5 form (ATen threads of a single evaluation); only its order of magnitude is used, as the cap on the workers."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    host = os.cpu_count() or 2
    per = 2
    workers = max(1, min(64, 2 * int(threads), host // (2 * per)))
    n = rays.shape[0]
    workers = max(1, min(workers, (n + 255) // 256))                # no chunk below 256 rays: the GEMMs would be all overhead
    sds = {dt: (oc.to_torch_sd(sd_c, dt), oc.to_torch_sd(sd_f, dt)) for dt in (torch.float32, torch.float64)}
    jobs = [(dt, c) for c32, c64 in zip(rays.to(torch.float32).chunk(workers), rays.to(torch.float64).chunk(workers))
            for dt, c in ((torch.float64, c64), (torch.float32, c32)) if c.shape[0]]

    def run(job):
        dt, r = job
        torch.set_num_threads(per)                                  # per calling thread (OpenMP ICV)
        with torch.no_grad():
            return oc.forward_rays(sds[dt][0], sds[dt][1], r, n_coarse, n_importance, white_bkgd)
    with ThreadPoolExecutor(max(1, len(jobs))) as ex:
        parts = list(ex.map(run, jobs))
    out = []
    for dt in (torch.float32, torch.float64):
        mine = [p for (d, _), p in zip(jobs, parts) if d == dt]
        out.append({k: torch.cat([p[k] for p in mine], 0) for k in mine[0]})
    return out[0], out[1]



def oracle_forward_parallel(sd_c, sd_f, rays, white_bkgd, n_coarse=64, n_importance=64, dtype=torch.float32, **kw):
    """One precision of the oracle over disjoint ray chunks side by side (see oracle_fp32_and_fp64): host threads // 2
    ATen threads in all, 2 per worker."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(64, (os.cpu_count() or 2) // 4, (rays.shape[0] + 255) // 256))
    sc, sf = oc.to_torch_sd(sd_c, dtype), oc.to_torch_sd(sd_f, dtype)
    chunks = [c for c in rays.to(dtype).chunk(workers) if c.shape[0]]

    def run(r):
        torch.set_num_threads(2)
        with torch.no_grad():
            return oc.forward_rays(sc, sf, r, n_coarse, n_importance, white_bkgd, **kw)
    with ThreadPoolExecutor(len(chunks)) as ex:
        parts = list(ex.map(run, chunks))
    return {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}



def explained_by_resampler_conditioning(sd_f, rays, white_bkgd, hip, ref, idx, n_coarse=64, n_importance=64,
                                        w_tol=2e-6, rgb_tol=1e-5, ref64=None):
    """Round 6 (VERDICT r5 "next" #2): which of the rays `idx` -- rays whose fine colour sits further from the fp32 oracle than
    the per-ray contract max(1e-4, 2 x the oracle's own fp32-vs-fp64 gap) allows -- are explained by the conditioning of the
    reference's inverse-CDF resampler ALONE?

    The gap is ONE draw of rounding noise: it samples the resampler's amplification with whatever the oracle's fp32 and fp64
    coarse weights happen to differ by in the sensitive bin, and can miss it (the ray of profiles/r5_parity_report.json,
    geometry 5, `fp32`: gap 1.6e-6, inverse-CDF slope 5,600 in a bin whose pdf sits 1.3e-6 above the `denom < 1e-5 -> 1` snap
    of models/utils.py:87-88; profiles/r6_fp32_ray_probe.json).  The direct test: feed the ORACLE's own fp32 fine pass
    (resampler, fine network, compositor) with the HIP path's coarse weights.  A ray is explained iff
      (a) the HIP coarse weights agree with the oracle's to fp32 rounding of the coarse network: <= `w_tol` (the random-init
          field's own fp32-vs-fp64 coarse weights differ by ~3e-7), or -- with `ref64`, the oracle's fp64 evaluation -- <= twice
          the oracle's OWN fp32-vs-fp64 distance of the coarse weights on that ray, the yardstick the colour contract itself uses
          (a trained field's densities reach hundreds: its fp32 oracle sits 3-4e-6 from its fp64 oracle in the coarse weights,
          the HIP path 2.3e-6 from the fp32 oracle on the ray of profiles/r6_trained_ray_probe_*.json), and
      (b) the oracle's fine pass on those weights reproduces the HIP colour (<= `rgb_tol`),
    i.e. everything behind the coarse weights is the reference's arithmetic, and the whole difference is the reference's own
    amplification of a rounding-level difference in front of it.  Returns a bool tensor over `idx`."""
    if len(idx) == 0:
        return torch.zeros(0, dtype=torch.bool)
    sf = oc.to_torch_sd(sd_f, torch.float32)
    r = rays[idx].float()
    with torch.no_grad():
        z, _ = oc.sample_coarse(r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8], n_coarse)
        w_hip = hip["coarse_weights"][idx].float()
        zf, xyzf = oc.resample_fine(r[:, 0:3], r[:, 3:6], z, w_hip, n_importance)
        de = oc.posenc(r[:, 8:11] if r.shape[1] == 11 else r[:, 3:6], 4)
        rgb, sig = oc.render_points(sf, xyzf, de)
        comp = oc.composite(rgb, sig, zf, white_bkgd)[0]
    tol_w = torch.full((len(idx),), float(w_tol), dtype=torch.float64)
    if ref64 is not None:
        own = (ref["coarse_weights"][idx].double() - ref64["coarse_weights"][idx].double()).abs().max(-1)[0]
        tol_w = torch.maximum(tol_w, 2.0 * own)
    a = (w_hip - ref["coarse_weights"][idx].float()).abs().max(-1)[0].double() <= tol_w
    b = (comp - hip["fine_comp_rgbs"][idx].float()).abs().max(-1)[0] <= rgb_tol
    return a & b
