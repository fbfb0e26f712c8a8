"""Frame-scale tests of the HIP path on the BASELINE frame geometries (run on the GPU box with ``pytest -m gpu``).

1. PARITY AT FRAME SCALE, configs #2-#5, both contract-grade precisions: 16,384 consecutive rays from the middle of
   one frame of each geometry against the CPU oracle.  Asserted on EVERY ray: |dRGB| of the fine colours <= max(1e-4,
   2 x the oracle's OWN fp32-vs-fp64 gap on that ray) -- the reference's inverse-CDF resampling amplifies the fp32
   rounding noise of the coarse weights, so two fp32 evaluations of an ill-conditioned ray may each sit `gap` away from
   the exact result, on opposite sides: no second fp32 implementation, the reference on another BLAS included, can be
   held closer.  The rays whose bound is the second term are counted, printed and bounded (at most 8 of 16,384 per
   geometry on the smooth field).  Also asserted: |dPSNR| <= 1e-3 dB against a common target, coarse colours <= 1e-5,
   s^2 means <= 1e-4.  The same protocol on a sharp stress field (envelope rule) follows; on TRAINED fields it is
   tests/test_gpu_trained.py.
2. SHARDING: config #4's frame (1008 x 756 <- 252 x 189) rendered in 2 and in 3 contiguous LR-pixel blocks on one
   device is bit-identical to the unsplit render (what ``render_image_sharded`` does on N GPUs, minus the gather).
3. CONFIG #5 COMPOSED: render -> depth -> warp (ray-distance variant) -> refinement network, stage by stage against
   the oracles on a small Blender-like frame, and end to end at 800 x 800.
"""
import time

import numpy as np
import pytest
import torch

from nerf_sr_amd import cameras
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc

pytestmark = pytest.mark.gpu

RGB_TOL, PSNR_TOL = 1e-4, 1e-3
N_RAYS = 16384
MAX_EXEMPT = 8
# BASELINE.json configs: HR size (W, H), supersampling, NDC?, white background?
CONFIGS = {2: ((504, 378), 2, True, False), 3: ((400, 400), 2, False, True), 4: ((1008, 756), 4, True, False),
           5: ((800, 800), 4, False, True)}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from nerf_sr_amd import ops as _ops   # raises if libnsr.so is missing: no fallback
    return _ops


def _camera(cid):
    wh, s, ndc, white = CONFIGS[cid]
    if ndc:
        return cameras.spiral_pose(0.4), cameras.llff_focal(wh[0]), (0.0, 1.0)
    return cameras.spheric_pose(40.0, -30.0, 4.0), cameras.blender_focal(wh[0]), (2.0, 6.0)


_ORACLE_CACHE = {}


N_RAYS_RECORD, CID_RECORD = 65536, 5      # the per-round parity record's block (scripts/parity_record.py) -- as a test on one geometry


def _oracle_block(cid, field="smooth", n=N_RAYS):
    """fp32 and fp64 oracle outputs of the first `n` rays of the test block of config `cid` (cached: both precisions use it;
    the record-sized block of CID_RECORD is evaluated once and serves the 16,384-ray test as its first quarter)."""
    n_eval = N_RAYS_RECORD if (cid == CID_RECORD and field == "smooth") else n
    if (cid, field) not in _ORACLE_CACHE:
        wh, s, ndc, white = CONFIGS[cid]
        c2w, f, nf = _camera(cid)
        rays = oc.subpixel_ray_grid(torch.from_numpy(c2w), wh[1], wh[0], f, s, ndc, *nf).reshape(-1, 8)
        lo = (rays.shape[0] // 2) - (rays.shape[0] // 2) % (s * s)
        blk = rays[lo:lo + n_eval].contiguous()
        sd_c, sd_f = make_state_dict(99, field=field), make_state_dict(100, field=field)
        from tests.util import oracle_fp32_and_fp64
        t0 = time.time()
        ref, ref64 = oracle_fp32_and_fp64(sd_c, sd_f, blk, white)        # both precisions, chunks side by side over the host's cores
        print(f"[config #{cid} {field}] oracle fp32 + fp64 on {n_eval} rays: {time.time() - t0:.1f} s")
        _ORACLE_CACHE[(cid, field)] = (lo, blk, ref, ref64)
    lo, blk, ref, ref64 = _ORACLE_CACHE[(cid, field)]
    assert n <= blk.shape[0]
    return lo, blk[:n].contiguous(), {k: v[:n] for k, v in ref.items()}, {k: v[:n] for k, v in ref64.items()}


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
@pytest.mark.parametrize("cid", [2, 3, 4, 5])
def test_frame_scale_parity(ops, cid, prec):
    wh, s, ndc, white = CONFIGS[cid]
    c2w, f, nf = _camera(cid)
    lo, blk, ref, ref64 = _oracle_block(cid)
    s2 = s * s
    n_lr = (wh[0] // s) * (wh[1] // s)
    # the block comes out of the device-side ray generator (range variant), as it does in a sharded render
    rays = ops.subpixel_rays(c2w, wh, f, s, ndc, *nf, lr_range=(lo // s2, (lo + N_RAYS) // s2)).view(-1, 8)
    assert rays.shape[0] == N_RAYS and n_lr * s2 == wh[0] * wh[1]
    assert float((rays.cpu() - blk).abs().max()) <= 2e-6
    sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(sd_c)
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
    # ... but the comparison itself runs on identical inputs: the oracle's rays
    o = {k: v.cpu() for k, v in ops.forward_rays(net_c, net_f, blk.cuda(), 64, 64, white).items()}
    # coarse colours: no resampling in front of them, plain fp32-grade agreement
    assert float((o["coarse_comp_rgbs"] - ref["coarse_comp_rgbs"]).abs().max()) <= 1e-5
    d = (o["fine_comp_rgbs"].double() - ref["fine_comp_rgbs"].double()).abs().max(-1)[0]
    gap = (ref["fine_comp_rgbs"].double() - ref64["fine_comp_rgbs"]).abs().max(-1)[0]      # the oracle's own fp32 vs fp64
    # the contract, per ray: |dRGB| <= max(1e-4, 2 x the oracle's own fp32-vs-fp64 gap on that ray) -- two fp32 evaluations of
    # an ill-conditioned ray may each sit `gap` away from the exact result, on opposite sides
    bound = torch.clamp_min(2.0 * gap, RGB_TOL)
    exempt = 2.0 * gap > RGB_TOL                 # rays whose bound is the conditioning term
    over = d > RGB_TOL
    # the round-2 form of the rule, kept as a REPORTED count next to the restated one: rays whose oracle gap is
    # <= 1e-4 yet differ by more than 1e-4 -- the rays the restatement lets through with a bound between 1e-4 and 2e-4
    strict_over = int((over & (gap <= RGB_TOL)).sum())
    assert strict_over <= 2, f"{strict_over} rays with an oracle gap <= 1e-4 are over 1e-4 (round-2 rule; the restated bound allows them only up to 2e-4)"
    print(f"[config #{cid} {prec}] fine max|dRGB| {float(d.max()):.2e} (non-exempt {float(d[~exempt].max()):.2e}), "
          f"p99.9 {float(torch.quantile(d, 0.999)):.2e}, median {float(d.median()):.1e}; rays over 1e-4: {int(over.sum())}, "
          f"rays whose bound is 2 x oracle gap (> 1e-4): {int(exempt.sum())}; oracle gap max {float(gap.max()):.2e}; "
          f"violations: {int((d > bound).sum())}; over 1e-4 with oracle gap <= 1e-4 (round-2 rule): {strict_over}")
    assert int(exempt.sum()) <= MAX_EXEMPT
    assert int((d > bound).sum()) == 0, f"{int((d > bound).sum())} rays exceed max(1e-4, 2 x oracle gap) (worst by {float((d - bound).max()):.2e})"
    # the s^2 means: the image the reference trains and evaluates on
    lr = ops.sr_mean(o["fine_comp_rgbs"].cuda(), N_RAYS // s2, s2).cpu()
    lr_ref = oc.sr_mean(ref["fine_comp_rgbs"], N_RAYS // s2, s2)
    lr_exempt = exempt.view(-1, s2).any(-1)
    assert float((lr - lr_ref).abs().max(-1)[0][~lr_exempt].max()) <= RGB_TOL
    # PSNR against a common target (the oracle's coarse render), and PSNR(build, oracle)
    tgt = ref["coarse_comp_rgbs"]
    assert abs(oc.psnr(o["fine_comp_rgbs"], tgt) - oc.psnr(ref["fine_comp_rgbs"], tgt)) <= PSNR_TOL
    assert oc.psnr(o["fine_comp_rgbs"], ref["fine_comp_rgbs"]) > 100.0
    far = nf[1]
    assert float((o["fine_depth"] - ref["fine_depth"]).abs()[~exempt].max()) <= 2e-4 * far
    assert float((o["fine_opacity"] - ref["fine_opacity"]).abs()[~exempt].max()) <= 2e-4


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_frame_scale_parity_record_block(ops, prec):
    """The per-round parity record's protocol (scripts/parity_record.py: 65,536 consecutive rays) as a TEST, on config #5's
    geometry -- the block on which round 5's record shows ONE ray of the `fp32` kernel at |dRGB| 1.63e-4 with an oracle
    fp32-vs-fp64 gap of 1.6e-6 (none of the 16,384 rays of the test above).  Root cause
    (scripts/fp32_ray_probe.py -> profiles/r6_fp32_ray_probe.json): the coarse passes agree to 4.6e-7 in the weights (the
    oracle's own fp32 and fp64: 3.4e-7); the ray's 108th fine sample falls in a coarse bin whose pdf (1.13e-5) sits 1.3e-6
    above the resampler's `denom < 1e-5 -> 1` snap (models/utils.py:87-88), where the inverse CDF has slope
    bin width / denom = 5,600: the sample moves by 6.7e-4 in depth (the oracle's two evaluations: 6.6e-6 -- their rounding
    happened to agree in that bin, which is why the GAP does not flag the ray), and the colour follows.  Everything behind
    the coarse weights is exact: the oracle's fp32 fine pass fed with the HIP coarse weights reproduces the HIP colour to
    1.2e-7.  So the contract per ray is as above, and a ray over it must be EXPLAINED that way
    (tests/util.py::explained_by_resampler_conditioning: coarse weights within 2e-6, colour reproduced within 1e-5) -- at
    most 2 such rays per 65,536, none unexplained."""
    from tests.util import explained_by_resampler_conditioning
    wh, s, ndc, white = CONFIGS[CID_RECORD]
    lo, blk, ref, ref64 = _oracle_block(CID_RECORD, n=N_RAYS_RECORD)
    sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(sd_c)
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
    o = {k: v.cpu() for k, v in ops.forward_rays(net_c, net_f, blk.cuda(), 64, 64, white).items()}
    assert float((o["coarse_comp_rgbs"] - ref["coarse_comp_rgbs"]).abs().max()) <= 1e-5
    assert float((o["coarse_weights"] - ref["coarse_weights"]).abs().max()) <= 2e-6
    d = (o["fine_comp_rgbs"].double() - ref["fine_comp_rgbs"].double()).abs().max(-1)[0]
    gap = (ref["fine_comp_rgbs"].double() - ref64["fine_comp_rgbs"]).abs().max(-1)[0]
    bound = torch.clamp_min(2.0 * gap, RGB_TOL)
    over = torch.nonzero(d > bound).flatten()
    explained = explained_by_resampler_conditioning(sd_f, blk, white, o, ref, over)
    print(f"[config #{CID_RECORD} {prec}, {N_RAYS_RECORD} rays] fine max|dRGB| {float(d.max()):.2e}, p99.9 {float(torch.quantile(d, 0.999)):.2e}, "
          f"median {float(d.median()):.1e}; rays over 1e-4: {int((d > RGB_TOL).sum())}; rays whose bound is 2 x oracle gap: "
          f"{int((2.0 * gap > RGB_TOL).sum())}; rays over their bound: {over.tolist()} (d {[float(d[i]) for i in over]}, gap "
          f"{[float(gap[i]) for i in over]}), explained by the resampler's conditioning: {explained.tolist()}")
    assert bool(explained.all()), f"rays {over[~explained].tolist()} exceed max(1e-4, 2 x oracle gap) and the oracle's fine pass on the HIP coarse weights does NOT reproduce them"
    assert over.numel() <= 2
    assert int((2.0 * gap > RGB_TOL).sum()) <= 4 * MAX_EXEMPT
    assert oc.psnr(o["fine_comp_rgbs"], ref["fine_comp_rgbs"]) > 100.0


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
@pytest.mark.parametrize("cid", [2, 3])
def test_sharp_field_envelope_parity(ops, cid, prec):
    """The "sharp" stress field (white PE spectrum, x30 density head, weights.make_state_dict) at frame scale, both
    contract-grade precisions: the reference algorithm itself is chaotic there -- its own fp32 and fp64 evaluations differ by
    > 1e-4 on 7-10 % of the rays and by up to 0.15 -- so a per-ray bound is meaningless and parity is an ENVELOPE: no
    statistic of the HIP-vs-oracle32 error distribution (median, p99, p99.9, rays over 1e-4, max) may be worse than 2 x the
    same statistic of the oracle's fp64-vs-fp32 distribution.  (The trained fields of test_gpu_trained.py are where the
    per-ray contract is asserted on realistic densities.)  Coarse colours, which have no resampling in front of them,
    stay tight on every ray."""
    wh, s, ndc, white = CONFIGS[cid]
    lo, blk, ref, ref64 = _oracle_block(cid, "sharp")
    sd_c, sd_f = make_state_dict(99, field="sharp"), make_state_dict(100, field="sharp")
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(sd_c)
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
    o = {k: v.cpu() for k, v in ops.forward_rays(net_c, net_f, blk.cuda(), 64, 64, white, check=True).items()}
    # coarse colours: no resampling in front of them, but the x30 density head multiplies the fp32 rounding of the trunk, so
    # the yardstick is again the oracle's own fp32-vs-fp64 distance (1.7e-5 / 2e-4 measured on #2 / #3), not 1e-5
    c_gap = float((ref64["coarse_comp_rgbs"] - ref["coarse_comp_rgbs"].double()).abs().max())
    c_err = float((o["coarse_comp_rgbs"] - ref["coarse_comp_rgbs"]).abs().max())
    print(f"[config #{cid} sharp {prec}] coarse max|dRGB| {c_err:.2e} (oracle fp32 vs fp64: {c_gap:.2e})")
    assert c_err <= max(1e-5, 3.0 * c_gap)
    e_hip = (o["fine_comp_rgbs"].double() - ref["fine_comp_rgbs"].double()).abs().max(-1)[0]
    e_ref = (ref64["fine_comp_rgbs"] - ref["fine_comp_rgbs"].double()).abs().max(-1)[0]
    q = lambda t, p: float(torch.quantile(t, p))
    print(f"[config #{cid} sharp {prec}] HIP vs oracle32: median {float(e_hip.median()):.1e} p99 {q(e_hip, 0.99):.1e} p99.9 "
          f"{q(e_hip, 0.999):.1e} max {float(e_hip.max()):.1e} over 1e-4: {int((e_hip > RGB_TOL).sum())} | oracle64 vs oracle32: "
          f"median {float(e_ref.median()):.1e} p99 {q(e_ref, 0.99):.1e} p99.9 {q(e_ref, 0.999):.1e} max {float(e_ref.max()):.1e} "
          f"over 1e-4: {int((e_ref > RGB_TOL).sum())}")
    assert int((e_ref > RGB_TOL).sum()) > N_RAYS // 50                       # the field is what it claims to be
    assert float(e_hip.median()) <= 2 * float(e_ref.median()) and q(e_hip, 0.99) <= 2 * q(e_ref, 0.99)
    assert q(e_hip, 0.999) <= 2 * q(e_ref, 0.999) and float(e_hip.max()) <= 2 * float(e_ref.max())
    assert int((e_hip > RGB_TOL).sum()) <= 2 * int((e_ref > RGB_TOL).sum())
    # the image the reference evaluates on: PSNR(build, oracle32) no worse than PSNR(oracle64, oracle32) by more than 3 dB
    assert oc.psnr(o["fine_comp_rgbs"], ref["fine_comp_rgbs"]) >= oc.psnr(ref64["fine_comp_rgbs"].float(), ref["fine_comp_rgbs"]) - 3.0


@pytest.mark.parametrize("cid", [2, 3])
def test_frame_scale_resampling_and_how_many_rays_are_ill_conditioned(ops, cid):
    """S2 alone at frame scale, on identical inputs (the oracle's coarse weights): every ray inside the conditioning-aware
    bound of tests/util.py, and a COUNT of the rays that bound has to treat as two-valued (a resampling denominator
    within 5e-7 of the reference's `< 1e-5 -> 1` snap): they exist, but they are a fraction of a percent of a frame."""
    from tests.util import assert_resample_close
    lo, blk, ref, _ = _oracle_block(cid)
    o, d, near, far = blk[:, 0:3], blk[:, 3:6], blk[:, 6:7], blk[:, 7:8]
    z_c, _ = oc.sample_coarse(o, d, near, far, 64)
    z_ref, _ = oc.resample_fine(o, d, z_c, ref["coarse_weights"], 64)
    z_got, _ = ops.resample_along_rays(o.cuda(), d.cuda(), z_c.cuda(), ref["coarse_weights"].cuda(), 64, False)
    err, n_chaotic = assert_resample_close(z_got, z_ref, z_c, ref["coarse_weights"], 64)
    print(f"[config #{cid}] z_fine max err {err:.2e}; rays with a snap-adjacent denominator: {n_chaotic} of {N_RAYS}")
    assert n_chaotic <= N_RAYS // 100


# ------------------------------------------------------------------------------------------------- sharding
def test_gen_rays_range_is_a_slice_of_the_frame(ops):
    for cid in (3, 4):
        wh, s, ndc, white = CONFIGS[cid]
        c2w, f, nf = _camera(cid)
        full = ops.subpixel_rays(c2w, wh, f, s, ndc, *nf)
        n_lr = full.shape[0]
        for lo, hi in ((0, 1), (17, 4099), (n_lr - 5, n_lr), (123, 123)):
            part = ops.subpixel_rays(c2w, wh, f, s, ndc, *nf, lr_range=(lo, hi))
            assert part.shape == (hi - lo, s * s, 8) and torch.equal(part, full[lo:hi])
        with pytest.raises(Exception):
            ops.subpixel_rays(c2w, wh, f, s, ndc, *nf, lr_range=(5, n_lr + 1))


def test_sharded_render_is_bit_identical_config4(ops):
    """BASELINE config #4 (1008 x 756 <- 252 x 189, 4x SS): the frame cut into 2 and into 8 contiguous LR-pixel blocks
    (dist.shard_bounds, what N ranks render) equals the unsplit render bit for bit: LR image, depth, HR outputs."""
    from nerf_sr_amd import dist as nsr_dist
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    wh, s, ndc, white = CONFIGS[4]
    c2w, f, nf = _camera(4)
    opt = default_options(img_wh=wh, downscale=s, white_bkgd=white, precision="f16x3")
    model = NeRFDownXModel(opt).load_networks(make_state_dict(99), make_state_dict(100)).eval()
    whole = model.render_image(c2w, f, ndc, *nf)
    lr_rgb, lr_depth = whole["lr_rgb"].clone(), whole["lr_depth"].clone()
    hr_rgb = model.out_fine_comp_rgbs_ori.clone()
    n_lr = lr_rgb.shape[0]
    assert n_lr == 47628 and hr_rgb.shape[0] == 762048
    for world in (2, 8):
        bounds = nsr_dist.shard_bounds(n_lr, world)
        parts = [model.render_image_sharded(c2w, f, ndc, *nf, lr_range=b) for b in bounds]
        assert torch.equal(torch.cat([p["lr_rgb"] for p in parts], 0), lr_rgb)
        assert torch.equal(torch.cat([p["lr_depth"] for p in parts], 0), lr_depth.reshape(-1))
        assert torch.equal(torch.cat([p["local"]["fine_comp_rgbs"] for p in parts], 0), hr_rgb)
    # world = 1 through the collective-free path of all_gather_pixels
    one = model.render_image_sharded(c2w, f, ndc, *nf)
    assert torch.equal(one["lr_rgb"], lr_rgb) and one["lr_range"] == (0, n_lr)
    # gather="hr": what the collective carries is the rendered pixels themselves (12 B per ray); the blocks, laid end to
    # end, ARE the ray-major tensor unflatten_reshape takes, so the assembled frame equals render_image's HR image
    hr_img = whole["hr_rgb"]
    assert hr_img.shape == (756, 1008, 3)
    for world in (2, 8):
        bounds = nsr_dist.shard_bounds(n_lr, world)
        parts = [model.render_image_sharded(c2w, f, ndc, *nf, lr_range=b, gather="hr") for b in bounds]
        hr_rays = torch.cat([p["hr_rays"] for p in parts], 0)
        assert torch.equal(hr_rays, hr_rgb)
        assert torch.equal(model.unflatten_reshape(hr_rays), hr_img)
        assert parts[0]["bytes_per_rank"] == (bounds[0][1] - bounds[0][0]) * 16 * 12
        assert abs(parts[0]["bytes_per_rank"] - 762048 * 12 / world) <= 16 * 12
    one = model.render_image_sharded(c2w, f, ndc, *nf, gather="hr")
    assert torch.equal(one["hr_rgb"], hr_img) and torch.equal(one["lr_rgb"], lr_rgb)


# ------------------------------------------------------------------------------------------------- config #5
def test_config5_composed_small_vs_oracles(ops):
    """render -> depth -> warp ('ray' depth) -> tiles -> refinement network -> stitch on a 128 x 128 <- 32 x 32
    Blender-like frame, every stage against its oracle fed with the device's own inputs, plus the end-to-end chain."""
    from nerf_sr_amd import pipeline, refine
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    from oracle import refine_oracle as ro, refine_tiler_oracle as rt, warp_oracle as wo
    W = H = 128
    s = 4
    c2w, ref_c2w = cameras.spheric_pose(40.0, -30.0, 4.0), cameras.spheric_pose(28.0, -30.0, 4.0)
    focal = cameras.blender_focal(W)
    sd_c, sd_f, sd_r = make_state_dict(99), make_state_dict(100), refine.make_refine_state_dict(7)
    opt = default_options(img_wh=(W, H), downscale=s, white_bkgd=True, precision="f16x3")
    model = NeRFDownXModel(opt).load_networks(sd_c, sd_f).eval()
    net = refine.MaxPoolingModel().load_state_dict(sd_r).eval()
    gen = torch.Generator().manual_seed(3)
    ref_img = torch.rand(3, H, W, generator=gen)
    out = pipeline.render_warp_refine(model, net, c2w, ref_c2w, ref_img.cuda(), focal, False, 2.0, 6.0)
    # stage 1: the render (colours and depth) vs the oracle
    rays = oc.subpixel_ray_grid(torch.from_numpy(c2w), H, W, focal, s, False, 2.0, 6.0).reshape(-1, 8)
    from tests.util import oracle_forward_parallel
    ref = oracle_forward_parallel(sd_c, sd_f, rays, True)              # 16,384 rays over the host's cores (VERDICT r5 "next" #5)
    hr_ref = oc.unflatten_hr(ref["fine_comp_rgbs"], H, W, s)
    d_ref = oc.unflatten_hr(ref["fine_depth"].reshape(-1, 1), H, W, s)[..., 0]
    assert float((out["hr_rgb"].cpu() - hr_ref).abs().max()) <= 2e-4          # a frame of 16k rays: see test 1 for the 1e-4 bar
    assert float((out["depth"].cpu() - d_ref).abs().max()) <= 2e-3
    # stage 2: the warp of the DEVICE's depth map: integer targets bit-exact with the oracle
    w2c = pipeline.world_to_camera(ref_c2w)
    locs_want = wo.depth_warp(out["depth"].cpu().numpy(), c2w, w2c, focal, "ray")
    assert np.array_equal(out["locs"].cpu().numpy(), locs_want)
    inside = (locs_want[..., 0] >= 0) & (locs_want[..., 0] < W) & (locs_want[..., 1] >= 0) & (locs_want[..., 1] < H)
    assert inside.mean() > 0.5                                                  # the two views overlap
    # ... and the end-to-end chain agrees on all but a few boundary pixels
    locs_ref = wo.depth_warp(d_ref.numpy(), c2w, w2c, focal, "ray")
    assert (locs_ref[..., :2] != locs_want[..., :2]).any(-1).mean() < 0.02
    # stage 3: tiles / gather / network / stitch on the device's own SR image and locs
    sr = (out["hr_rgb"].cpu().permute(2, 0, 1) * 2 - 1).numpy()
    starts, refs = rt.tile(locs_want, W, H, 64, 8)
    srp, refp = rt.gather(sr, (ref_img * 2 - 1).numpy(), starts, refs, 64)
    # the fp64 network oracle, one tile (1 + 8 patches) per host thread team: the tiles are independent
    from concurrent.futures import ThreadPoolExecutor
    import os

    def one_tile(i):
        torch.set_num_threads(max(1, (os.cpu_count() or 4) // (2 * srp.shape[0])))
        return ro.forward(sd_r, torch.from_numpy(srp[i:i + 1]), torch.from_numpy(refp[i:i + 1]), dtype=torch.float64).float().numpy()
    with ThreadPoolExecutor(srp.shape[0]) as ex:
        pred = np.concatenate(list(ex.map(one_tile, range(srp.shape[0]))), 0)
    want = (rt.stitch(pred, starts, 64, W, H) + 1.0) * 0.5
    assert float(np.abs(out["refined"].cpu().numpy() - want).max()) <= 2e-5
    assert refs.min() >= -1 and (refs >= 0).any()                               # warped reference patches were used
    # the caller-facing switch: 'metric' = the depth as it is (the reference's spheric_poses branch, warp.py:120-126)
    out_m = pipeline.render_warp_refine(model, net, c2w, ref_c2w, ref_img.cuda(), focal, False, 2.0, 6.0, depth_kind="metric")
    assert np.array_equal(out_m["locs"].cpu().numpy(), wo.depth_warp(out_m["depth"].cpu().numpy(), c2w, w2c, focal, "metric"))
    assert not np.array_equal(out_m["locs"].cpu().numpy(), locs_want)
    with pytest.raises(ValueError):
        pipeline.render_warp_refine(model, net, c2w, ref_c2w, ref_img.cuda(), focal, False, 2.0, 6.0, depth_kind="z")


def test_config5_full_size_end_to_end(ops):
    """800 x 800 <- 200 x 200 + refine pass: shapes, finiteness, the warp of a view into itself is the identity."""
    from nerf_sr_amd import pipeline, refine
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    wh, s, ndc, white = CONFIGS[5]
    c2w, focal, nf = _camera(5)
    opt = default_options(img_wh=wh, downscale=s, white_bkgd=white, precision="f16x3")
    model = NeRFDownXModel(opt).load_networks(make_state_dict(99), make_state_dict(100)).eval()
    net = refine.MaxPoolingModel().load_state_dict(refine.make_refine_state_dict(7)).eval()
    ref_img = torch.rand(3, wh[1], wh[0], generator=torch.Generator().manual_seed(1)).cuda()
    out = pipeline.render_warp_refine(model, net, c2w, c2w, ref_img, focal, ndc, *nf)
    assert out["hr_rgb"].shape == (800, 800, 3) and out["refined"].shape == (3, 800, 800) and out["lr_rgb"].shape == (40000, 3)
    assert torch.isfinite(out["refined"]).all() and float(out["refined"].min()) >= 0.0 and float(out["refined"].max()) <= 1.0
    assert 0.0 <= float(out["depth"].min()) and float(out["depth"].max()) <= 6.0 * 1.0001    # sum(w z), opacity <= 1
    xs = torch.arange(800, device="cuda", dtype=torch.float64)
    assert float((out["locs"][..., 0] == xs[None, :]).double().mean()) > 0.97
    assert float((out["locs"][..., 1] == xs[:, None]).double().mean()) > 0.97


# ------------------------------------------------------------------------------------------------- bench.py, N > 1
def test_bench_two_ranks_on_one_gpu_gloo():
    """`python bench.py --gpus 2`, started PLAINLY (no torchrun: bench.py launches its own ranks, one process per rank through
    torch.distributed.run on a free port -- round 5), on THIS box: both ranks share the one GPU and exchange over gloo
    (NSR_DIST_BACKEND; RCCL refuses two ranks per device).  Checks the N > 1 code path end to end: the SAME workload and
    metric string as N = 1 (config #2's frame cut in two LR-pixel blocks), config #4 cut the same way as the `config4`
    sub-object with the HR frame as its collective's payload (12 B per ray), one all-gather per step whose result is verified
    identical on both ranks, max-over-ranks timing, one JSON line from rank 0 with the contract's fields."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    one = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                          "--no-config4"], cwd=repo, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    # one workload per scaling curve: N = 2 reports the metric N = 1 reports, on the same frame
    assert d["metric"] == d1["metric"] == "rays/sec (64+128 samples, 2x SS)"
    assert d["config"]["rays_per_step"] == d1["config"]["rays_per_step"] == 190512
    assert d1["n_gpus"] == 1 and d1["scaling"] == "strong" and d1["collective"]["world"] == 1 and d1["collective"]["backend"] is None
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "config #2" in d["config"]["workload"] and "2 contiguous LR-pixel blocks" in d["config"]["workload"]
    assert abs(d["value"] - 190512 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 2e5 < d["value"] < 1e7                       # two ranks time-slicing one GPU: about the single-GPU rate
    assert d["roofline"]["bound"] == "mfma" and 0.0 < d["roofline"]["frac"] < 1.0 and d["cpu_baseline"] is None
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world"] == 2 and c["calls_per_step"] == 1 and c["bytes_per_rank"] == 23814 * 12
    assert c["result_identical_on_all_ranks"] is True and c["rccl_version"] is None          # gloo here; RCCL on a multi-GPU node
    c4 = d["config4"]
    assert "config #4" in c4["workload"] and c4["rays_per_step"] == 762048 and c4["metric"] == "rays/sec (64+128 samples, 4x SS)"
    assert c4["result_identical_on_all_ranks"] is True and abs(c4["value"] - 762048 / (c4["ms_per_step"] * 1e-3)) <= 1e-6 * c4["value"]
    assert c4["bytes_per_rank"] == 762048 * 12 // 2 and c4["hr_frame_assembled"] is True      # the HR frame, 12 B per ray
    assert d["non_mlp_ms_per_step"] > 0.0 and c4["non_mlp_ms_per_step"] > 0.0
    assert d["numerics_status"] == [0, 0]
    src = d["roofline"]["pmc_source"]
    assert set(src) == {"file", "csrc_sha256", "this_build_sha256", "matches_this_build"}
    assert d["roofline"]["traffic"] is None             # the counter constants belong to the unsharded config #2 launch


def test_bench_train_two_ranks_on_one_gpu_gloo():
    """The training twin of the test above (VERDICT r5 "next" #6): `python bench.py --mode train --gpus 2` started plainly on
    THIS box, both ranks on the one GPU, gradients exchanged over gloo.  The code path RCCL enters on a node: every rank its own
    ray batch (a different pose per rank), loss scaled by 1 / world, ONE all-reduce per network of the flat 2.4 MB gradient
    buffer per step, Adam replicated -- after 3 + 1 steps the weights are bit-identical on both ranks (checked by checksums
    inside the run, after the timed region).  Replaces DistributedDataParallel, models/networks.py:72-86 / train.py:154-156."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(NSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--mode", "train", "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--train-rays", "512", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=repo, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["rays_per_step"] == 1024 and d["config"]["parallelism"] == "data-parallel x2"
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world"] == 2 and c["calls_per_step"] == 2     # one per network
    assert 595844 * 4 <= c["bytes_per_call"] < 595844 * 4 + 24 * 16 and c["grad_scale"] == 0.5 and c["adam_steps"] == 4   # views 16-byte aligned
    assert c["weights_identical_on_all_ranks"] is True
    assert all(0.0 < x < 1.0 for x in d["losses"])
