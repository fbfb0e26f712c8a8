"""N3 refinement network of the HIP path (include/nsr_refine.h) against the fixture produced by the reference's own
MaxPoolingModel and against the CPU oracle, in both arithmetic modes (fp32 MFMA with explicit im2col; split-fp16 MFMA
with implicit im2col).  BatchNorm is folded into the weights; the output (tanh, |y| < 1) is held to 2e-5 absolute
(19 layers of re-associated fp32 sums; the split-fp16 products add ~2^-21 relative)."""
import os

import numpy as np
import pytest
import torch

from nerf_sr_amd.refine import make_refine_state_dict
from oracle import refine_oracle as ro

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module", params=["f16x3", "fp32"])
def net(request):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from nerf_sr_amd import refine as _r
    return _r.MaxPoolingModel(precision=request.param).load_state_dict(make_refine_state_dict(7)).eval()


def test_refine_vs_reference_fixture(net, golden_dir):
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    for tag in ("a", "b"):
        y = net(torch.from_numpy(g[f"x_{tag}"]).cuda(), torch.from_numpy(g[f"c_{tag}"]).cuda())
        err = float((y.cpu() - torch.from_numpy(g[f"y_{tag}"])).abs().max())
        assert err <= TOL, (tag, err)


def test_refine_patch_64_vs_oracle(net):
    """The reference's patch size (64 x 64, 8 reference patches, refine_model / llff_refine_dataset), batch of 2;
    batch invariance; argument checking."""
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 64, 64, generator=gen) * 2 - 1
    c = torch.rand(2, 8, 3, 64, 64, generator=gen) * 2 - 1
    y = net(x.cuda(), c.cuda())
    want = ro.forward(make_refine_state_dict(7), x, c, dtype=torch.float64)
    assert float((y.cpu().double() - want).abs().max()) <= TOL
    y0 = net(x[:1].cuda(), c[:1].cuda())
    assert torch.equal(y0, y[:1])
    with pytest.raises(ValueError):
        net(x[:, :, :60].cuda(), c[:, :, :, :60].cuda())
    assert net(x[:0].cuda(), c[:0].cuda()).shape == (0, 3, 64, 64)


def test_refine_nan_travels_like_torch(net):
    """A NaN input pixel (synthesised patch, and one reference patch) poisons exactly the outputs torch poisons: nn.ReLU keeps
    NaN, torch.max over the reference patches propagates it (networks.py:980-983), tanh(NaN) = NaN.  The epilogues' ReLU
    was fmaxf (NaN -> 0: a diverged feature came out finite) until round 4."""
    gen = torch.Generator().manual_seed(21)
    x = torch.rand(2, 3, 64, 64, generator=gen) * 2 - 1
    c = torch.rand(2, 8, 3, 64, 64, generator=gen) * 2 - 1
    x[0, 1, 5, 60] = float("nan")
    c[1, 3, 2, 40, 7] = float("nan")
    y = net(x.cuda(), c.cuda()).cpu()
    want = ro.forward(make_refine_state_dict(7), x, c, dtype=torch.float64)
    assert bool(torch.isnan(want).any()) and not bool(torch.isnan(want).all())
    assert torch.equal(torch.isnan(y), torch.isnan(want))
    ok = ~torch.isnan(want)
    assert float((y.double() - want)[ok].abs().max()) <= TOL


def test_refine_batches_around_the_eight_image_blocks(net):
    """Round 6: the 8 x 8-pixel plain layers (decoder conv1 / conv2, the last encoder layer over the synthesised patches) run
    eight IMAGES per workgroup block (conv_halo_kernel MULTI, csrc/nsr_gemm_f16.hip); a batch that is not a multiple of eight
    ends in a block whose missing images are gathered clamped and never stored.  Batches of 7, 8 and 9 patch sets: every set
    gives the bits it gives alone, and the set behind the block boundary agrees with the oracle."""
    gen = torch.Generator().manual_seed(33)
    x = torch.rand(9, 3, 64, 64, generator=gen) * 2 - 1
    c = torch.rand(9, 8, 3, 64, 64, generator=gen) * 2 - 1
    alone = [net(x[i:i + 1].cuda(), c[i:i + 1].cuda()) for i in range(9)]
    for n in (7, 8, 9):
        y = net(x[:n].cuda(), c[:n].cuda())
        for i in range(n):
            assert torch.equal(y[i:i + 1], alone[i]), (n, i)
    want = ro.forward(make_refine_state_dict(7), x[8:9], c[8:9], dtype=torch.float64)
    assert float((alone[8].cpu().double() - want).abs().max()) <= TOL


@pytest.mark.parametrize("hw", [(128, 64), (64, 96), (48, 80)])
def test_refine_other_patch_shapes_vs_oracle(net, hw):
    """Patches that are not the reference's 64 x 64 square: several spatial blocks per image in x and in y (the LDS-patch
    convolution kernel tiles an image into 16 x 16 / 16 x 32 / 8 x 4 / 8 x 8 pixel blocks, csrc/nsr_gemm_f16.hip), image
    borders on some sides of a block only, and (48 x 80) scales whose size is not a multiple of the block -- those layers
    take the staged kernels, the others the patch kernel, inside one forward pass."""
    H, W = hw
    gen = torch.Generator().manual_seed(11 + H)
    x = torch.rand(1, 3, H, W, generator=gen) * 2 - 1
    c = torch.rand(1, 8, 3, H, W, generator=gen) * 2 - 1
    y = net(x.cuda(), c.cuda())
    want = ro.forward(make_refine_state_dict(7), x, c, dtype=torch.float64)
    assert float((y.cpu().double() - want).abs().max()) <= TOL
    # two patch sets in one call: the same bits per patch set (the kernel choice depends on the shape, never on the batch)
    y2 = net(torch.cat([x, x]).cuda(), torch.cat([c, c]).cuda())
    assert torch.equal(y2[0], y[0]) and torch.equal(y2[1], y[0])


def test_same_bits_per_patch_set_at_the_default_tile_batch(net):
    """refine.py's default tile batch, 256 patch sets of the reference's shape (2,048 reference patches): the 128-channel
    full-resolution input planes of the reference encoder then span exactly 2^32 bytes, the limit of the LDS-patch kernel's
    32-bit offsets -- round 4 silently dropped that layer to the staged kernel (another K order: other bits) at this batch
    size only.  The entry point now cuts the batch below the limit, so a patch set gives the bits it gives alone."""
    gen = torch.Generator().manual_seed(21)
    x = (torch.rand(4, 3, 64, 64, generator=gen) * 2 - 1).cuda()
    c = (torch.rand(4, 8, 3, 64, 64, generator=gen) * 2 - 1).cuda()
    alone = net(x, c)
    big = net(x.repeat(64, 1, 1, 1), c.repeat(64, 1, 1, 1, 1))          # 256 patch sets: sets 0..3 repeated
    assert big.shape[0] == 256
    for i in (0, 1, 2, 3, 127, 252, 253, 254, 255):
        assert torch.equal(big[i], alone[i % 4]), i


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_not_use_ref_vs_reference_fixture_and_oracle(prec, golden_dir):
    """--not_use_ref (Model_VNPCAT_Decoder_NoPooling, networks.py:866-945): fixture from the reference's own module, and
    the oracle at the reference's 64 x 64 patch size."""
    from nerf_sr_amd import refine as _r
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    sd = make_refine_state_dict(int(g["noref_seed"]), not_use_ref=True)
    net = _r.MaxPoolingModel(precision=prec, not_use_ref=True).load_state_dict(sd).eval()
    y = net(torch.from_numpy(g["x_noref"]).cuda())
    assert float((y.cpu() - torch.from_numpy(g["y_noref"])).abs().max()) <= TOL
    x = torch.rand(3, 3, 64, 64, generator=torch.Generator().manual_seed(4)) * 2 - 1
    want = ro.forward(sd, x, None, dtype=torch.float64)
    got = net(x.cuda(), None)
    assert float((got.cpu().double() - want).abs().max()) <= TOL
    with pytest.raises(ValueError):                      # a with-reference state dict does not fit the narrower layers
        _r.MaxPoolingModel(precision=prec, not_use_ref=True).load_state_dict(make_refine_state_dict(7))


def test_tiler_gather_stitch_vs_reference_fixture(golden_dir):
    """Patch tiler around the network: bit-exact with what the reference's own dataset class and stitching loop
    produced (tests/golden/refine_tiler.npz), and with the oracle on a full-size random case."""
    from nerf_sr_amd import refine as r
    from oracle import refine_tiler_oracle as rt
    g = np.load(os.path.join(golden_dir, "refine_tiler.npz"))
    W, H, P, NR = int(g["W"]), int(g["H"]), int(g["patch_len"]), int(g["num_ref"])
    ref_img = torch.from_numpy(g["ref_img"]).cuda()
    for img in range(2):
        starts, refs = r.tile_refs(torch.from_numpy(g[f"locs_{img}"]).cuda(), P, NR)
        assert np.array_equal(starts.cpu().numpy(), g[f"start_locs_{img}"].astype(np.int32))
        sr, ref = r.gather_patches(torch.from_numpy(g["sr_imgs"][img]).cuda(), ref_img, starts, refs, P)
        assert np.array_equal(sr.cpu().numpy(), g[f"sr_patch_{img}"])
        assert np.array_equal(ref.cpu().numpy(), g[f"ref_patches_{img}"])
        out = r.stitch_patches(torch.from_numpy(g[f"pred_{img}"]).cuda(), starts, (W, H))
        assert np.array_equal(out.cpu().numpy(), g[f"stitched_{img}"])
    # the reference's sizes: 504 x 378, 64-pixel tiles, 8 references
    rng = np.random.default_rng(2)
    W, H = 504, 378
    locs = np.stack([rng.integers(-200, W + 200, (H, W)), rng.integers(-200, H + 200, (H, W)), -np.ones((H, W))], -1).astype(np.float64)
    locs[:100, :130] = -1.0
    starts, refs = r.tile_refs(torch.from_numpy(locs).cuda(), 64, 8)
    want_s, want_r = rt.tile(locs, W, H, 64, 8)
    assert np.array_equal(starts.cpu().numpy(), want_s) and np.array_equal(refs.cpu().numpy(), want_r)


def test_refine_image_end_to_end(net):
    """tile -> network -> stitch on a small frame equals the same steps done through the oracle's tiler."""
    from nerf_sr_amd import refine as r
    from oracle import refine_tiler_oracle as rt
    rng = np.random.default_rng(4)
    W, H, P, NR = 80, 48, 32, 4
    sr = (rng.random((3, H, W)) * 2 - 1).astype(np.float32)
    ref = (rng.random((3, H, W)) * 2 - 1).astype(np.float32)
    locs = np.stack([rng.integers(-20, W + 20, (H, W)), rng.integers(-20, H + 20, (H, W)), -np.ones((H, W))], -1).astype(np.float64)
    got = r.refine_image(net, torch.from_numpy(sr).cuda(), torch.from_numpy(ref).cuda(), torch.from_numpy(locs).cuda(), P, NR, batch=4)
    starts, refs = rt.tile(locs, W, H, P, NR)
    srp, refp = rt.gather(sr, ref, starts, refs, P)
    pred = ro.forward(make_refine_state_dict(7), srp, refp, dtype=torch.float64).numpy()
    want = rt.stitch(pred, starts, P, W, H)
    assert float(np.abs(got.cpu().numpy() - want).max()) <= TOL
