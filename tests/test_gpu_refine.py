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
