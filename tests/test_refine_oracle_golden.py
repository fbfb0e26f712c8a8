"""Pins oracle/refine_oracle.py to the fixture produced by the reference's own MaxPoolingModel
(tests/golden/make_golden_refine.py).  CPU only."""
import os

import numpy as np

from nerf_sr_amd.refine import make_refine_state_dict, REFINE_SPEC
from oracle import refine_oracle as ro


def test_refine_forward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    sd = make_refine_state_dict(int(g["seed"]))
    assert list(sd) == list(REFINE_SPEC)
    for tag in ("a", "b"):
        y, fs, _ = ro.forward(sd, g[f"x_{tag}"], g[f"c_{tag}"], return_features=True)
        # same ATen convolutions on the same build
        np.testing.assert_allclose(y.numpy(), g[f"y_{tag}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(fs[3].numpy(), g[f"f3_{tag}"], rtol=0, atol=1e-5)


def test_state_dict_layout_matches_reference(golden_dir):
    """REFINE_SPEC = the float tensors of the reference's MaxPoolingModel.state_dict(), same order and shapes (this is
    the order nsr_refine_pack_weights expects)."""
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    assert list(REFINE_SPEC) == list(g["float_keys"])
    assert [str(tuple(s)) for s in REFINE_SPEC.values()] == list(g["float_shapes"])


def test_not_use_ref_matches_reference(golden_dir):
    """--not_use_ref (Model_VNPCAT_Decoder_NoPooling): fixture from the reference's own module; spec shapes included."""
    from nerf_sr_amd.refine import REFINE_SPEC_NOREF
    g = np.load(os.path.join(golden_dir, "refine.npz"))
    assert [str(tuple(s)) for s in REFINE_SPEC_NOREF.values()] == list(g["noref_float_shapes"])
    sd = make_refine_state_dict(int(g["noref_seed"]), not_use_ref=True)
    y = ro.forward(sd, g["x_noref"], None)
    np.testing.assert_allclose(y.numpy(), g["y_noref"], rtol=0, atol=1e-6)
