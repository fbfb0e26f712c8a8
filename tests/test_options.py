"""Host logic (no GPU): option values of the reference that the built path does not implement must raise, never be ignored
(VERDICT r3 "missing" #3: ``color_activation='none'`` used to be accepted and silently rendered with the sigmoid head).
Round 5: ``color_activation='none'``, ``no_dir`` and ``sigma_activation='softplus'`` are built for the render path
(tests/test_gpu_options.py, tests/golden/options.npz) -- they are accepted there and still refused by the training step.
Reference: models/networks.py:124-128,160-173 (VanillaMLP options), models/rendering.py:69-73 (sigma_activation)."""
from types import SimpleNamespace

import pytest

from nerf_sr_amd import ops
from nerf_sr_amd.model import NeRFDownXModel, default_options


@pytest.mark.parametrize("kw", [
    {"color_activation": "tanh"}, {"no_dir": 2}, {"D": 6}, {"W": 128}, {"skips": [4, 6]}, {"skips": []},
    {"deg_pos": 8}, {"deg_dir": 2}, {"dim_rgb": 4}, {"stop_grad": "yes"},
])
def test_unbuilt_mlp_options_raise(kw):
    with pytest.raises(ValueError, match="outside the built path"):
        ops.check_mlp_options(SimpleNamespace(**kw))
    with pytest.raises(ValueError, match="outside the built path"):
        ops.VanillaMLP(SimpleNamespace(**kw), precision="fp32")          # before any device is touched
    if set(kw) & {"color_activation", "no_dir", "stop_grad"}:      # not architecture: no class serves a malformed value
        with pytest.raises(ValueError):
            NeRFDownXModel(default_options(**kw))


def test_architecture_flags_select_the_layer_by_layer_class():
    """--D --W --skips / degrees / dim_rgb (round 5): refused by the fused-kernel class, served by ops.GenericMLP through
    ops.make_mlp (host logic only here: constructing either class touches no device memory beyond the packed blob)."""
    from nerf_sr_amd.weights import arch_of, arch_spec, is_default_arch, make_state_dict_arch, check_state_dict_arch
    assert is_default_arch(arch_of(None)) and is_default_arch(arch_of(default_options())) and is_default_arch(arch_of(SimpleNamespace(no_dir=True)))
    a = arch_of(SimpleNamespace(D=4, W=128, skips=[2], deg_pos=6, deg_dir=2))
    assert not is_default_arch(a)
    spec = arch_spec(**a)
    assert spec["xyz_encoding_3.0.weight"] == (128, 128 + 39) and spec["dir_encoding.0.weight"] == (64, 128 + 15) and len(spec) == 16
    sd = make_state_dict_arch(3, **a)
    check_state_dict_arch(sd, **a)
    with pytest.raises(ValueError):
        check_state_dict_arch(sd, **{**a, "W": 64})
    for bad in ({"D": 0}, {"W": 5}, {"skips": (0,)}, {"skips": (9,)}, {"dim_rgb": 0}):
        with pytest.raises(ValueError):
            arch_spec(**{**a, **bad})
    ops.check_mlp_options(SimpleNamespace(D=4, W=128), fused=False)
    with pytest.raises(ValueError, match="outside the built path"):
        ops.check_mlp_options(SimpleNamespace(dim_pos=2), fused=False)


def test_reference_defaults_pass():
    ops.check_mlp_options(None)
    ops.check_mlp_options(default_options())
    ops.check_mlp_options(SimpleNamespace(D=8, W=256, skips=(4,), no_dir=False, color_activation="sigmoid"))
    ops.check_mlp_options(SimpleNamespace(unrelated=1))
    ops.check_mlp_options(SimpleNamespace(no_dir=True, color_activation="none"))     # options of VanillaMLP since round 5


def test_density_activations():
    from nerf_sr_amd import _lib
    assert ops.VolumetricRenderer(SimpleNamespace(sigma_activation="softplus")).sigma_activation == "softplus"
    assert ops.VolumetricRenderer(SimpleNamespace(sigma_activation="relu")).sigma_activation == "relu"
    assert ops.VolumetricRenderer(None).sigma_activation == "relu"
    with pytest.raises(ValueError, match="sigma_activation='elu'"):
        ops.VolumetricRenderer(SimpleNamespace(sigma_activation="elu"))
    # the renderer option word of include/nsr.h
    assert ops.renderer_flags(False) == 0 and ops.renderer_flags(True) == _lib.NSR_WHITE_BKGD == 1
    assert ops.renderer_flags(True, "softplus") == (_lib.NSR_WHITE_BKGD | _lib.NSR_SIGMA_SOFTPLUS) == 3


def test_model_rejects_unknown_density_activation_before_touching_a_device():
    with pytest.raises(ValueError):
        NeRFDownXModel(default_options(sigma_activation="elu"), device="cuda")


def test_no_dir_state_dict_shapes():
    """--no_dir (models/networks.py:160-169): dir_encoding.0.weight is (128, 256); the host pads it with 27 zero columns."""
    import numpy as np
    from nerf_sr_amd.weights import check_state_dict, make_state_dict, pad_no_dir, DIR_W
    sd = make_state_dict(3)
    narrow = dict(sd)
    narrow[DIR_W] = sd[DIR_W][:, :256].copy()
    check_state_dict(narrow, no_dir=True)
    with pytest.raises(ValueError, match="no_dir"):
        check_state_dict(sd, no_dir=True)
    with pytest.raises(ValueError):
        check_state_dict(narrow)
    padded = pad_no_dir(narrow[DIR_W])
    assert padded.shape == (128, 283) and np.array_equal(padded[:, :256], narrow[DIR_W]) and not padded[:, 256:].any()


def test_training_option_word():
    """render_rays applies rgb ** (1 / 2.2) in training too (models/nerf_downX_model.py:271-276): rounds 3-4 refused the
    option (the step had no such branch); since round 5 it is NSR_TRAIN_GAMMA_CORRECT of the step's option word
    (include/nsr_train.h; parity in tests/test_gpu_options.py against the reference's own --gamma_correct iteration)."""
    import re
    from nerf_sr_amd import _lib
    here = __import__("os").path.dirname(__import__("os").path.abspath(__file__))
    hdr = open(__import__("os").path.join(here, "..", "include", "nsr_train.h")).read()
    assert int(re.search(r"#define NSR_TRAIN_GAMMA_CORRECT (\d+)", hdr).group(1)) == _lib.NSR_TRAIN_GAMMA_CORRECT == 4
    assert int(re.search(r"#define NSR_TRAIN_COLOR_NONE (\d+)", hdr).group(1)) == _lib.NSR_TRAIN_COLOR_NONE == 8
    assert int(re.search(r"#define NSR_TRAIN_STOP_GRAD (\d+)", hdr).group(1)) == _lib.NSR_TRAIN_STOP_GRAD == 16
    top = open(__import__("os").path.join(here, "..", "include", "nsr.h")).read()
    assert int(re.search(r"#define NSR_WHITE_BKGD (\d+)", top).group(1)) == _lib.NSR_WHITE_BKGD
    assert int(re.search(r"#define NSR_SIGMA_SOFTPLUS (\d+)", top).group(1)) == _lib.NSR_SIGMA_SOFTPLUS
    assert int(re.search(r"#define NSR_OPT_COLOR_NONE (\d+)u", top).group(1)) == _lib.NSR_OPT_COLOR_NONE


def test_training_precision_values_match_the_header_and_size_the_chain_workspace():
    """Round 6 (include/nsr_train.h): the backward chain's arithmetic is a precision value of the training entry points --
    NSR_F16X3_BWD3 / _BWD2 / _BWD1 / _BWDM -- and the Python mirror's names map onto exactly those numbers.  All five chain values
    (NSR_F16X3 is the default among them) take the chain path's workspace, NSR_F16X3_GEMM / NSR_FP32 the GEMM path's; an
    unknown value sizes nothing.  No compute: the library answers from its arguments (CPU box)."""
    import os
    import re
    from nerf_sr_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "nsr_train.h")).read()
    for name, key in (("NSR_F16X3_GEMM", "f16x3_gemm"), ("NSR_F16X3_BWD3", "f16x3_bwd3"), ("NSR_F16X3_BWD2", "f16x3_bwd2"),
                      ("NSR_F16X3_BWD1", "f16x3_bwd1"), ("NSR_F16X3_BWDM", "f16x3_bwdm")):
        assert int(re.search(rf"#define {name} (\d+)", hdr).group(1)) == _lib.TRAIN_PRECISIONS[key]
    assert len(set(_lib.TRAIN_PRECISIONS.values())) == len(_lib.TRAIN_PRECISIONS)
    lib = _lib.load()
    chain = lib.nsr_train_workspace_bytes_for(_lib.NSR_F16X3, 2048, 64, 64)
    assert chain > 0
    for key in ("f16x3_bwd3", "f16x3_bwd2", "f16x3_bwd1", "f16x3_bwdm"):
        assert lib.nsr_train_workspace_bytes_for(_lib.TRAIN_PRECISIONS[key], 2048, 64, 64) == chain
    gemm = lib.nsr_train_workspace_bytes_for(_lib.NSR_F16X3_GEMM, 2048, 64, 64)
    assert gemm == lib.nsr_train_workspace_bytes_for(_lib.NSR_FP32, 2048, 64, 64) and gemm != chain
    assert lib.nsr_train_workspace_bytes_for(23, 2048, 64, 64) == 0
