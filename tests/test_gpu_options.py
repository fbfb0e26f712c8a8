"""The three option values of the path that no script of the reference turns on, through the C ABI, against fixtures made
by the reference's own ``forward`` with the option set (tests/golden/make_golden_options.py) and against the oracle:

  --no_dir                      models/networks.py:128, 160-169, 213-216   dir_encoding sees xyz_encoding_final alone
  --color_activation none       models/networks.py:173-180                 the rgb head ends in nn.Identity
  --sigma_activation softplus   models/rendering.py:69-73                  density = log(1 + exp(sigma - 1))

``color_activation`` is a bit of the packed network's option word (``nsr_weights_set_options``), ``softplus`` a bit of the
compositing entry points' ``white_bkgd`` word (include/nsr.h), ``no_dir`` is host-side: the narrow layer is packed as the
full layer with 27 zero columns.  Rounds 3-4 refused all three loudly (tests/test_options.py).  The training step takes
them too (and --gamma_correct), as bits of its own option word (include/nsr_train.h).
"""
import os

import numpy as np
import pytest
import torch

from nerf_sr_amd.weights import make_state_dict, DIR_W, STATE_DICT_SPEC
from oracle import nerf_oracle as oc
from oracle import train_oracle as to
from tests.util import train_draws

pytestmark = pytest.mark.gpu

CASES = {"no_dir": {"no_dir": True}, "color_none": {"color_activation": "none"}, "softplus": {"sigma_activation": "softplus"}}
OUT_KEYS = ("coarse_comp_rgbs", "coarse_depth", "coarse_opacity", "coarse_weights",
            "fine_comp_rgbs", "fine_depth", "fine_opacity", "fine_weights")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from nerf_sr_amd import ops as _ops
    return _ops


def _narrow(sd):
    sd = dict(sd)
    sd[DIR_W] = sd[DIR_W][:, :256].copy()
    return sd


def _networks(p, case):
    sds = [make_state_dict(int(p["seed_coarse"])), make_state_dict(int(p["seed_fine"]))]
    return [_narrow(sd) for sd in sds] if case == "no_dir" else sds


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("tag,white", [("llff", False), ("blender", True)])
@pytest.mark.parametrize("case", list(CASES))
def test_option_matches_the_reference(ops, golden_dir, case, tag, white, prec):
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    g = np.load(os.path.join(golden_dir, "options.npz"))
    p = np.load(os.path.join(golden_dir, f"path_{tag}.npz"))
    n = int(g["n_rays"])
    r = torch.from_numpy(p["rays"])[:n].cuda()
    sd_c, sd_f = _networks(p, case)
    m = NeRFDownXModel(default_options(white_bkgd=white, precision=prec, **CASES[case])).load_networks(sd_c, sd_f).eval()
    m.set_input({"rays": r[None]})
    m.forward()                                   # the fused route: one enqueue sequence, compositing inside the MLP launches
    fused = {k: getattr(m, f"out_{k}").cpu() for k in OUT_KEYS}
    # colours without the sigmoid are not confined to [0, 1]: the fixture's reach |rgb| ~ 4, the tolerance scales with them
    tol = 1e-4 * max(1.0, float(np.abs(g[f"{case}_{tag}_fine_comp_rgbs"]).max()))
    for k in ("coarse_comp_rgbs", "fine_comp_rgbs", "coarse_opacity", "fine_opacity"):
        assert float((fused[k] - torch.from_numpy(g[f"{case}_{tag}_{k}"])).abs().max()) <= tol, k
    assert float((fused["coarse_weights"] - torch.from_numpy(g[f"{case}_{tag}_coarse_weights"])).abs().max()) <= 1e-5
    assert float((fused["coarse_depth"] - torch.from_numpy(g[f"{case}_{tag}_coarse_depth"])).abs().max()) <= 1e-4
    # the oracle with the same option (pinned to the same fixture on the CPU, tests/test_oracle_golden.py)
    kw = {k: v for k, v in CASES[case].items() if k != "no_dir"}
    ref = oc.forward_rays(oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), r.cpu(), 64, 64, white, **kw)
    assert float((fused["fine_comp_rgbs"] - ref["fine_comp_rgbs"]).abs().max()) <= tol
    # per-sample values through the unfused render_rays route + the stand-alone compositor: same bits as the fused launch
    z, _ = ops.sample_along_rays(r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8], 64, False, False)
    rgb, sig = ops.render_rays(m.netCoarse, r, z)
    assert float((rgb[:16].cpu() - torch.from_numpy(g[f"{case}_{tag}_coarse_point_rgb"])).abs().max()) <= (2e-5 if case != "color_none" else 1e-4)
    assert float((sig[:16].cpu() - torch.from_numpy(g[f"{case}_{tag}_coarse_point_sigma"])).abs().max()) <= 1e-3
    comp, depth, opac, w = m.renderer(rgb.contiguous(), sig.contiguous(), z, white)
    assert torch.equal(comp.cpu(), fused["coarse_comp_rgbs"]) and torch.equal(w.cpu(), fused["coarse_weights"])
    # ... and the option does something
    plain = NeRFDownXModel(default_options(white_bkgd=white, precision=prec)).load_networks(
        make_state_dict(int(p["seed_coarse"])), make_state_dict(int(p["seed_fine"]))).eval()
    plain.set_input({"rays": r[None]})
    plain.forward()
    assert float((plain.out_fine_comp_rgbs.cpu() - fused["fine_comp_rgbs"]).abs().max()) > 1e-3
    assert m.netCoarse.status() == 0 and m.netFine.status() == 0


def test_colour_head_options_combine_and_survive_repacking(ops, golden_dir):
    """gamma_correct and color_activation are bits of ONE option word: setting one must not clear the other, and
    load_state_dict (which re-packs and clears the word) re-applies both."""
    from types import SimpleNamespace
    p = np.load(os.path.join(golden_dir, "path_llff.npz"))
    x = torch.from_numpy(p["mlp_in_512"]).cuda()
    sd = make_state_dict(int(p["seed_coarse"]))
    ref = oc.mlp_forward(oc.to_torch_sd(sd), x.cpu(), color_activation="none")
    net = ops.VanillaMLP(SimpleNamespace(color_activation="none"), precision="fp32").load_state_dict(sd)
    assert float((net(x).cpu() - ref).abs().max()) <= 2e-5
    net.set_gamma_correct(True)        # pow(rgb, 1 / 2.2) of the raw head output: NaN for negative values, like torch.pow
    got = net(x).cpu()
    want = torch.pow(ref[:, :3], 1 / 2.2)
    ok = torch.isfinite(want)
    assert bool((torch.isfinite(got[:, :3]) == ok).all()) and float((got[:, :3][ok] - want[ok]).abs().max()) <= 1e-4
    net.set_gamma_correct(False)
    assert float((net(x).cpu() - ref).abs().max()) <= 2e-5          # 'none' survived both writes
    net.load_state_dict(make_state_dict(7))
    assert float((net(x).cpu() - oc.mlp_forward(oc.to_torch_sd(make_state_dict(7)), x.cpu(), color_activation="none")).abs().max()) <= 2e-5
    net.status(clear=True)


def test_narrow_network_round_trips(ops):
    sd = _narrow(make_state_dict(11))
    from types import SimpleNamespace
    net = ops.VanillaMLP(SimpleNamespace(no_dir=True), precision="f16x3").load_state_dict(sd)
    back = net.state_dict()
    assert tuple(back[DIR_W].shape) == (128, 256) and np.array_equal(back[DIR_W].cpu().numpy(), sd[DIR_W])
    with pytest.raises(ValueError, match="no_dir"):
        net.load_state_dict(make_state_dict(11))            # a full network into a no_dir object


@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_training_a_no_dir_network(golden_dir, prec):
    """The training step on a --no_dir pair: losses and gradients against the training oracle run on the NARROW networks
    (autograd through oracle/nerf_oracle.py, whose no_dir forward is pinned to the reference's fixture); the padded columns
    get no gradient, so Adam never moves them and the exported state_dict is the narrow one."""
    from nerf_sr_amd import train as tr
    g = np.load(os.path.join(golden_dir, "train_llff_rand.npz"))
    sd_c, sd_f = _narrow(make_state_dict(int(g["seed_coarse"]))), _narrow(make_state_dict(int(g["seed_fine"])))
    t = tr.Trainer(sd_c, sd_f, white_bkgd=bool(g["white_bkgd"]), downscale=int(round(int(g["s2"]) ** 0.5)),
                   randomized=bool(g["randomized"]), noise_std=float(g["noise_std"]), lr=float(g["lr"]), beta1=float(g["beta1"]),
                   lambda_coarse_mse=float(g["lambda_coarse"]), lambda_fine_mse=float(g["lambda_fine"]), precision=prec, no_dir=True)
    t.set_input(torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["target_lr"]).cuda())
    draws = {k: v for k, v in train_draws(g).items() if k != "noise_std"}
    t.loss_and_grads(draws)
    res, gc, gf = to.loss_and_grads(sd_c, sd_f, g["rays"], g["target_lr"], int(g["s2"]), 64, 64, bool(g["white_bkgd"]),
                                    float(g["lambda_coarse"]), float(g["lambda_fine"]), dtype=torch.float64, **train_draws(g))
    losses = t.losses.cpu().numpy()
    assert abs(losses[0] - res["loss_coarse_mse"]) < 1e-6 and abs(losses[1] - res["loss_fine_mse"]) < 2e-6
    for n, ref in enumerate((gc, gf)):
        for k in STATE_DICT_SPEC:
            got = t.grads[n][k].cpu().double()
            if k == DIR_W:
                assert not bool(got[:, 256:].any())
                got = got[:, :256]
            err, nrm = float((got - ref[k]).norm()), float(ref[k].norm())
            assert err <= 2e-3 * nrm + 1e-9, (n, k, err / nrm)
    for _ in range(3):
        t.optimize_parameters(draws)
    for n in range(2):
        assert not bool(t.params[n][DIR_W][:, 256:].any())
    out = t.state_dicts()
    assert tuple(out[0][DIR_W].shape) == (128, 256) and not torch.equal(out[0][DIR_W].cpu(), torch.from_numpy(sd_c[DIR_W]))
    assert t.status() == 0


def test_training_option_word_is_checked(golden_dir):
    """Unknown bits of the step's option word are NSR_ERR_INVALID_ARG; --gamma_correct with --color_activation none (the
    power of an unbounded head: NaN for every negative value) is refused by the host mirror and by the C entry point."""
    from nerf_sr_amd import train as tr, _lib
    with pytest.raises(ValueError, match="gamma_correct"):
        tr.Trainer(make_state_dict(1), make_state_dict(2), gamma_correct=True, color_activation="none")
    with pytest.raises(ValueError):
        tr.Trainer(make_state_dict(1), make_state_dict(2), sigma_activation="elu")
    g = np.load(os.path.join(golden_dir, "train_llff_det.npz"))
    t = tr.Trainer(make_state_dict(1), make_state_dict(2), randomized=False)
    t.set_input(torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["target_lr"]).cuda())
    for word, code in ((32, "-1"), (_lib.NSR_TRAIN_GAMMA_CORRECT | _lib.NSR_TRAIN_COLOR_NONE, "-2")):
        t.white_bkgd = word                      # int(self.white_bkgd) is what reaches the C ABI
        with pytest.raises(_lib.NsrError, match=f"nsr_status {code}"):
            t.loss_and_grads({})


TRAIN_OPTIONS = {"llff_gamma": {"gamma_correct": True}, "blender_softplus": {"sigma_activation": "softplus"},
                 "llff_colornone": {"color_activation": "none"}, "blender_stopgrad": {"stop_grad": True}}


@pytest.mark.parametrize("prec", ["f16x3", "fp32", "f16x3_gemm"])
@pytest.mark.parametrize("name", list(TRAIN_OPTIONS))
def test_training_with_the_option(golden_dir, name, prec):
    """--gamma_correct (render_rays, models/nerf_downX_model.py:271-276), --sigma_activation softplus
    (models/rendering.py:69-73), --color_activation none (models/networks.py:173-180) and --stop_grad true (:218-219: zero
    gradients for xyz_encoding_final, exactly) in the TRAINING step: one
    optimize_parameters of the reference with the option on (tests/golden/train_<name>.npz) -- losses, forward outputs, every
    gradient tensor against the reference's digests and the fp64 oracle, all three implementations of the step."""
    from nerf_sr_amd import train as tr
    from tests.util import sample_idx
    opts = TRAIN_OPTIONS[name]
    g = np.load(os.path.join(golden_dir, f"train_{name}.npz"))
    sd_c, sd_f = make_state_dict(int(g["seed_coarse"])), make_state_dict(int(g["seed_fine"]))
    t = tr.Trainer(sd_c, sd_f, white_bkgd=bool(g["white_bkgd"]), downscale=int(round(int(g["s2"]) ** 0.5)),
                   randomized=bool(g["randomized"]), noise_std=float(g["noise_std"]), lr=float(g["lr"]), beta1=float(g["beta1"]),
                   lambda_coarse_mse=float(g["lambda_coarse"]), lambda_fine_mse=float(g["lambda_fine"]), precision=prec, **opts)
    t.set_input(torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["target_lr"]).cuda())
    draws = {k: v for k, v in train_draws(g).items() if k != "noise_std"}
    t.loss_and_grads(draws)
    losses = t.losses.cpu().numpy()
    scale = max(1.0, float(np.abs(g["hr_fine"]).max()))          # colours without the sigmoid are not confined to [0, 1]
    assert abs(losses[0] - float(g["loss_coarse_mse"])) < 1e-6 * scale * scale and abs(losses[1] - float(g["loss_fine_mse"])) < 2e-6 * scale * scale
    np.testing.assert_allclose(t.out["coarse_comp_rgbs"].cpu().numpy(), g["hr_coarse"], rtol=0, atol=2e-6 * scale)
    np.testing.assert_allclose(t.out["lr_fine"].cpu().numpy(), g["lr_fine"], rtol=0, atol=1e-4 * scale)
    _, gc, gf = to.loss_and_grads(sd_c, sd_f, g["rays"], g["target_lr"], int(g["s2"]), 64, 64, bool(g["white_bkgd"]),
                                  float(g["lambda_coarse"]), float(g["lambda_fine"]), dtype=torch.float64, **opts, **train_draws(g))
    for n, (net, ref) in enumerate((("coarse", gc), ("fine", gf))):
        for k in STATE_DICT_SPEC:
            got = t.grads[n][k].cpu().double()
            err, nrm = float((got - ref[k]).norm()), float(ref[k].norm())
            assert err <= 2e-3 * nrm + 1e-9, (net, k, err / nrm)
            ref_norm = float(g[f"gnorm_{net}.{k}"])
            assert abs(float(got.norm()) - ref_norm) <= 2e-3 * ref_norm + 1e-9, (net, k)
            sub = got.reshape(-1).numpy()[sample_idx(got.numel())]
            want_sub = g[f"grad_{net}.{k}"].astype(np.float64)
            assert np.linalg.norm(sub - want_sub) <= 4e-3 * np.linalg.norm(want_sub) + 1e-9, (net, k)
    # the option changes the step (same inputs without it)
    t0 = tr.Trainer(sd_c, sd_f, white_bkgd=bool(g["white_bkgd"]), randomized=True, noise_std=float(g["noise_std"]), precision=prec)
    t0.set_input(torch.from_numpy(g["rays"]).cuda(), torch.from_numpy(g["target_lr"]).cuda())
    t0.loss_and_grads(draws)
    if name == "blender_stopgrad":      # a detach: same forward, other gradients -- xyz_encoding_final's exactly zero
        assert abs(float(t0.losses[0]) - float(losses[0])) < 1e-6
        for n in range(2):
            assert not bool(t.grads[n]["xyz_encoding_final.weight"].any()) and not bool(t.grads[n]["xyz_encoding_final.bias"].any())
            assert float((t.grads[n]["xyz_encoding_8.0.weight"] - t0.grads[n]["xyz_encoding_8.0.weight"]).norm()) > \
                0.05 * float(t0.grads[n]["xyz_encoding_8.0.weight"].norm())
    else:
        assert abs(float(t0.losses[0]) - float(losses[0])) > 1e-3
    assert t.status() == 0


ARCH_CASES = {"small": ({"D": 4, "W": 128, "skips": (2,), "deg_pos": 6, "deg_dir": 2}, "llff", False),
              "odd": ({"D": 6, "W": 192, "skips": (1, 3), "deg_pos": 10, "deg_dir": 4}, "blender", True)}


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("case", list(ARCH_CASES))
def test_architecture_flags_run_layer_by_layer(ops, golden_dir, case, prec):
    """--D --W --skips --deg_pos --deg_dir (models/networks.py:124-157, models/nerf_model.py:53-57): no script of the
    reference changes them and the fused kernels are laid out for the defaults, so such a network runs nn.Linear by
    nn.Linear on the fp32-MFMA GEMM (ops.GenericMLP) behind the same model interface -- against the reference's own forward
    for two non-default networks (tests/golden/arch.npz) and the oracle.  Round 6: under precision 'f16x3' the same layers run
    on the split-fp16 MFMA (nsr_linear_f16x3: three terms per product, weights split once at load time) -- same bounds."""
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    from nerf_sr_amd.weights import make_state_dict_arch
    g = np.load(os.path.join(golden_dir, "arch.npz"))
    arch, tag, white = ARCH_CASES[case]
    p = np.load(os.path.join(golden_dir, f"path_{tag}.npz"))
    sd_c, sd_f = make_state_dict_arch(int(g["seed_coarse"]), **arch), make_state_dict_arch(int(g["seed_fine"]), **arch)
    m = NeRFDownXModel(default_options(white_bkgd=white, precision=prec, **{**arch, "skips": list(arch["skips"])})).load_networks(sd_c, sd_f).eval()
    assert isinstance(m.netCoarse, ops.GenericMLP) and not m.fused and m.netCoarse.precision == prec
    assert (len(m.netCoarse._split) > 0) == (prec == "f16x3")
    x = torch.from_numpy(g[f"{case}_mlp_in_256"]).cuda()
    assert float((m.netCoarse(x).cpu() - torch.from_numpy(g[f"{case}_mlp_out_256"])).abs().max()) <= 2e-5
    assert float((m.netCoarse(x[:64], sigma_only=True).cpu() - torch.from_numpy(g[f"{case}_mlp_sigma_only_64"])).abs().max()) <= 2e-5
    r = torch.from_numpy(p["rays"])[:int(g["n_rays"])].cuda()
    m.set_input({"rays": r[None]})
    m.forward()
    for k in ("coarse_comp_rgbs", "fine_comp_rgbs", "coarse_opacity", "fine_opacity"):
        assert float((getattr(m, f"out_{k}").cpu() - torch.from_numpy(g[f"{case}_{k}"])).abs().max()) <= 1e-4, k
    assert float((m.out_coarse_weights.cpu() - torch.from_numpy(g[f"{case}_coarse_weights"])).abs().max()) <= 1e-5
    ref = oc.forward_rays(oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), r.cpu(), 64, 64, white, deg_pos=arch["deg_pos"], deg_dir=arch["deg_dir"])
    assert float((m.out_fine_comp_rgbs.cpu() - ref["fine_comp_rgbs"]).abs().max()) <= 1e-4
    # chunked rows give the same bits (a GEMM row does not depend on its neighbours), the state_dict round-trips, NaN is reported
    big = x.repeat(5, 1)
    m.netCoarse.POINT_CHUNK = 512
    assert torch.equal(m.netCoarse(big)[:256], m.netCoarse(x))
    assert all(torch.equal(v.cpu(), torch.from_numpy(sd_c[k])) for k, v in m.netCoarse.state_dict().items())
    bad = x.clone()
    bad[3, 0] = float("nan")
    m.netCoarse(bad)
    from nerf_sr_amd import _lib
    with pytest.raises(_lib.NsrNumericsError):
        m.netCoarse.check()
    assert m.netCoarse.status() == 0        # check() cleared it


def test_default_architecture_keeps_the_fused_kernels(ops):
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    m = NeRFDownXModel(default_options(precision="f16x3"))
    assert m.fused and isinstance(m.netCoarse, ops.VanillaMLP) and not isinstance(m.netCoarse, ops.GenericMLP)
    with pytest.raises(ValueError, match="outside the built path"):
        ops.VanillaMLP(default_options(W=128))            # the fused class itself still refuses what it is not laid out for


def test_generic_mlp_warns_once_and_applies_gamma(ops, golden_dir):
    """Round 6: a non-default architecture leaves the fused kernels SILENTLY no longer -- one
    RuntimeWarning per architecture names the layer-by-layer route -- and --gamma_correct, which the reference applies in
    render_rays for any architecture (models/nerf_downX_model.py:271-276), works there too: against the oracle with the option
    on, and equal to pow(., 1 / 2.2) of the colours without it."""
    import warnings
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    from nerf_sr_amd.weights import make_state_dict_arch
    arch = {"D": 4, "W": 128, "skips": (2,), "deg_pos": 6, "deg_dir": 2}
    g = np.load(os.path.join(golden_dir, "arch.npz"))
    p = np.load(os.path.join(golden_dir, "path_llff.npz"))
    sd_c, sd_f = make_state_dict_arch(int(g["seed_coarse"]), **arch), make_state_dict_arch(int(g["seed_fine"]), **arch)
    ops._GENERIC_WARNED.clear()
    with pytest.warns(RuntimeWarning, match="layer-by-layer fp32 GEMM"):
        m = NeRFDownXModel(default_options(gamma_correct=True, **{**arch, "skips": [2]})).load_networks(sd_c, sd_f).eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")             # the second model of the same architecture does not warn again
        m0 = NeRFDownXModel(default_options(**{**arch, "skips": [2]})).load_networks(sd_c, sd_f).eval()
    r = torch.from_numpy(p["rays"])[:64].cuda()
    ref = oc.forward_rays(oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), r.cpu(), 64, 64, False, gamma_correct=True,
                          deg_pos=arch["deg_pos"], deg_dir=arch["deg_dir"])
    m.set_input({"rays": r[None]})
    m.forward()
    for k in ("coarse_comp_rgbs", "fine_comp_rgbs"):
        assert float((getattr(m, f"out_{k}").cpu() - ref[k]).abs().max()) <= 1e-4, k
    z, xyz = ops.sample_along_rays(r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8], 64, False, False)
    de = m.embeddings["dir"](r[:, 3:6].contiguous())
    rgb_g, sig_g = m.render_rays(m.netCoarse, xyz, de)
    rgb_0, sig_0 = m0.render_rays(m0.netCoarse, xyz, de)
    assert torch.equal(sig_g, sig_0) and torch.equal(rgb_g, torch.pow(rgb_0, 1.0 / 2.2))
