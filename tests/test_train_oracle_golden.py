"""Pins the TRAINING oracle (oracle/train_oracle.py) to fixtures produced by the reference's own
optimize_parameters (tests/golden/make_golden_train.py): forward outputs, losses, every gradient tensor
(norm, sum, 512-element subsample) and the weights after one Adam step.  CPU only."""
import os

import numpy as np
import pytest
import torch

from nerf_sr_amd.weights import make_state_dict, STATE_DICT_SPEC
from oracle import nerf_oracle as oc
from oracle import train_oracle as tr
from tests.util import sample_idx, train_draws

CASES = ["llff_det", "llff_rand", "blender_rand", "blender_var", "llff_gamma", "blender_softplus", "llff_colornone", "blender_stopgrad"]
# blender_var: --use_var_loss --use_depth_var_loss; the last four: --gamma_correct, --sigma_activation softplus,
# --color_activation none, --stop_grad true
OPTIONS = {"llff_gamma": {"gamma_correct": True}, "blender_softplus": {"sigma_activation": "softplus"},
           "llff_colornone": {"color_activation": "none"}, "blender_stopgrad": {"stop_grad": True}}


@pytest.fixture(scope="module", params=CASES)
def case(request, golden_dir):
    g = np.load(os.path.join(golden_dir, f"train_{request.param}.npz"))
    sd_c, sd_f = make_state_dict(int(g["seed_coarse"])), make_state_dict(int(g["seed_fine"]))
    res, gc, gf = tr.loss_and_grads(sd_c, sd_f, g["rays"], g["target_lr"], int(g["s2"]), 64, 64,
                                    bool(g["white_bkgd"]), float(g["lambda_coarse"]), float(g["lambda_fine"]),
                                    lambda_var=(g["lambda_var"].tolist() if "lambda_var" in g else None),
                                    **OPTIONS.get(request.param, {}), **train_draws(g))
    return g, sd_c, res, gc, gf


def test_forward_and_losses(case):
    g, _, res, _, _ = case
    # same ATen ops on the same build: the restatement reproduces the reference to the last bits
    for k_ref, k in (("lr_coarse", "lr_coarse"), ("lr_fine", "lr_fine"), ("hr_coarse", "coarse_comp_rgbs"),
                     ("hr_fine", "fine_comp_rgbs"), ("fine_weights", "fine_weights")):
        np.testing.assert_allclose(res[k].numpy(), g[k_ref], rtol=0, atol=2e-6, err_msg=k)
    for k in ("loss_coarse_mse", "loss_fine_mse", "loss_tot"):
        assert abs(res[k] - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), k
    if "lambda_var" in g:      # the reference's loss_out_*_var / loss_*_depth_var (nerf_downX_model.py:332-336, 349-353)
        want = g["lambda_var"] * g["var_losses_raw"]
        np.testing.assert_allclose(np.array(res["var_losses"]), want, rtol=2e-6, atol=1e-9)
        assert float(g["loss_tot"]) > float(g["loss_coarse_mse"]) + float(g["loss_fine_mse"]) + 0.5 * float(want.sum())


def test_gradients(case):
    g, _, _, gc, gf = case
    for name, grads in (("coarse", gc), ("fine", gf)):
        for k in STATE_DICT_SPEC:
            got = grads[k].numpy().reshape(-1)
            want_norm = float(g[f"gnorm_{name}.{k}"])
            sub = got[sample_idx(got.size)]
            want = g[f"grad_{name}.{k}"]
            scale = max(float(np.abs(want).max()), 1e-12)
            assert np.abs(sub - want).max() <= 2e-5 * scale + 1e-9, (name, k)
            assert abs(float(np.linalg.norm(got.astype(np.float64))) - want_norm) <= 1e-5 * want_norm + 1e-9, (name, k)
            assert abs(float(got.astype(np.float64).sum()) - float(g[f"gsum_{name}.{k}"])) <= 1e-4 * want_norm + 1e-9


def test_adam_step(case):
    g, sd_c, _, gc, _ = case
    params = oc.to_torch_sd(sd_c)
    params = {k: v.clone() for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    tr.adam_step(params, gc, m, v, step=1, lr=float(g["lr"]), beta1=float(g["beta1"]))
    for k in STATE_DICT_SPEC:
        got = params[k].numpy().reshape(-1)
        np.testing.assert_allclose(got[sample_idx(got.size)], g[f"w1_coarse.{k}"], rtol=0, atol=2e-7, err_msg=k)
        w0 = sd_c[k].reshape(-1)
        dn = float(np.linalg.norm((got - w0).astype(np.float64)))
        assert abs(dn - float(g[f"dw_norm_coarse.{k}"])) <= 1e-4 * max(dn, 1e-12), k
