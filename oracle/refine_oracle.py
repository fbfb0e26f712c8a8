"""CPU oracle of the refinement network's forward pass -- test infrastructure only.

Restates ``MaxPoolingModel.forward`` in eval mode with torch functional ops on the 106-tensor state dict:
``Model_VNPCAT_Encoder.forward`` (models/networks.py:760-774), the max over the reference patches (:971-983) and
``Model_VNPCAT_Decoder.forward`` (:827-857).  Pinned by ``tests/golden/refine.npz`` (produced by the reference's own
module, ``make_golden_refine.py``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _cbr(sd, x, conv, bn=None, stride=1, act="relu"):
    y = F.conv2d(x, sd[f"{conv}.weight"], sd[f"{conv}.bias"], stride=stride, padding=1)
    if bn:
        y = F.batch_norm(y, sd[f"{bn}.running_mean"], sd[f"{bn}.running_var"], sd[f"{bn}.weight"], sd[f"{bn}.bias"],
                         training=False, eps=1e-5)
    return torch.relu(y) if act == "relu" else (torch.tanh(y) if act == "tanh" else y)


def encoder(sd, x):
    x1 = _cbr(sd, x, "E.conv1")
    x2 = _cbr(sd, x1, "E.conv2", "E.conv2_bnorm")
    x3 = _cbr(sd, x2, "E.conv3", "E.conv3_bnorm", 2)
    x4 = _cbr(sd, x3, "E.conv4", "E.conv4_bnorm")
    x5 = _cbr(sd, x4, "E.conv5", "E.conv5_bnorm", 2)
    x6 = _cbr(sd, x5, "E.conv6", "E.conv6_bnorm")
    x7 = _cbr(sd, x6, "E.conv7", "E.conv7_bnorm", 2)
    return [x2, x4, x6, x7]


def decoder(sd, fs, fm):
    """``fm = None``: Model_VNPCAT_Decoder_NoPooling (networks.py:906-935, --not_use_ref): no F_max_i in the concats."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    if fm is None:
        cat = lambda lvl, *xs: torch.cat(xs + (fs[lvl],), 1) if xs else fs[lvl]
    else:
        cat = lambda lvl, *xs: torch.cat(xs + (fs[lvl], fm[lvl]), 1)
    x1 = _cbr(sd, cat(3), "D.conv1", "D.conv1_bnorm")
    x2 = _cbr(sd, x1, "D.conv2", "D.conv2_bnorm")
    x2u = _cbr(sd, up(x2), "D.conv2_up", "D.conv2_up_bnorm")
    x3 = _cbr(sd, cat(2, x2u), "D.conv3", "D.conv3_bnorm")
    x4 = _cbr(sd, x3, "D.conv4", "D.conv4_bnorm")
    x4u = _cbr(sd, up(x4), "D.conv4_up", "D.conv4_up_bnorm")
    x5 = _cbr(sd, cat(1, x4u), "D.conv5", "D.conv5_bnorm")
    x6 = _cbr(sd, x5, "D.conv6", "D.conv6_bnorm")
    x6u = _cbr(sd, up(x6), "D.conv6_up", "D.conv6_up_bnorm")
    x7 = _cbr(sd, cat(0, x6u), "D.conv7", "D.conv7_bnorm")
    x8 = _cbr(sd, x7, "D.conv8", "D.conv8_bnorm")
    return _cbr(sd, x8, "D.conv9", None, 1, "tanh")


def forward(sd_np, x_synth, x_candi, dtype=torch.float32, return_features=False):
    """x_synth (B, 3, H, W), x_candi (B, R, 3, H, W) -> (B, 3, H, W); ``x_candi=None``: the --not_use_ref model
    (MaxPoolingModel.forward, networks.py:963-969)."""
    sd = {k: torch.as_tensor(v).to(dtype) for k, v in sd_np.items()}
    x = torch.as_tensor(x_synth).to(dtype)
    fs = encoder(sd, x)
    if x_candi is None:
        y = decoder(sd, fs, None)
        return (y, fs, None) if return_features else y
    c = torch.as_tensor(x_candi).to(dtype)
    B, R = c.shape[:2]
    fc = encoder(sd, c.reshape(B * R, *c.shape[2:]))
    fm = [f.view(B, R, *f.shape[1:]).max(1)[0] for f in fc]
    y = decoder(sd, fs, fm)
    return (y, fs, fm) if return_features else y
