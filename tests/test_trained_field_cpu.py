"""CPU checks of the analytic scenes the trained-field GPU tests learn (tests/trained_field.py): the target colours must be
what ray / sphere intersection says, for both families, through the oracle's own ray construction."""
import numpy as np
import torch

from oracle import nerf_oracle as oc
from tests import trained_field as tf


def _ray(o, d, near, far):
    return torch.tensor([[*o, *d, near, far]], dtype=torch.float32)


def test_blender_scene_hits_and_misses():
    # straight at the big sphere at the origin (radius 0.75) from z = +4: hit at t = 3.25, normal +z
    c = tf.analytic_colours(_ray((0, 0, 4), (0, 0, -1), 2.0, 6.0), "blender")[0]
    light = np.array(tf.LIGHT) / np.linalg.norm(tf.LIGHT)
    shade = 0.25 + 0.75 * max(0.0, float(np.dot([0, 0, 1], -light)))
    assert np.allclose(c.numpy(), np.array(tf.SPHERES_WORLD[0][2]) * shade, atol=1e-6)
    # a ray that passes everything: the white background; a sphere beyond `far` is not seen either
    assert torch.equal(tf.analytic_colours(_ray((0, 3.5, 4), (0, 0, -1), 2.0, 6.0), "blender")[0], torch.ones(3))
    assert torch.equal(tf.analytic_colours(_ray((0, 0, 4), (0, 0, -1), 2.0, 3.0), "blender")[0], torch.ones(3))
    # un-normalised directions are fine: t is measured in units of |d|, like z_vals
    c2 = tf.analytic_colours(_ray((0, 0, 4), (0, 0, -2), 1.0, 3.0), "blender")[0]
    assert np.allclose(c2.numpy(), c.numpy(), atol=1e-6)


def test_llff_scene_every_ray_ends_on_something_opaque():
    wh, s, ndc, white, nf, _ = tf.FAMILIES["llff"]
    c2w, focal = tf.eval_pose("llff")
    rays = oc.subpixel_ray_grid(torch.from_numpy(c2w), wh[1], wh[0], focal, s, ndc, *nf).reshape(-1, 8)[::97]
    col = tf.analytic_colours(rays, "llff")
    assert col.shape == (rays.shape[0], 3) and bool(((col >= 0) & (col <= 1)).all())
    # NDC rays run from z = -1 to z = +1: the wall at z = 0.9 is reached at t = 0.95 by every ray that misses the spheres
    o, d = rays[:, 0:3].double(), rays[:, 3:6].double()
    t_wall = (tf.WALL_Z_NDC - o[:, 2]) / d[:, 2]
    assert float((t_wall - 0.95).abs().max()) < 1e-6
    # a fair share of the frame sees a sphere (otherwise the scene would not exercise surfaces in front of the wall)
    p = o + t_wall[:, None] * d
    check = ((torch.floor(p[:, 0] * 4.0) + torch.floor(p[:, 1] * 4.0)) % 2 == 0).double()[:, None]
    tone = 0.5 + 0.5 * torch.stack([torch.sin(3.0 * p[:, 0]), torch.cos(2.0 * p[:, 1]), torch.sin(p[:, 0] + p[:, 1])], -1)
    wall = (check * (0.25 + 0.5 * tone) + (1 - check) * (0.85 - 0.35 * tone)).float()
    on_sphere = (col - wall).abs().max(-1)[0] > 1e-6
    assert 0.15 < float(on_sphere.float().mean()) < 0.85


def test_layer_abs_max_matches_the_oracle_network():
    from nerf_sr_amd.weights import make_state_dict
    sd = make_state_dict(5)
    x = torch.rand(64, 90) * 2 - 1
    m = tf.layer_abs_max(sd, x)
    out = oc.mlp_forward(oc.to_torch_sd(sd, torch.float64), x.double())
    assert abs(m["sigma"] - float(out[:, 3].abs().max())) < 1e-9
    assert m["input"] <= 1.0 and m["max_weight"] == max(float(np.abs(v).max()) for k, v in sd.items() if k.endswith("weight"))
