"""On-disk formats, CPU only: COLMAP binaries and the LLFF pose pipeline against what the REFERENCE's
readers / read_meta made of the same files (tests/golden/io_colmap.npz), checkpoints and Blender transforms round
trips."""
import json
import os

import numpy as np
import pytest
import torch

from nerf_sr_amd import io as nio
from nerf_sr_amd.weights import make_state_dict, STATE_DICT_SPEC


@pytest.fixture(scope="module")
def scene(golden_dir, tmp_path_factory):
    g = np.load(os.path.join(golden_dir, "io_colmap.npz"))
    d = tmp_path_factory.mktemp("scene") / "sparse" / "0"
    d.mkdir(parents=True)
    for fn in ("cameras.bin", "images.bin", "points3D.bin"):
        (d / fn).write_bytes(g["file_" + fn].tobytes())
    return g, str(d)


def test_colmap_readers_match_reference(scene):
    g, d = scene
    cams = nio.read_cameras_binary(os.path.join(d, "cameras.bin"))
    assert list(cams) == [1] and cams[1].model == "SIMPLE_RADIAL"
    assert np.array_equal(cams[1].params, g["cam_params"]) and [cams[1].width, cams[1].height] == list(g["cam_wh"])
    imgs = nio.read_images_binary(os.path.join(d, "images.bin"))
    ids = sorted(imgs)
    assert ids == list(g["img_ids"])
    assert [imgs[k].name for k in ids] == list(g["img_names"])
    assert np.array_equal(np.stack([imgs[k].qvec for k in ids]), g["img_qvec"])
    assert np.array_equal(np.stack([imgs[k].tvec for k in ids]), g["img_tvec"])
    np.testing.assert_allclose(np.stack([imgs[k].qvec2rotmat() for k in ids]), g["img_rot"], rtol=0, atol=1e-15)
    assert [len(imgs[k].point3D_ids) for k in ids] == list(g["img_npts"])
    pts = nio.read_points3d_binary(os.path.join(d, "points3D.bin"))
    pids = sorted(pts)
    assert np.array_equal(np.stack([pts[k].xyz for k in pids]), g["pt_xyz"])
    assert [len(pts[k].image_ids) for k in pids] == list(g["pt_track_len"])
    assert [int(pts[k].image_ids[0]) for k in pids] == list(g["pt_first_image"])


def test_llff_pose_pipeline_matches_reference(scene):
    g, d = scene
    s = nio.llff_scene_from_colmap(d, int(g["W"]))
    assert s["focal"] == float(g["focal"])
    assert s["names"] == list(g["image_paths"])
    np.testing.assert_allclose(s["poses"], g["poses"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(s["bounds"], g["bounds"], rtol=0, atol=1e-12)
    assert abs(s["bounds"].min() - 1 / 0.75) < 1e-12        # nearest depth at 1.33 after rescaling


def test_checkpoint_round_trip(tmp_path):
    sd = make_state_dict(5)
    pc, pf = nio.checkpoint_paths(str(tmp_path), "exp", 30)
    assert pc.endswith("exp/30_net_Coarse.pth") and pf.endswith("exp/30_net_Fine.pth")
    nio.save_network_state(sd, pc)
    back = nio.load_network_state(pc)
    assert list(back) == list(STATE_DICT_SPEC)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)
    # a DataParallel-saved file ('module.' prefix) with an extra integer buffer loads too
    torch.save({"module." + k: torch.from_numpy(v) for k, v in sd.items()} | {"module.steps": torch.tensor(3)}, pf)
    back = nio.load_network_state(pf)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)
    bad = dict(sd)
    bad.pop("sigma.bias")
    nio.save_network_state(bad, pc)
    with pytest.raises(KeyError):
        nio.load_network_state(pc)


def test_blender_transforms(tmp_path):
    frames = [{"file_path": f"./train/r_{i}", "transform_matrix": (np.eye(4) + i).tolist()} for i in range(3)]
    p = tmp_path / "transforms_train.json"
    p.write_text(json.dumps({"camera_angle_x": 0.6911112070083618, "frames": frames}))
    s = nio.load_blender_transforms(str(p), 400)
    assert abs(s["focal"] - 0.5 * 800 / np.tan(0.5 * 0.6911112070083618) * 0.5) < 1e-9     # 555.555 at 400 px
    assert s["poses"].shape == (3, 3, 4) and np.array_equal(s["poses"][2], (np.eye(4) + 2)[:3])
    assert (s["near"], s["far"]) == (2.0, 6.0) and s["files"][1] == "./train/r_1"


def test_lr_schedules_match_torch_lambda_lr():
    """Trainer.update_learning_rate restates get_scheduler (models/networks.py:89-118): compare with torch's LambdaLR /
    StepLR driven the way the reference drives them (host-side logic only; no GPU needed)."""
    import math
    from types import SimpleNamespace
    from nerf_sr_amd.train import Trainer
    opt = SimpleNamespace(lr=5e-4, lr_final=5e-6, n_epochs=30, n_epochs_decay=10, lr_decay_epochs=10, lr_decay_gamma=0.1)
    t = Trainer.__new__(Trainer)            # schedule logic only
    t.lr = opt.lr
    for policy in ("exp", "linear", "step"):
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.Adam([{"params": [p], "initial_lr": opt.lr}], lr=opt.lr)
        if policy == "step":
            sch = torch.optim.lr_scheduler.StepLR(o, step_size=opt.lr_decay_epochs, gamma=opt.lr_decay_gamma)
        else:
            def rule(epoch, policy=policy):
                tt = max(0, epoch + 1 - opt.n_epochs + opt.n_epochs_decay) / float(opt.n_epochs_decay + 1)
                lr = opt.lr * (1 - tt) + opt.lr_final * tt if policy == "linear" else \
                    math.exp(math.log(opt.lr) * (1 - tt) + math.log(opt.lr_final) * tt)
                return lr / opt.lr
            sch = torch.optim.lr_scheduler.LambdaLR(o, lr_lambda=rule)
        for epoch in range(1, 32):
            o.step()
            sch.step()
            got = t.update_learning_rate(epoch, policy, opt.n_epochs, opt.n_epochs_decay, opt.lr_final, opt.lr_decay_epochs,
                                         opt.lr_decay_gamma, lr_initial=opt.lr)
            assert abs(got - o.param_groups[0]["lr"]) <= 1e-12 * opt.lr + 1e-18, (policy, epoch, got, o.param_groups[0]["lr"])
