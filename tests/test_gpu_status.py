"""The boundary's error convention on the GPU (``pytest -m gpu``): the numerics status word of a packed network, checked
weight packing, and the ``--gamma_correct`` epilogue.

The reference traps NaN colours with ``if torch.isnan(out_rgbs).any(): pdb.set_trace()`` inside ``render_rays``
(models/nerf_downX_model.py:273-274) and applies ``pow(rgb, 1/2.2)`` right there when ``--gamma_correct`` is set
(:271-276).  The replacement reports instead of trapping (SURVEY 8b "Error conventions"): every launch ORs sticky flags
into the blob's status word, ``nsr_weights_status`` reads it, ``ops.forward_rays(check=True)`` / ``model.forward()`` raise.
"""
import os

import numpy as np
import pytest
import torch

from nerf_sr_amd import _lib
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc

pytestmark = pytest.mark.gpu

WEIGHT, INPUT, ACT, OUTPUT = 1, 2, 4, 8


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from nerf_sr_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def rays(golden_dir):
    return torch.from_numpy(np.load(os.path.join(golden_dir, "path_llff.npz"))["rays"]).cuda()


def _scaled(sd, factor, keys):
    out = {k: v.copy() for k, v in sd.items()}
    for k in keys:
        out[k] = (out[k] * np.float32(factor)).astype(np.float32)
    return out


TRUNK = [f"xyz_encoding_{i}.0.weight" for i in range(1, 9)]


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "f16", "bf16"])
def test_healthy_network_leaves_the_status_word_clear(ops, rays, prec):
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(99))
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(100))
    ops.forward_rays(net_c, net_f, rays, 64, 64, False, check=True)        # every fused / unfused route of the precision
    z, _ = ops.sample_along_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8], 64, False, False)
    ops.render_rays(net_c, rays, z)
    x = torch.cat([ops.PositionalEncoding(3, 10)(torch.rand(300, 3, device="cuda")),
                   ops.PositionalEncoding(3, 4)(torch.rand(300, 3, device="cuda"))], -1)
    net_c(x)
    net_c(x, sigma_only=True)
    assert net_c.status() == 0 and net_f.status() == 0


@pytest.mark.parametrize("prec", ["fp32", "f16x3", "f16", "bf16"])
def test_non_finite_ray_raises_input_flag_and_is_sticky(ops, rays, prec):
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(99))
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(100))
    bad = rays.clone()
    bad[17, 4] = float("nan")
    ops.forward_rays(net_c, net_f, bad, 64, 64, False)
    ops.forward_rays(net_c, net_f, rays, 64, 64, False)                    # a healthy frame afterwards: the flag stays
    assert net_c.status() & INPUT
    with pytest.raises(_lib.NsrNumericsError, match="INPUT_RANGE") as e:
        ops.forward_rays(net_c, net_f, bad, 64, 64, False, check=True)
    assert e.value.flags & INPUT and isinstance(e.value, FloatingPointError)
    assert net_c.status() == 0                                             # check() clears
    inf_ray = rays.clone()
    inf_ray[3, 0] = float("inf")
    ops.forward_rays(net_c, net_f, inf_ray, 64, 64, False)
    assert net_c.status(clear=True) & INPUT and net_c.status() == 0


def test_diverged_trunk_trips_activation_range_in_f16x3_not_in_fp32(ops, rays):
    """Trunk weights x 1e3: hidden activations grow by 1e3 per layer.  The split-fp16 path must say so (its hi part
    saturates at 1023.75); the fp32 path carries 1e24 without trouble and reports nothing."""
    sd_c, sd_f = _scaled(make_state_dict(99), 1e3, TRUNK[:4]), _scaled(make_state_dict(100), 1e3, TRUNK[:4])
    nets = {}
    for prec in ("fp32", "f16x3"):
        nets[prec] = (ops.VanillaMLP(precision=prec).load_state_dict(sd_c), ops.VanillaMLP(precision=prec).load_state_dict(sd_f))
    out32 = ops.forward_rays(*nets["fp32"], rays, 64, 64, False, check=True)
    assert bool(torch.isfinite(out32["fine_comp_rgbs"]).all())
    with pytest.raises(_lib.NsrNumericsError, match="ACTIVATION_RANGE.*precision='fp32'"):
        ops.forward_rays(*nets["f16x3"], rays, 64, 64, False, check=True)
    # the flag comes from every route through the kernel: embedded rows, sigma_only, the unfused render
    net = nets["f16x3"][0]
    assert net.status() == 0
    x = torch.cat([ops.PositionalEncoding(3, 10)(torch.rand(256, 3, device="cuda")),
                   ops.PositionalEncoding(3, 4)(torch.rand(256, 3, device="cuda"))], -1)
    net(x)
    assert net.status(clear=True) == ACT
    net(x, sigma_only=True)
    assert net.status(clear=True) == ACT
    # a modest blow-up (x 3 on four layers: |h| ~ 1e2 -> 1e4 x 64 > 65,520) trips it as well, results still finite
    mild = ops.VanillaMLP(precision="f16x3").load_state_dict(_scaled(make_state_dict(99), 6.0, TRUNK[:4]))
    out = mild(x)
    assert mild.status() & ACT and bool(torch.isfinite(out).all())


def test_fp32_overflow_reports_non_finite_outputs(ops, rays):
    """fp32 carries anything up to 3e38; beyond that inf - inf = NaN reaches sigma, and relu keeps NaN like torch.relu."""
    sd = _scaled(make_state_dict(99), 1e6, TRUNK)
    net = ops.VanillaMLP(precision="fp32").load_state_dict(sd)
    x = torch.cat([ops.PositionalEncoding(3, 10)(torch.rand(256, 3, device="cuda")),
                   ops.PositionalEncoding(3, 4)(torch.rand(256, 3, device="cuda"))], -1)
    out = net(x)
    want = oc.mlp_forward(oc.to_torch_sd(sd), x.cpu())
    assert not bool(torch.isfinite(want).all())                             # the reference's own arithmetic overflows here
    agree = (torch.isfinite(out.cpu()) == torch.isfinite(want)).float().mean()
    assert not bool(torch.isfinite(out).all()) and float(agree) > 0.95      # ... and the kernel hides none of it (inf vs NaN
    #                                                                         can depend on the summation order)
    assert net.status() & OUTPUT


def test_checked_packing_rejects_weights_outside_the_operand_range(ops):
    sd = make_state_dict(99)
    big = {k: v.copy() for k, v in sd.items()}
    big["xyz_encoding_3.0.weight"][5, 7] = 1024.0                            # 2^6 w = 65,536 > fp16 max
    with pytest.raises(_lib.NsrNumericsError, match="1023.75"):
        ops.VanillaMLP(precision="f16x3").load_state_dict(big)
    for prec in ("fp32", "bf16", "f16"):
        assert ops.VanillaMLP(precision=prec).load_state_dict(big).status() == 0
    edge = {k: v.copy() for k, v in sd.items()}
    edge["xyz_encoding_3.0.weight"][5, 7] = 1023.0                           # the largest magnitudes still carried
    edge["sigma.bias"][0] = 3.0e5                                           # biases stay fp32: only finiteness matters
    assert ops.VanillaMLP(precision="f16x3").load_state_dict(edge).status() == 0
    huge = {k: v.copy() for k, v in sd.items()}
    huge["dir_encoding.0.weight"][0, 0] = 7.0e4
    with pytest.raises(_lib.NsrNumericsError):
        ops.VanillaMLP(precision="f16").load_state_dict(huge)
    for key in ("rgb.0.weight", "xyz_encoding_8.0.bias"):
        nan = {k: v.copy() for k, v in sd.items()}
        nan[key].reshape(-1)[1] = np.nan
        for prec in ("fp32", "f16x3", "f16", "bf16"):
            with pytest.raises(_lib.NsrNumericsError, match="non-finite"):
                ops.VanillaMLP(precision=prec).load_state_dict(nan)


def test_async_packing_leaves_the_verdict_in_the_status_word(ops):
    from ctypes import c_void_p, c_uint, byref
    lib = _lib.load()
    sd = make_state_dict(99)
    sd["xyz_encoding_2.0.weight"][0, 0] = -2048.0
    dev = [torch.from_numpy(v).cuda() for v in sd.values()]
    ptrs = (c_void_p * 24)(*[c_void_p(t.data_ptr()) for t in dev])
    blob = torch.empty(lib.nsr_packed_weights_bytes(_lib.NSR_F16X3), dtype=torch.uint8, device="cuda")
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.nsr_pack_weights_async(ptrs, c_void_p(blob.data_ptr()), _lib.NSR_F16X3, st) == 0
    flags = c_uint(0)
    assert lib.nsr_weights_status(c_void_p(blob.data_ptr()), _lib.NSR_F16X3, 0, byref(flags), st) == 0 and flags.value == WEIGHT
    assert lib.nsr_pack_weights(ptrs, c_void_p(blob.data_ptr()), _lib.NSR_F16X3, st) == _lib.NSR_ERR_RANGE
    blob32 = torch.empty(lib.nsr_packed_weights_bytes(_lib.NSR_FP32), dtype=torch.uint8, device="cuda")
    assert lib.nsr_pack_weights(ptrs, c_void_p(blob32.data_ptr()), _lib.NSR_FP32, st) == 0     # fp32 carries -2048 as it is


def test_model_forward_raises_like_the_reference_traps(ops, rays):
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    sd_c, sd_f = _scaled(make_state_dict(99), 1e3, TRUNK[:4]), make_state_dict(100)
    m = NeRFDownXModel(default_options(precision="f16x3")).load_networks(sd_c, sd_f).eval()
    m.set_input({"rays": rays[None]})
    with pytest.raises(FloatingPointError, match="coarse network"):
        m.forward()
    quiet = NeRFDownXModel(default_options(precision="f16x3", check_numerics=False)).load_networks(sd_c, sd_f).eval()
    quiet.set_input({"rays": rays[None]})
    quiet.forward()                                                         # no host check: the flag waits in the blob
    assert quiet.netCoarse.status() & ACT and quiet.netFine.status() == 0
    m.train()
    torch.manual_seed(0)
    with pytest.raises(FloatingPointError):
        m.forward()                                                         # the randomized stage-by-stage route checks too


# ------------------------------------------------------------------------------------------- --gamma_correct
@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("tag,white", [("llff", False), ("blender", True)])
def test_gamma_correct_matches_the_reference(ops, golden_dir, prec, tag, white):
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    g = np.load(os.path.join(golden_dir, "gamma.npz"))
    p = np.load(os.path.join(golden_dir, f"path_{tag}.npz"))
    n = int(g[f"{tag}_n_rays"])
    r = torch.from_numpy(p["rays"])[:n].cuda()
    opt = default_options(white_bkgd=white, gamma_correct=True, precision=prec)
    m = NeRFDownXModel(opt).load_networks(make_state_dict(int(p["seed_coarse"])), make_state_dict(int(p["seed_fine"]))).eval()
    m.set_input({"rays": r[None]})
    m.forward()
    for k in ("coarse_comp_rgbs", "fine_comp_rgbs", "coarse_opacity", "fine_opacity"):
        assert float((getattr(m, f"out_{k}").cpu() - torch.from_numpy(g[f"{tag}_{k}"])).abs().max()) <= 1e-4, k
    assert float((m.out_coarse_weights.cpu() - torch.from_numpy(g[f"{tag}_coarse_weights"])).abs().max()) <= 1e-5
    # per-sample colours through the unfused render_rays route
    z, _ = ops.sample_along_rays(r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8], 64, False, False)
    rgb, sig = ops.render_rays(m.netCoarse, r, z)
    assert float((rgb[:16].cpu() - torch.from_numpy(g[f"{tag}_coarse_point_rgb"])).abs().max()) <= 2e-5
    # ... and switching the option off again restores the plain colours (it is an option of the blob, not of the build)
    m.netCoarse.set_gamma_correct(False)
    rgb0, _ = ops.render_rays(m.netCoarse, r, z)
    assert float((rgb0[:16].cpu() - torch.from_numpy(p["coarse_point_rgb"])).abs().max()) <= 2e-5


# ------------------------------------------------------------------------------------------- HIP graph capture
@pytest.mark.parametrize("prec", ["f16x3", "fp32"])
def test_forward_rays_is_hip_graph_capturable(ops, rays, prec):
    """The library only ENQUEUES on the caller's stream (include/nsr.h): no allocation, no synchronisation, no host-side
    state -- so one forward_rays call can be captured into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed on new
    ray data.  This is what a caller with the reference's small 4,096-ray chunks needs (launch-bound there); the replay must
    be bit-identical to the eager call, and the numerics flags raised by a replay land in the same status words."""
    net_c = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(99))
    net_f = ops.VanillaMLP(precision=prec).load_state_dict(make_state_dict(100))
    R = rays.shape[0]
    static_rays = rays.clone()
    ws = torch.empty(max(_lib.load().nsr_forward_rays_workspace_bytes_for(net_c._prec, R, 64, 64), 256), dtype=torch.uint8, device="cuda")
    outs = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                   # warm-up outside the capture (allocates the outputs)
        ops.forward_rays(net_c, net_f, static_rays, 64, 64, False, workspace=ws, outs=outs)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.forward_rays(net_c, net_f, static_rays, 64, 64, False, workspace=ws, outs=outs)
    other = rays.flip(0).contiguous()
    static_rays.copy_(other)
    graph.replay()
    torch.cuda.synchronize()
    got = {k: v.clone() for k, v in outs.items()}
    want = ops.forward_rays(net_c, net_f, other, 64, 64, False)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert net_c.status() == 0 and net_f.status() == 0
    static_rays[5, 3] = float("nan")
    graph.replay()
    assert net_c.status(clear=True) & INPUT
