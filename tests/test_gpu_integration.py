"""The boundary as a reference maintainer would use it (VERDICT r3 "next" #6):

* the ctypes stub printed in INTEGRATION.md section 1 is EXTRACTED from the document and EXECUTED verbatim (only the library
  path is substituted) against the reference-generated fixture ``path_llff.npz`` -- a typo in an argument order there
  cannot ship;
* ``include/nsr.h`` promises "re-entrant from any number of host threads" (nn.DataParallel runs one thread per GPU,
  models/networks.py:54-69): two host threads on two streams call ``nsr_forward_rays`` concurrently, with separate packed
  networks and with ONE shared pair, and must reproduce the serial results bit for bit, status words included.
"""
import os
import re
import threading
from collections import OrderedDict

import numpy as np
import pytest
import torch

from nerf_sr_amd import _lib
from nerf_sr_amd.weights import make_state_dict
from nerf_sr_amd.ops import STATE_DICT_SPEC

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = text[text.index("## 1."):text.index("## 2.")]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section 1 has no python code block"
    src = m.group(1)
    assert '"/path/to/libnsr.so"' in src
    return src.replace('"/path/to/libnsr.so"', repr(_lib.LIB_PATH))


class _Net:
    """What the stub's pack() needs from models/networks.py::VanillaMLP: state_dict() in the module's key order."""

    def __init__(self, sd):
        self._sd = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(sd[k])).cuda()) for k in STATE_DICT_SPEC)

    def state_dict(self):
        return self._sd


def test_integration_md_stub_runs_verbatim(golden_dir):
    assert torch.cuda.is_available()
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md#1", "exec"), ns)
    g = np.load(os.path.join(golden_dir, "path_llff.npz"))
    blob_c = ns["pack"](_Net(make_state_dict(int(g["seed_coarse"]))))
    blob_f = ns["pack"](_Net(make_state_dict(int(g["seed_fine"]))))
    rays = torch.from_numpy(g["rays"]).cuda()
    out = ns["forward_rays"](blob_c, blob_f, rays, 64, 64, bool(g["white_bkgd"]), False)
    torch.cuda.synchronize()
    assert list(out) == ["coarse_comp_rgbs", "coarse_depth", "coarse_opacity", "coarse_weights",
                         "fine_comp_rgbs", "fine_depth", "fine_opacity", "fine_weights"]
    for k in ("coarse_comp_rgbs", "fine_comp_rgbs"):
        err = np.abs(out[k].cpu().numpy() - g[k]).max()
        assert err <= 1e-4, (k, err)                      # north_star: 1e-4 RGB
    for k in ("coarse_opacity", "fine_opacity", "coarse_depth", "fine_depth"):
        assert np.abs(out[k].cpu().numpy() - g[k]).max() <= 1e-4, k
    assert np.abs(out["coarse_weights"].cpu().numpy() - g["coarse_weights"]).max() <= 1e-4
    # the gamma option of the stub reaches the library too: colours become rgb ** (1 / 2.2) per sample, so the composite moves
    blob_g = ns["pack"](_Net(make_state_dict(int(g["seed_coarse"]))), gamma_correct=True)
    out_g = ns["forward_rays"](blob_g, blob_f, rays, 64, 64, bool(g["white_bkgd"]), False)
    torch.cuda.synchronize()
    assert float((out_g["coarse_comp_rgbs"] - out["coarse_comp_rgbs"]).abs().max()) > 1e-3
    # the two option words of round 5 through the stub, against the reference's own forward with the option on (options.npz)
    o = np.load(os.path.join(golden_dir, "options.npz"))
    n = int(o["n_rays"])
    out_s = ns["forward_rays"](blob_c, blob_f, rays[:n].contiguous(), 64, 64, bool(g["white_bkgd"]), False, softplus=True)
    blob_nc, blob_nf = (ns["pack"](_Net(make_state_dict(int(g[k]))), color_none=True) for k in ("seed_coarse", "seed_fine"))
    out_n = ns["forward_rays"](blob_nc, blob_nf, rays[:n].contiguous(), 64, 64, bool(g["white_bkgd"]), False)
    torch.cuda.synchronize()
    assert np.abs(out_s["fine_comp_rgbs"].cpu().numpy() - o["softplus_llff_fine_comp_rgbs"]).max() <= 1e-4
    assert np.abs(out_n["fine_comp_rgbs"].cpu().numpy() - o["color_none_llff_fine_comp_rgbs"]).max() <= 1e-4 * max(1.0, np.abs(o["color_none_llff_fine_comp_rgbs"]).max())
    # ... and check() of the stub raises on a poisoned input (the reference drops into pdb there)
    bad = rays.clone()
    bad[3, 0] = float("nan")
    with pytest.raises(FloatingPointError):
        ns["forward_rays"](blob_c, blob_f, bad, 64, 64, False, False)


def _forward(lib, blob_c, blob_f, rays, outs, ws, stream):
    ptrs = (_lib.c_void_p * 8)(*[_lib.c_void_p(t.data_ptr()) for t in outs])
    return lib.nsr_forward_rays(blob_c.data_ptr(), blob_f.data_ptr(), _lib.NSR_F16X3, rays.data_ptr(), 8, rays.shape[0], 64, 64,
                                0, 0, ptrs, ws.data_ptr(), ws.numel(), _lib.c_void_p(stream.cuda_stream))


def _alloc(R, dev="cuda"):
    return [torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev), torch.empty(R, 64, device=dev),
            torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev), torch.empty(R, 128, device=dev)]


@pytest.mark.parametrize("shared", [False, True], ids=["separate-blobs", "shared-blobs"])
def test_forward_rays_from_two_host_threads(shared):
    """Two host threads, two streams, concurrent nsr_forward_rays (each with its own workspace and outputs: the library owns
    no memory).  Separate packed networks: nothing is shared.  Shared networks: the only shared device state is the sticky
    status word, which kernels touch with atomicOr and only when a flag is raised -- thread 1's poisoned rays must raise
    INPUT_RANGE | OUTPUT_NONFINITE there without disturbing thread 0's results."""
    from nerf_sr_amd import ops
    lib = _lib.load()
    sd_c, sd_f = make_state_dict(21), make_state_dict(22)
    nets = [(ops.VanillaMLP(precision="f16x3").load_state_dict(sd_c), ops.VanillaMLP(precision="f16x3").load_state_dict(sd_f))]
    nets.append(nets[0] if shared else (ops.VanillaMLP(precision="f16x3").load_state_dict(sd_c),
                                        ops.VanillaMLP(precision="f16x3").load_state_dict(sd_f)))
    R = 6000                                   # 3,000 / 6,000 tiles per pass: long enough for the two streams to overlap
    gen = torch.Generator().manual_seed(9)
    rays = []
    for t in range(2):
        d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=1)
        rays.append(torch.cat([torch.rand(R, 3, generator=gen) - 0.5, d, torch.zeros(R, 1), torch.ones(R, 1)], 1).cuda())
    rays[1][17, 4] = float("inf")              # thread 1 carries one poisoned ray
    ws_bytes = lib.nsr_forward_rays_workspace_bytes_for(_lib.NSR_F16X3, R, 64, 64)
    # ---- serial reference on the default stream
    want = []
    for t in range(2):
        outs, ws = _alloc(R), torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        assert _forward(lib, nets[t][0].packed, nets[t][1].packed, rays[t], outs, ws, torch.cuda.current_stream()) == 0
        torch.cuda.synchronize()
        want.append([o.clone() for o in outs])
    def read_flags():
        # shared networks have ONE status word each: read (and clear) it once and report it for both threads
        first = (nets[0][0].status(clear=True), nets[0][1].status(clear=True))
        return [first, first if shared else (nets[1][0].status(clear=True), nets[1][1].status(clear=True))]

    flags_serial = read_flags()
    assert flags_serial[1][0] & 2 and (shared or flags_serial[0] == (0, 0))
    # ---- two threads, two streams, several rounds
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[_alloc(R) for _ in range(4)] for _ in range(2)]
    wss = [torch.empty(ws_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
    rcs = [[], []]
    barrier = threading.Barrier(2)

    def work(t):
        torch.cuda.set_device(0)
        barrier.wait()
        for i in range(4):
            rcs[t].append(_forward(lib, nets[t][0].packed, nets[t][1].packed, rays[t], got[t][i], wss[t], streams[t]))
        streams[t].synchronize()

    torch.cuda.synchronize()
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    assert rcs == [[0] * 4, [0] * 4]
    for t in range(2):
        for i in range(4):
            for a, b in zip(got[t][i], want[t]):
                assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (t, i)
    flags = read_flags()
    if shared:
        assert flags[0] == flags_serial[1]      # one status word: the poisoned thread's flags, exactly as in the serial run
    else:
        assert flags[0] == (0, 0) and flags[1] == flags_serial[1]
