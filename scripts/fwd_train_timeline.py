"""Development aid: phase timeline (scripts/timeline.py's stamps) of the TRAINING step's forward kernel (the TRAIN instantiation
of mlp_f16x3_kernel, fine network: the last forward launch of a step) from an NSR_ABL_TIMELINE build.
NSR_LIB_PATH=.../libnsr_tl.so python scripts/fwd_train_timeline.py [out.json]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import _lib, train as tr  # noqa: E402
from nerf_sr_amd.weights import make_state_dict  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fwd_train_timeline.json"
lib = _lib.load()
R, S2 = 2048, 4
gen = torch.Generator().manual_seed(0)
rays = torch.zeros(R, 8)
rays[:, 0:3] = torch.randn(R, 3, generator=gen) * 0.1
rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen) + torch.tensor([0.0, 0.0, -3.0]), dim=-1)
rays[:, 7] = 1.0
t = tr.Trainer(make_state_dict(99), make_state_dict(100), N_coarse=64, N_importance=128, noise_std=1.0)
t.set_input(rays.cuda(), torch.rand(R // S2, 3, generator=gen).cuda())
for _ in range(5):
    t.optimize_parameters()
torch.cuda.synchronize()
G, S = 65536, 10
buf = np.zeros(G * 4 * S, dtype=np.uint64)
lib.nsr_dbg_timeline.restype = ctypes.c_int
assert lib.nsr_dbg_timeline(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
n_wg = R * 192 // 128
st = buf.reshape(G, 4, S)[:n_wg].astype(np.int64)[:, :, :8]
names = ["prologue", "L1", "trunk", "sigma", "dir", "tail(rgb+drain)", "composite"]
d = np.diff(st, axis=2)
steady = slice(1024, n_wg)            # skip the first resident generation
out = {"n_workgroups": int(n_wg), "phases_cycles": {nm: {"median": float(np.median(d[steady, :, i])), "mean": float(d[steady, :, i].mean())}
                                                       for i, nm in enumerate(names)},
       "workgroup_cycles": {"median": float(np.median(st[steady, :, 7] - st[steady, :, 0]))}}
kb = np.zeros(G * 4 * 8, dtype=np.uint64)
if lib.nsr_dbg_ksteps(kb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(kb.nbytes)) == 0:
    k = kb.reshape(G, 4, 8)[:n_wg].astype(np.int64)[steady]
    seg = np.diff(k[:, :, :6], axis=2)
    out["ksteps_of_one_trunk_chunk_cycles"] = {n: float(np.median(seg[:, :, i])) for i, n in enumerate(["k0-3", "k4-7", "k8-10", "k11-13", "k14-15"])}
json.dump(out, open(OUT, "w"), indent=1)
print(json.dumps(out))

# per-chunk stamps (scripts/patches/fwd_train_chunks.patch): start of every trunk chunk (L2 .. L8, xyz_encoding_final: 64) + the end
if hasattr(lib, "nsr_dbg_chunks"):
    cb = np.zeros(4096 * 4 * 80, dtype=np.uint64)
    lib.nsr_dbg_chunks.restype = ctypes.c_int
    if lib.nsr_dbg_chunks(cb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cb.nbytes)) == 0:
        c = cb.reshape(4096, 4, 80)[:n_wg].astype(np.int64)[steady]
        dur = c[:, :, 1:65] - c[:, :, 0:64]
        # the last stamp (index 64) is the kernel's end: chunk 63's figure includes sigma, dir_encoding and the tail
        tab = [[round(float(dur[:, :, 8 * l + nb].mean())) for nb in range(8)] for l in range(8)]
        print("chunk cycles by layer (L2 .. L8, final) and block:")
        for row in tab:
            print(row)
        out["chunk_cycles_mean_by_layer_and_block"] = tab
        json.dump(out, open(OUT, "w"), indent=1)
