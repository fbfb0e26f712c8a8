"""Development aid: chunk timeline of the training step's backward chain kernel from an NSR_ABL_TIMELINE build that carries
scripts/patches/chain_bwd_timeline.patch (s_memtime at the start of every chunk, around the prologue and the tail, kept in LDS and
dumped at the end of the kernel; 100 MHz counter).   NSR_LIB_PATH=.../libnsr_tl.so python scripts/bwd_timeline.py [out.json]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import _lib, train as tr  # noqa: E402
from nerf_sr_amd.weights import make_state_dict  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bwd_timeline.json"
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
SLOTS, TILES = 96, 4096
buf = np.zeros(TILES * 4 * SLOTS, dtype=np.uint64)
lib.nsr_dbg_bwd_timeline.restype = ctypes.c_int
rc = lib.nsr_dbg_bwd_timeline(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
n_tiles = R * 192 // 128                                   # the last launch: the fine network's backward
tl = buf.reshape(TILES, 4, SLOTS)[:n_tiles].astype(np.int64)
TICK_NS = 10.0                                             # s_memtime counts at 100 MHz on gfx950
chunk = (tl[:, :, 1:73] - tl[:, :, 0:72]) * TICK_NS       # duration of chunk q = stamp q+1 - stamp q (72 = start of the tail)
rep = {"tiles": int(n_tiles), "tick_ns": TICK_NS,
       "prologue_ns": float(((tl[:, :, 81] - tl[:, :, 80]) * TICK_NS).mean()),
       "first_chunk_wait_ns": float(((tl[:, :, 82] - tl[:, :, 81]) * TICK_NS).mean()),
       "to_first_chunk_ns": float(((tl[:, :, 0] - tl[:, :, 82]) * TICK_NS).mean()),
       "tail_ns": float(((tl[:, :, 73] - tl[:, :, 72]) * TICK_NS).mean()), "drain_ns": float(((tl[:, :, 74] - tl[:, :, 73]) * TICK_NS).mean()),
       "kernel_ns_per_tile": float(((tl[:, :, 74] - tl[:, :, 80]) * TICK_NS).mean()),
       "chunk_ns_mean_by_layer_and_block": [[round(float(chunk[:, :, 8 * lam + nb].mean()), 1) for nb in range(8)] for lam in range(9)],
       "chunk_ns_p90_by_layer_and_block": [[round(float(np.percentile(chunk[:, :, 8 * lam + nb], 90)), 1) for nb in range(8)] for lam in range(9)],
       "chunks_total_ns": float(chunk.sum(-1).mean())}
json.dump(rep, open(OUT, "w"), indent=1)
print(json.dumps(rep)[:3000])
