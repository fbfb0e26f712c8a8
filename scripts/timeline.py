"""Development aid: phase timeline of the fine-pass f16x3 MLP launch from an NSR_ABL_TIMELINE build
(python -m nerf_sr_amd.build --variant tl -DNSR_ABL_TIMELINE; NSR_LIB_PATH=.../libnsr_tl.so python scripts/timeline.py [out.json]).
Per workgroup and wave: s_memtime at kernel entry / after the encoding prologue / after L1 / after the trunk / after the density
head / after dir_encoding / before the compositing epilogue / at exit, plus HW_ID and XCC_ID -> per-phase cycle statistics, wave
skew, and the gap between consecutive workgroups on one CU (dispatch + launch overhead a persistent kernel would not pay)."""
import ctypes, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import ops, cameras, _lib
from nerf_sr_amd.weights import make_state_dict

ns = int(os.environ.get("TL_SAMPLES", "128"))
net = ops.VanillaMLP(precision="f16x3").load_state_dict(make_state_dict(100))
rays = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True).reshape(-1, 8)
z = torch.sort(torch.rand(rays.shape[0], ns, device='cuda'), -1)[0].contiguous()
for i in range(3):
    ops.render_rays_composited(net, rays, z, False)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
G, S = 65536, 10
buf = np.zeros(G * 4 * S, dtype=np.uint64)
rc = lib.nsr_dbg_timeline(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
n_wg = min(G, (rays.shape[0] * ns + 127) // 128)
t = buf.reshape(G, 4, S)[:n_wg].astype(np.int64)
st, hw, xcc = t[:, :, :8], t[:, :, 8], t[:, :, 9]
names = ["prologue", "L1", "trunk", "sigma", "dir", "tail(rgb+drain)", "composite"]
d = np.diff(st, axis=2)                        # (wg, wave, 7)
out = {"n_workgroups": int(n_wg), "samples_per_ray": ns, "phases_cycles": {}}
# skip the first resident generation (cold caches) for the steady-state numbers
steady = slice(2048, n_wg)
for i, nm in enumerate(names):
    x = d[steady, :, i]
    out["phases_cycles"][nm] = {"median": float(np.median(x)), "mean": float(x.mean()), "p90": float(np.percentile(x, 90))}
tot = st[steady, :, 7] - st[steady, :, 0]
out["workgroup_cycles"] = {"median": float(np.median(tot)), "mean": float(tot.mean())}
out["wave_skew_at_exit_cycles"] = {"median": float(np.median(st[steady, :, 7].max(1) - st[steady, :, 7].min(1)))}
out["wave_skew_at_entry_cycles"] = {"median": float(np.median(st[steady, :, 0].max(1) - st[steady, :, 0].min(1)))}
# gap between consecutive workgroups on one CU: key = (xcc, se, sh, cu) from HW_ID (gfx9: cu_id [11:8], sh_id [12], se_id [15:13])
key = (xcc[:, 0] << 16) | (hw[:, 0] & 0xFF00)
gaps, spans = [], []
for k in np.unique(key):
    idx = np.nonzero(key == k)[0]
    o = idx[np.argsort(st[idx, :, 0].min(1))]
    start = st[o, :, 0].min(1)
    end = st[o, :, 7].max(1)
    g = start[1:] - end[:-1]
    gaps.append(g[2:])
    spans.append((start[1:] - start[:-1])[2:])
gaps = np.concatenate(gaps); spans = np.concatenate(spans)
out["n_cu_keys"] = int(len(np.unique(key)))
out["gap_between_workgroups_on_a_cu_cycles"] = {"median": float(np.median(gaps)), "mean": float(gaps.mean()),
                                                 "p10": float(np.percentile(gaps, 10)), "p90": float(np.percentile(gaps, 90))}
out["start_to_start_on_a_cu_cycles"] = {"median": float(np.median(spans)), "mean": float(spans.mean())}
kb = np.zeros(G * 4 * 8, dtype=np.uint64)
if hasattr(lib, "nsr_dbg_ksteps") and lib.nsr_dbg_ksteps(kb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(kb.nbytes)) == 0:
    k = kb.reshape(G, 4, 8)[:n_wg].astype(np.int64)[steady]
    seg = np.diff(k[:, :, :6], axis=2)     # k-steps [0,4) [4,8) [8,11) [11,14) [14,16) of chunk (L8's layer 7, block 3)
    out["ksteps_of_one_trunk_chunk_cycles"] = {n: {"median": float(np.median(seg[:, :, i])), "mean": float(seg[:, :, i].mean())}
                                                for i, n in enumerate(["k0-3", "k4-7", "k8-10", "k11-13", "k14-15"])}
    out["ksteps_of_one_trunk_chunk_cycles"]["chunk"] = {"median": float(np.median(k[:, :, 5] - k[:, :, 0]))}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
