"""Development aid: interleaved same-box A/B of libnsr builds through the three C-ABI entry points every round's library shares
(nsr_packed_weights_bytes, nsr_pack_weights, nsr_render_rays): the f16x3 network launch of the fine pass (190,512 rays x 128
samples, no compositing -- round 1 had no fused compositor), 10 launches per sample, libraries alternated.
usage: ab_libs.py <rounds> <name>=<path.so> [<name>=<path.so> ...] [--json out.json]"""
import ctypes, json, os, sys, time
from ctypes import c_void_p, c_int, c_int64, c_size_t, POINTER
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import cameras
from nerf_sr_amd.weights import make_state_dict
from nerf_sr_amd.ops import STATE_DICT_SPEC

out_json = None
argv = sys.argv[1:]
if "--json" in argv:
    i = argv.index("--json")
    out_json = argv[i + 1]
    del argv[i:i + 2]
rounds = int(argv[0])
libs = [a.split("=", 1) for a in argv[1:]]
F16X3 = 2
sd = make_state_dict(100)
dev = [torch.from_numpy(sd[k]).float().cuda().contiguous() for k in STATE_DICT_SPEC]
ptrs = (c_void_p * 24)(*[c_void_p(t.data_ptr()) for t in dev])
H, W, s = 378, 504, 2
R, N = H * W, 128
g = torch.Generator(device="cuda").manual_seed(1)
rays = torch.empty(R, 8, device="cuda")
rays[:, 0:3] = torch.rand(R, 3, device="cuda", generator=g) - 0.5
d = torch.randn(R, 3, device="cuda", generator=g)
rays[:, 3:6] = d / d.norm(dim=1, keepdim=True)
rays[:, 6], rays[:, 7] = 0.0, 1.0
z = torch.sort(torch.rand(R, N, device="cuda", generator=g), -1)[0].contiguous()
out = torch.empty(R * N, 4, device="cuda")
handles = {}
for name, path in libs:
    L = ctypes.CDLL(path)
    L.nsr_packed_weights_bytes.restype = c_size_t
    L.nsr_packed_weights_bytes.argtypes = [c_int]
    L.nsr_pack_weights.argtypes = [POINTER(c_void_p), c_void_p, c_int, c_void_p]
    L.nsr_render_rays.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]
    blob = torch.zeros(L.nsr_packed_weights_bytes(F16X3) + 64, dtype=torch.uint8, device="cuda")
    assert L.nsr_pack_weights(ptrs, blob.data_ptr(), F16X3, None) == 0
    handles[name] = (L, blob)
torch.cuda.synchronize()

def run(name, n):
    L, blob = handles[name]
    for _ in range(n):
        rc = L.nsr_render_rays(blob.data_ptr(), F16X3, rays.data_ptr(), 8, z.data_ptr(), R, N, out.data_ptr(), None)
        assert rc == 0, rc

res = {name: [] for name, _ in libs}
ref = None
for name, _ in libs:          # warm-up + agreement of the outputs (same arithmetic family: report the difference)
    run(name, 2); torch.cuda.synchronize()
    o = out.clone()
    if ref is None: ref = o
    else: print(f"max |out[{name}] - out[{libs[0][0]}]| = {float((o - ref).abs().max()):.3e}")
for r in range(rounds):
    for name, _ in libs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(name, 10)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        res[name].append(dt * 1e3)
        print(f"round {r} {name}: {dt * 1e3:.3f} ms")
summary = {name: {"ms_median": sorted(v)[len(v) // 2], "ms_min": min(v), "ms_all": [round(x, 3) for x in v]} for name, v in res.items()}
print(json.dumps(summary))
if out_json: json.dump(summary, open(out_json, "w"), indent=1)
