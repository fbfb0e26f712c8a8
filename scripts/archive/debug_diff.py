"""Development aid: where do two library builds disagree?  usage: debug_diff.py ref.so new.so [R] [N]"""
import ctypes, os, sys
from ctypes import c_void_p, c_int, c_int64, c_size_t, POINTER
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd.weights import make_state_dict
from nerf_sr_amd.ops import STATE_DICT_SPEC
R = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
N = int(sys.argv[4]) if len(sys.argv) > 4 else 128
sd = make_state_dict(100)
dev = [torch.from_numpy(sd[k]).float().cuda().contiguous() for k in STATE_DICT_SPEC]
ptrs = (c_void_p * 24)(*[c_void_p(t.data_ptr()) for t in dev])
g = torch.Generator(device="cuda").manual_seed(1)
rays = torch.empty(R, 8, device="cuda")
rays[:, 0:3] = torch.rand(R, 3, device="cuda", generator=g) - 0.5
d = torch.randn(R, 3, device="cuda", generator=g)
rays[:, 3:6] = d / d.norm(dim=1, keepdim=True)
rays[:, 6], rays[:, 7] = 0.0, 1.0
z = torch.sort(torch.rand(R, N, device="cuda", generator=g), -1)[0].contiguous()
outs = []
for path in sys.argv[1:3]:
    L = ctypes.CDLL(path)
    L.nsr_packed_weights_bytes.restype = c_size_t
    L.nsr_packed_weights_bytes.argtypes = [c_int]
    L.nsr_pack_weights.argtypes = [POINTER(c_void_p), c_void_p, c_int, c_void_p]
    L.nsr_render_rays.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]
    blob = torch.zeros(L.nsr_packed_weights_bytes(2) + 64, dtype=torch.uint8, device="cuda")
    assert L.nsr_pack_weights(ptrs, blob.data_ptr(), 2, None) == 0
    out = torch.full((R * N, 4), float("nan"), device="cuda")
    assert L.nsr_render_rays(blob.data_ptr(), 2, rays.data_ptr(), 8, z.data_ptr(), R, N, out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    outs.append(out.clone())
a, b = outs
diff = (a - b).abs().nan_to_num(nan=1e9).max(1)[0]
P = R * N
n_tiles = (P + 127) // 128
pad = torch.zeros(n_tiles * 128, device="cuda"); pad[:P] = diff
t = pad.view(n_tiles, 4, 32)
bad_tiles = (t.amax((1, 2)) > 0).nonzero().flatten()
print(f"P={P} tiles={n_tiles} bad tiles={len(bad_tiles)} first bad: {bad_tiles[:20].tolist()}")
print("bad points per wave:", (t > 0).sum((0, 2)).tolist(), " per lane m (first 32):", (t > 0).sum((0, 1)).tolist())
print("nan in new:", int(torch.isnan(b).sum()), " max diff:", float(diff[diff < 1e8].max()) if (diff < 1e8).any() else None)
bt = bad_tiles[:3].tolist()
for ti in bt:
    print("tile", ti, "ref", a[ti * 128: ti * 128 + 2].tolist(), "new", b[ti * 128: ti * 128 + 2].tolist())
