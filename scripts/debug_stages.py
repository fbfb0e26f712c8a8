"""Development aid: per-stage max error of the HIP path vs oracle/golden, with locations."""
import os, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import ops, cameras
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc
torch.set_printoptions(precision=8, linewidth=200)
def cu(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def rep(name, got, want):
    got = got.detach().cpu().double(); want = torch.as_tensor(np.asarray(want)).double() if not isinstance(want, torch.Tensor) else want.double()
    d = (got - want).abs()
    i = int(d.argmax()); idx = np.unravel_index(i, d.shape) if d.ndim else ()
    print(f"{name:40s} max {d.max().item():.3e} at {idx} got {got.flatten()[i].item():.8g} want {want.flatten()[i].item():.8g}  mean {d.mean().item():.2e}")
    return d
G = 'tests/golden'
e = np.load(f'{G}/edge_cases.npz')
# composite at several N
gen = torch.Generator().manual_seed(5)
for N in (2, 7, 64, 100, 128, 192, 300):
    R = 37
    z = torch.sort(torch.rand(R, N, generator=gen) * 4 + 2, -1)[0]
    rgb = torch.rand(R, N, 3, generator=gen); sig = torch.randn(R, N, generator=gen) * 20
    want = oc.composite(rgb, sig, z, True)
    got = ops.VolumetricRenderer()(rgb.cuda(), sig.cuda(), z.cuda(), True)
    for nm, a, b in zip(("comp", "depth", "opac", "w"), got, want):
        rep(f"composite N={N} {nm}", a, b)
# resample edge
z, w = cu(e["z"]), cu(e["weights"]); R = z.shape[0]
o = torch.zeros(R, 3).cuda(); d = torch.tensor([[0., 0., -1.]]).repeat(R, 1).cuda()
zf, _ = ops.resample_along_rays(o, d, z, w, 64, False)
dd = rep("resample edge", zf, e["z_fine"])
print(" per-ray max err:", dd.max(1)[0].numpy())
for tag in ("llff", "blender"):
    g = np.load(f'{G}/path_{tag}.npz')
    rays = cu(g["rays"])
    net_c = ops.VanillaMLP().load_state_dict(make_state_dict(99)); net_f = ops.VanillaMLP().load_state_dict(make_state_dict(100))
    zc = cu(g["z_coarse"])
    rgb, sig = ops.render_rays(net_c, rays, zc)
    rep(f"{tag} fused sigma", sig, g["coarse_point_sigma"]); rep(f"{tag} fused rgb16", rgb[:16], g["coarse_point_rgb"])
    x = cu(g["mlp_in_512"]); out = net_c(x)
    rep(f"{tag} unfused mlp sigma512", out[:, 3], g["mlp_out_coarse_512"][:, 3])
    rep(f"{tag} fused sigma first512", sig.reshape(-1)[:512], g["mlp_out_coarse_512"][:, 3])
    pe = ops.PositionalEncoding(3, 10)(ops.cast_rays(rays[:, :3], rays[:, 3:6], zc).reshape(-1, 3)[:512].contiguous())
    rep(f"{tag} posenc vs mlp_in", pe, g["mlp_in_512"][:, :63])
    zf, _ = ops.resample_along_rays(rays[:, :3], rays[:, 3:6], zc, cu(g["coarse_weights"]), 64, False)
    dd = rep(f"{tag} resample path", zf, g["z_fine"])
    out = ops.forward_rays(net_c, net_f, rays, 64, 64, bool(g["white_bkgd"]))
    for k in ops.OUT_KEYS: rep(f"{tag} fwd {k}", out[k], g[k])
