"""Development aid: refinement-network forward time (64 x 64 patches, 8 reference patches each)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import refine
from nerf_sr_amd.refine import LAYERS
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
net = refine.MaxPoolingModel(precision=prec).load_state_dict(refine.make_refine_state_dict(7))
macs = refine.refine_macs
for B in (1, 8, 32):
    x = torch.rand(B, 3, 64, 64, device="cuda") * 2 - 1
    c = torch.rand(B, 8, 3, 64, 64, device="cuda") * 2 - 1
    for _ in range(2): net(x, c)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 5; e0.record()
    for _ in range(n): net(x, c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2 * macs(64, 64, 8) * B
    print(f"{prec} B={B}: {ms:.2f} ms per batch, {ms / B:.3f} ms per patch set, {fl / ms / 1e9:.1f} TFLOP/s (true conv MACs)")
