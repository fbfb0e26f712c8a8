"""Development probe (GPU box): the training step replayed from ONE hipGraph (torch.cuda.CUDAGraph over Trainer.optimize_parameters:
~45 launches per step) against the eager enqueue; same batch, same weights at the start.  usage: train_graph_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_sr_amd import ops, cameras, train as tr
from nerf_sr_amd.weights import make_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
R = 2048
frame = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True)
torch.manual_seed(1234)
sel = torch.randperm(frame.shape[0], device="cuda")[: R // 4]
rays = frame[sel].reshape(-1, 8).contiguous()
target = torch.rand(R // 4, 3, device="cuda")


def make():
    t = tr.Trainer(make_state_dict(99), make_state_dict(100), randomized=True, noise_std=1.0, downscale=2, ray_chunk=R)
    t.set_input(rays, target)
    return t


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


t = make()
for _ in range(5): t.optimize_parameters()
eager = timed(t.optimize_parameters, steps)
# graph: warm up on a side stream, capture one iteration, replay
t2 = make()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): t2.optimize_parameters()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t2.optimize_parameters()
for _ in range(5): g.replay()
graph = timed(g.replay, steps)
print(f"eager {eager:.3f} ms per step, graph replay {graph:.3f} ms per step; losses eager {t.losses.tolist()} graph {t2.losses.tolist()} finite {bool(torch.isfinite(t2.losses).all())}")
