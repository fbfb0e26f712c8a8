"""Development aid: time of the training GEMM's forward variant (nsr_linear) on the fine-pass layer shape."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_sr_amd import train as tr
P, K, N = 262144, 256, 256
x = torch.randn(P, K, device="cuda"); w = torch.randn(N, K, device="cuda") / 16; b = torch.randn(N, device="cuda")
for _ in range(3): tr.linear(x, w, b, act=1)
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
n = 20; e0.record()
for _ in range(n): tr.linear(x, w, b, act=1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{os.environ.get('NSR_LIB_PATH', 'default')}: {ms*1e3:.0f} us  {2*P*K*N/ms/1e9:.1f} TFLOP/s")
