#!/bin/bash
# round 3, GPU session 8: max over the reference patches fused into the producing GEMM's epilogue (grouped rows)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s8
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_refine.py tests/test_gpu_frames.py::test_config5_composed_small_vs_oracles tests/test_gpu_frames.py::test_config5_full_size_end_to_end -q -m gpu -x 2>&1 | tail -3 | tee $O/summary.txt
NSR_REFINE_SEPARATE_MAX=1 timeout 600 python -m pytest tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -2 | tee -a $O/summary.txt
# bit-identity of the two routes on the reference's shape
timeout 300 python - <<'PY' 2>&1 | tail -3 | tee -a $O/summary.txt
import os, subprocess, sys, torch
sys.path.insert(0, os.getcwd())
code = '''
import torch, sys, os
sys.path.insert(0, os.getcwd())
from nerf_sr_amd import refine
net = refine.MaxPoolingModel().load_state_dict(refine.make_refine_state_dict(7)).eval()
g = torch.Generator().manual_seed(1)
x = (torch.rand(5, 3, 64, 64, generator=g) * 2 - 1).cuda(); c = (torch.rand(5, 8, 3, 64, 64, generator=g) * 2 - 1).cuda()
torch.save(net(x, c).cpu(), sys.argv[1])
'''
subprocess.run([sys.executable, "-c", code, "/tmp/fused.pt"], check=True)
subprocess.run([sys.executable, "-c", code, "/tmp/sep.pt"], check=True, env=dict(os.environ, NSR_REFINE_SEPARATE_MAX="1"))
a, b = torch.load("/tmp/fused.pt"), torch.load("/tmp/sep.pt")
print("fused max vs separate max kernels: bit-identical =", bool(torch.equal(a, b)), " max |diff| =", float((a - b).abs().max()))
PY
for r in 1 2 3; do
  timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r fused max: /" | tee -a $O/summary.txt
  NSR_REFINE_SEPARATE_MAX=1 timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r separate max kernels: /" | tee -a $O/summary.txt
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python $R/scripts/prof_refine.py 256 3 > $O/trace.log 2>&1)
python scripts/refine_layers.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 2>&1 | tail -34 | tee -a $O/summary.txt
