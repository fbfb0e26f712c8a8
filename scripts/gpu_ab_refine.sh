#!/bin/bash
# refinement pass: product vs libnsr_base.so on one box; outputs compared (bit for bit when the K order is unchanged)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/refab; mkdir -p $O; cd $R
for r in 1 2; do
NSR_LIB_PATH=$R/ab/libnsr_${BASE:-base}.so timeout 300 python scripts/refine_out.py $O/base.pt 2>&1 | tail -1
timeout 300 python scripts/refine_out.py $O/new.pt 2>&1 | tail -1
done
python - <<PY
import torch
a, b = torch.load("$O/base.pt"), torch.load("$O/new.pt")
print("max |new - base| =", float((a - b).abs().max()), "mean", float((a - b).abs().mean()), "equal:", bool(torch.equal(a, b)))
PY
