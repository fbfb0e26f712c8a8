#!/bin/bash
# refinement pass: interleaved same-box A/B of library builds (names = libnsr_<name>.so, "new" = product) + the refine tests
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5_refine; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_refine.py -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  for v in "$@"; do
    lib=$R/nerf_sr_amd/libnsr_$v.so; [ "$v" = "new" ] && lib=$R/nerf_sr_amd/libnsr.so
    NSR_LIB_PATH=$lib timeout 300 python scripts/refine_out.py $O/$v.pt 5 2>&1 | tail -1 | sed "s|.*libnsr|libnsr|" | tee -a $O/ab.txt
  done
done
python - "$@" <<PY
import sys, torch
names = sys.argv[1:]
ref = torch.load("$O/%s.pt" % names[0])
for n in names[1:]:
    b = torch.load("$O/%s.pt" % n)
    print("max |%s - %s| = %.3e  equal: %s" % (n, names[0], float((ref - b).abs().max()), bool(torch.equal(ref, b))))
PY
