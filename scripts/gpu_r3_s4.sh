#!/bin/bash
# round 3, GPU session 4: the four-wave 128 x 256 GEMM tile (64 x 128 per wave) -- correctness and A/B against the others
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s4
mkdir -p $O
cd $R
NSR_GEMM_TILE=q timeout 600 python -m pytest tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -3 | tee $O/summary.txt
for r in 1 2; do for tile in auto narrow q; do
  NSR_GEMM_TILE=$tile timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r tile=$tile: /" | tee -a $O/summary.txt
done; done
(cd /tmp && NSR_GEMM_TILE=q timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_q -o run -- python $R/scripts/prof_refine.py 256 3 > $O/trace_q.log 2>&1)
python scripts/refine_layers.py $(ls $O/trace_q/*/*kernel_trace.csv $O/trace_q/*kernel_trace.csv 2>/dev/null | head -1) 2>&1 | tail -40 | tee -a $O/summary.txt
