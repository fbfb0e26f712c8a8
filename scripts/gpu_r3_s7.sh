#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s7
mkdir -p $O
cd $R
NSR_GEMM_TILE=big timeout 600 python -m pytest tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -2 | tee $O/summary.txt
for r in 1 2; do for tile in auto big; do
  NSR_GEMM_TILE=$tile timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r tile=$tile: /" | tee -a $O/summary.txt
done; done
(cd /tmp && NSR_GEMM_TILE=big timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_big -o run -- python $R/scripts/prof_refine.py 256 3 > $O/trace_big.log 2>&1)
python scripts/refine_layers.py $(find $O/trace_big -name "*kernel_trace.csv" | head -1) 2>&1 | tail -36 | tee -a $O/summary.txt
