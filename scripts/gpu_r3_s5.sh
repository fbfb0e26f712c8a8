#!/bin/bash
# round 3, GPU session 5: refinement pass with the new tile rule vs 4-wave 128 x 128 everywhere, per-layer traces of both
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_refine.py tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -3 | tee $O/summary.txt
for r in 1 2 3; do for tile in auto narrow wide; do
  NSR_GEMM_TILE=$tile timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r tile=$tile: /" | tee -a $O/summary.txt
done; done
for tile in auto narrow; do
  (cd /tmp && NSR_GEMM_TILE=$tile timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tile -o run -- python $R/scripts/prof_refine.py 256 3 > $O/trace_$tile.log 2>&1)
  echo "== per-layer, tile=$tile" | tee -a $O/summary.txt
  python scripts/refine_layers.py $(find $O/trace_$tile -name "*kernel_trace.csv" | head -1) 2>&1 | tail -36 | tee -a $O/summary.txt
done
