#!/bin/bash
# round 3, GPU session 6: the big-tile GEMM (one wave per SIMD, 128 x 128 per wave): correctness, A/B, per-layer trace
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s6
mkdir -p $O
cd $R
NSR_GEMM_TILE=big timeout 600 python -m pytest tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -3 | tee $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_refine.py -q -m gpu -x 2>&1 | tail -2 | tee -a $O/summary.txt
for r in 1 2 3; do for tile in auto big quad; do
  NSR_GEMM_TILE=$tile timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r tile=$tile: /" | tee -a $O/summary.txt
done; done
for tile in big; do
  (cd /tmp && NSR_GEMM_TILE=$tile timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tile -o run -- python $R/scripts/prof_refine.py 256 3 > $O/trace_$tile.log 2>&1)
  echo "== per-layer, tile=$tile" | tee -a $O/summary.txt
  python scripts/refine_layers.py $(find $O/trace_$tile -name "*kernel_trace.csv" | head -1) 2>&1 | tail -36 | tee -a $O/summary.txt
done
