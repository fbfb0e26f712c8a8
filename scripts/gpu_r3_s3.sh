#!/bin/bash
# round 3, GPU session 3: the whole GPU suite as the driver runs it (timed), refinement-pass tile experiments, PMC constants
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s3
mkdir -p $O
cd $R
t0=$(date +%s)
NSR_PARITY_REPORT=$O/parity_trained_tests.json timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 > $O/gpu_suite.log 2>&1; echo "suite rc $? in $(( $(date +%s) - t0 )) s" | tee $O/summary.txt
tail -22 $O/gpu_suite.log | cut -c1-200 | tee -a $O/summary.txt
for tile in auto narrow wide; do for tk in 64 32; do
  NSR_GEMM_TILE=$tile NSR_GEMM_TK=$tk timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/tile=$tile tk=$tk: /" | tee -a $O/summary.txt
done; done
timeout 600 python scripts/pmc_collect.py $O/r3_pmc.json f16x3 2>&1 | tail -6 | cut -c1-400 | tee -a $O/summary.txt
