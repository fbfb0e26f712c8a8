#!/bin/bash
# round 3, GPU session 10: training tests on the per-path workspace, then the PMC constants of the final kernel sources
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s10
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -3 | tee $O/summary.txt
NSR_TRAIN_PATH=gemm timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -x -k "not chain" 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-250 | tee -a $O/summary.txt
timeout 600 python scripts/pmc_collect.py $O/r3_pmc.json f16x3 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 400 bash scripts/pmc_train_traffic.sh 2>&1 | tail -2 | cut -c1-300 | tee -a $O/summary.txt
cp gpurun_out/train_traffic/FETCH_SIZE.json $O/train_FETCH_SIZE.json; cp gpurun_out/train_traffic/WRITE_SIZE.json $O/train_WRITE_SIZE.json
