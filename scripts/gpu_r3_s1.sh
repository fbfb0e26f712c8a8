#!/bin/bash
# round 3, GPU session 1: new tests, smoke, bench, amax A/B, trained-field parity report
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_status.py tests/test_gpu_frames.py -q -m gpu -s -x > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/summary.txt
grep -E "passed|failed|Error|error" $O/tests.log | tail -8 | cut -c1-300 | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400 | tee -a $O/summary.txt
tail -3 $O/bench.err | tee -a $O/summary.txt
timeout 300 bash scripts/ab_variants.sh 3 "" noamax 2>&1 | grep round | tee -a $O/summary.txt
timeout 1200 python scripts/parity_trained.py 16384 6000 > $O/parity_trained.log 2>&1; echo "parity_trained rc $?" | tee -a $O/summary.txt
grep -E "^\[train|^trained|^sharp|Error" $O/parity_trained.log | cut -c1-700 | tee -a $O/summary.txt
