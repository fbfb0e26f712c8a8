#!/bin/bash
# development aid: training tests + the training bench (3 runs) + kernel stats of the step
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/train_check
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -2 | tee $O/summary.txt
for r in 1 2 3; do
  for v in 1 0; do
  NSR_WGRAD_JOBS=$v timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r jobs=$v  ms_per_step %.3f' % d['ms_per_step'])" | tee -a $O/summary.txt
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/traced.log 2>&1)
head -9 $(find $O/trace -name "*kernel_stats.csv" | head -1) | cut -c1-160 | tee -a $O/summary.txt
rm -rf $O/trace
