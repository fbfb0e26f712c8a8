#!/bin/bash
# round 3, GPU session 11: two staging register sets in the weight-gradient slice (two point groups in flight) vs one
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s11
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -3 | tee $O/summary.txt
for r in 1 2 3; do
  for v in "" wg1; do
    lib=$R/nerf_sr_amd/libnsr${v:+_$v}.so
    NSR_LIB_PATH=$lib timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r lib \"$v\"  ms_per_step %.3f  rays/s %.0f  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))" | tee -a $O/summary.txt
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/train_traced.log 2>&1)
head -6 $(find $O/trace_train -name "*kernel_stats.csv" | head -1) | cut -c1-200 | tee -a $O/summary.txt
