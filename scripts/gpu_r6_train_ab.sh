#!/bin/bash
# round 6, training step: interleaved A/B on ONE box over (library, train precision) pairs; then per-kernel times.
# usage (GPU box): bash scripts/gpu_r6_train_ab.sh "new:f16x3_bwd2 w4:f16x3_bwd2 ..." [pytest]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_train_ab; mkdir -p $O; cd $R
libof() { [ "$1" = "new" ] && echo $R/nerf_sr_amd/libnsr.so || echo $R/ab/libnsr_$1.so; }
for r in 1 2 3; do
  for pair in $1; do
    v=${pair%%:*}; p=${pair##*:}
    NSR_LIB_PATH=$(libof $v) timeout 300 python bench.py --mode train --train-precision $p --steps 40 --warmup 8 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r $pair  ms_per_step %.3f  losses %s' % (d['ms_per_step'], d['losses']))"
  done
done | tee $O/ab.txt
for pair in $1; do
  v=${pair%%:*}; p=${pair##*:}
  rm -rf /tmp/ta_$v_$p
  (cd /tmp && NSR_LIB_PATH=$(libof $v) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ta_${v}_$p -o run -- python $R/bench.py --mode train --train-precision $p --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
  f=$(find /tmp/ta_${v}_$p -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_${v}_$p.csv
  echo "== $pair"; python3 -c "import csv,sys; [print(r[\"Name\"][:60], r[\"Calls\"], round(float(r[\"AverageNs\"])/1e3,1)) for r in list(csv.DictReader(open(sys.argv[1])))[:4]]" $f
done | tee $O/kernels.txt
if [ "$2" = "pytest" ]; then timeout 1500 python -m pytest tests/test_gpu_train.py -q --maxfail=30 2>&1 | tail -30 | tee $O/pytest.txt; fi
