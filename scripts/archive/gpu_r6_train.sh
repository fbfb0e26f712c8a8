#!/bin/bash
# round 6, training step: the backward chain on 3 / 2 / 1 MFMA terms -- gradients, trajectory, step time (interleaved on ONE box),
# per-kernel times.  usage (GPU box): bash scripts/gpu_r6_train.sh [pytest]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_train; mkdir -p $O; cd $R
timeout 900 python scripts/bwd_terms_check.py > $O/bwd_terms_gpu.json 2> $O/bwd_terms_gpu.err; echo "terms check rc=$?"; tail -3 $O/bwd_terms_gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_train/bwd_terms_gpu.json"))
for c, v in d["fixtures"].items():
    for p, r in v.items(): print(c, p, "worst %.2e (%s) head %.2e net %.2e" % (r["worst_tensor"], r["worst_tensor_name"], r["worst_head_tensor"], r["whole_network"]))
for p, r in d["bench_scale"].items(): print("bench scale", p, r)
for p, r in d["trajectory"].items(): print("traj", p, r)
PY
for r in 1 2 3; do
  for v in f16x3_bwd3 f16x3_bwd2 f16x3_bwd1; do
    timeout 300 python bench.py --mode train --train-precision $v --steps 40 --warmup 8 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r $v  ms_per_step %.3f' % d['ms_per_step'])"
  done
done | tee $O/ab.txt
for v in f16x3_bwd3 f16x3_bwd2 f16x3_bwd1; do
  rm -rf /tmp/ta_$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ta_$v -o run -- python $R/bench.py --mode train --train-precision $v --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
  f=$(find /tmp/ta_$v -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$v.csv
  echo "== $v"; python3 -c "import csv,sys; [print(r[\"Name\"][:60], r[\"Calls\"], round(float(r[\"AverageNs\"])/1e3,1)) for r in list(csv.DictReader(open(sys.argv[1])))[:6]]" $f
done | tee $O/kernels.txt
if [ "$1" = "pytest" ]; then timeout 1500 python -m pytest tests/test_gpu_train.py -q --maxfail=30 2>&1 | tail -40 | tee $O/pytest.txt; fi
