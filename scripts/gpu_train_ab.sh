#!/bin/bash
# training step: interleaved same-box A/B of library builds (names = libnsr_<name>.so, "new" = product) + per-kernel times
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/train_ab; mkdir -p $O; cd $R
for r in 1 2 3; do
  for v in "$@"; do
    lib=$R/ab/libnsr_$v.so; [ "$v" = "new" ] && lib=$R/nerf_sr_amd/libnsr.so
    NSR_LIB_PATH=$lib timeout 300 python bench.py --mode train --steps 25 --warmup 5 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r lib=$v  ms_per_step %.3f' % d['ms_per_step'])"
  done
done
for v in "$@"; do
  lib=$R/ab/libnsr_$v.so; [ "$v" = "new" ] && lib=$R/nerf_sr_amd/libnsr.so
  rm -rf /tmp/ta_$v
  (cd /tmp && NSR_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ta_$v -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
  echo "== $v"; python3 -c "import csv,sys; [print(r[\"Name\"][:50], r[\"Calls\"], round(float(r[\"AverageNs\"])/1e3,1)) for r in list(csv.DictReader(open(sys.argv[1])))[:4]]" $(find /tmp/ta_$v -name "*kernel_stats.csv" | head -1)
done
