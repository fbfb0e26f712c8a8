#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_s7; mkdir -p $O; cd $R
bash scripts/gpu_r6_train_ab.sh "new:f16x3 base:f16x3 new:f16x3_bwd1 base:f16x3_bwd1" 2>&1 | grep -v "^==\|^void\|^nsr::\|^(anon" | tail -14
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_options.py -q --maxfail=10 2>&1 | tail -8
rm -rf /tmp/tl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
python scripts/train_timeline.py /tmp/tl | tee $O/timeline.txt | tail -45
