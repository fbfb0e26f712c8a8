#!/bin/bash
# round 6 session 4: default = two-term chain; train tests, the two-rank training test, a step timeline, final A/B
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_s4; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_frames.py::test_bench_train_two_ranks_on_one_gpu_gloo -q --maxfail=10 --durations=8 2>&1 | tail -30 | tee $O/pytest.txt
rm -rf /tmp/tl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
python scripts/train_timeline.py /tmp/tl | tee $O/timeline.txt
