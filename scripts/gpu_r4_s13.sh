#!/bin/bash
# round-4 session 13: new boundary tests, CPU baseline sweep, the default bench line with its new sub-objects
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s13; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_integration.py "tests/test_gpu_frames.py::test_sharded_render_is_bit_identical_config4" -x -q -m gpu > $O/pytest_new.log 2>&1; tail -3 $O/pytest_new.log
timeout 600 python scripts/cpu_sweep.py $O/cpu_sweep.json 4096 > $O/cpu_sweep.log 2>&1; tail -2 $O/cpu_sweep.log
cp $O/cpu_sweep.json profiles/r4_cpu_sweep.json 2>/dev/null
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -4 $O/bench.err
