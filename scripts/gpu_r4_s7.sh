#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s7; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16x3" > $O/pytest_quick.log 2>&1; tail -2 $O/pytest_quick.log
bash scripts/gpu_ab_tl.sh r4s7 5 r3 gap np new
