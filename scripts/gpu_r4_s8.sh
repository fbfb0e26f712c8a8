#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s8; mkdir -p $O; cd $R
for v in "" _tlaund; do echo "== libnsr$v"; NSR_LIB_PATH=$R/nerf_sr_amd/libnsr$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "f16x3" 2>&1 | tail -2; done
bash scripts/gpu_ab.sh r4s8 5 r3 gap tlaund new
