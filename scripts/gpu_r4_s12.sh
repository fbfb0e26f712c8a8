#!/bin/bash
# round-4 session 12: product (one tile per workgroup, gap-aware) vs persistent builds, then the whole GPU suite on the product
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s12; mkdir -p $O; cd $R
bash scripts/gpu_ab_tl.sh r4s12 5 r3 gap new ps pe r1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log
