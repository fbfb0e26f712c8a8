#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s6; mkdir -p $O; cd $R
timeout 300 python scripts/debug_steps.py f16x3 > $O/dbg.log 2>&1 || { tail -5 $O/dbg.log; exit 1; }
tail -1 $O/dbg.log
bash scripts/gpu_ab_tl.sh r4s6 5 r3 gap np new
