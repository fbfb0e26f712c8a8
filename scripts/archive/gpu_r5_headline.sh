#!/bin/bash
# round 5, headline kernel: upper bounds of the two named experiments (ablation builds), interleaved on one box; parity record
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5_headline; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh r5_headline 4 new noamax nodensity
cat $O/ab.json | head -c 1500; echo
( time timeout 1500 python scripts/parity_record.py $O/r5_parity_report.json 65536 16384 4000 ) 2>&1 | tail -12 | cut -c1-400
