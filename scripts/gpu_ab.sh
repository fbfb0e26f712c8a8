#!/bin/bash
# interleaved same-box A/B of the library variants named on the command line ("" = product): scripts/gpu_ab.sh <tag> <rounds> name ...
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; rounds=$2; shift 2
O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
args=""
for v in "$@"; do
  if [ "$v" = "new" ]; then args="$args new=$R/nerf_sr_amd/libnsr.so"; else args="$args $v=$R/ab/libnsr_$v.so"; fi
done
timeout 900 python scripts/ab_libs.py $rounds $args --json $O/ab.json > $O/ab.log 2>&1
grep 'max |' $O/ab.log; tail -1 $O/ab.log
