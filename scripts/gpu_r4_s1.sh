#!/bin/bash
# round-4 session 1: where does a tile's time go (timeline build), and was round 1's library faster on the same box?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s1; mkdir -p $O
cd $R
NSR_LIB_PATH=$R/nerf_sr_amd/libnsr_tl.so TL_SAMPLES=128 timeout 300 python scripts/timeline.py $O/timeline_128.json > $O/timeline_128.log 2>&1
NSR_LIB_PATH=$R/nerf_sr_amd/libnsr_tl.so TL_SAMPLES=64 timeout 300 python scripts/timeline.py $O/timeline_64.json > $O/timeline_64.log 2>&1
timeout 600 python scripts/ab_libs.py 6 r1=$R/nerf_sr_amd/libnsr_r1.so r3=$R/nerf_sr_amd/libnsr.so tl=$R/nerf_sr_amd/libnsr_tl.so --json $O/ab_r1_r3.json > $O/ab_r1_r3.log 2>&1
tail -5 $O/timeline_128.log; tail -3 $O/ab_r1_r3.log
