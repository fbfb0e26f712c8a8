#!/bin/bash
# round-4 session 2: gap-aware k-steps (block_mma3) -- bit-identity vs the round-3 library, interleaved A/B, timeline, parity tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s2; mkdir -p $O
cd $R
timeout 600 python scripts/ab_libs.py 5 r3=$R/nerf_sr_amd/libnsr_r3.so new=$R/nerf_sr_amd/libnsr.so r1=$R/nerf_sr_amd/libnsr_r1.so --json $O/ab.json > $O/ab.log 2>&1
tail -4 $O/ab.log
NSR_LIB_PATH=$R/nerf_sr_amd/libnsr_tl.so TL_SAMPLES=128 timeout 300 python scripts/timeline.py $O/timeline_128.json > $O/timeline_128.log 2>&1
grep -A3 '"median"' $O/timeline_128.json | head -0; python - <<PY
import json; d=json.load(open("$O/timeline_128.json"))
print({k: v["median"] for k, v in d["phases_cycles"].items()}, d["start_to_start_on_a_cu_cycles"], d.get("ksteps_of_one_trunk_chunk_cycles"))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_status.py -x -q -m gpu > $O/pytest_parity.log 2>&1; tail -5 $O/pytest_parity.log
