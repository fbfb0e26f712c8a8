#!/bin/bash
# A/B + timeline of the current build: scripts/gpu_ab_tl.sh <tag> <rounds> name ...
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1
bash $R/scripts/gpu_ab.sh "$@"
O=$R/gpurun_out/$tag; cd $R
NSR_LIB_PATH=$R/nerf_sr_amd/libnsr_tl.so TL_SAMPLES=128 timeout 300 python scripts/timeline.py $O/timeline_128.json > $O/timeline_128.log 2>&1
python - <<PY
import json; d=json.load(open("$O/timeline_128.json"))
print({k: v["median"] for k, v in d["phases_cycles"].items()}, d["start_to_start_on_a_cu_cycles"], {k: v["median"] for k, v in d.get("ksteps_of_one_trunk_chunk_cycles", {}).items()})
PY
