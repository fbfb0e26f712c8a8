#!/bin/bash
# re-collect ONLY the counter constants bench.py quotes (they are tied to the source hash) and the default bench line, after a
# source change that moves no instruction (a header comment): scripts/gpu_r6_final.sh holds the full protocol
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6pmc; mkdir -p $O; cd $R
timeout 600 python scripts/pmc_collect.py $O/r6_pmc.json f16x3 2>&1 | tail -1; cp $O/r6_pmc.json profiles/r6_pmc.json
timeout 400 bash scripts/pmc_train_traffic.sh 2>&1 | tail -1 | cut -c1-120
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from nerf_sr_amd import build as b
f = json.load(open("$R/gpurun_out/train_traffic/FETCH_SIZE.json")); w = json.load(open("$R/gpurun_out/train_traffic/WRITE_SIZE.json"))
fk = sum(v["kb_per_step"] for v in f.values()); wk = sum(v["kb_per_step"] for v in w.values())
rec = {"how": "scripts/pmc_train_traffic.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --mode train (3 identical steps, sums / 3); hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per the guide's gfx950 correction",
       "csrc_sha256": b.source_hash(), "hbm_bytes_per_step": int((2 * fk + wk) * 1024),
       "fetch_kb_per_step_by_kernel": f, "write_kb_per_step_by_kernel": w}
json.dump(rec, open("$O/r6_train_traffic.json", "w"), indent=1)
print("train hbm bytes per step", rec["hbm_bytes_per_step"])
PY
timeout 600 python scripts/pmc_refine.py $O/r6_refine_pmc.json > $O/refine_pmc.log 2>&1; tail -1 $O/refine_pmc.log | cut -c1-120
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-200
