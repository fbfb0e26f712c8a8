#!/bin/bash
# round 5, training step: the 200-step drift test + its record, HBM traffic of the step (PMC), SQ counters of the three big kernels
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_train2
mkdir -p $O
cd $R
scripts/bin/probe_tr16 2>&1 | tee $O/probe_tr16.txt
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -s -k "drift or status" 2>&1 | tail -12 | cut -c1-600 | tee $O/pytest_drift.txt
timeout 600 python scripts/train_drift.py 200 > $O/r5_train_drift.json 2> $O/drift.err; head -c 1500 $O/r5_train_drift.json
timeout 500 bash scripts/pmc_train_traffic.sh 2>&1 | tail -2 | cut -c1-300
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from nerf_sr_amd import build as b
f = json.load(open("$R/gpurun_out/train_traffic/FETCH_SIZE.json")); w = json.load(open("$R/gpurun_out/train_traffic/WRITE_SIZE.json"))
fk = sum(v["kb_per_step"] for v in f.values()); wk = sum(v["kb_per_step"] for v in w.values())
rec = {"how": "scripts/pmc_train_traffic.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --mode train (3 identical steps, sums / 3); hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per the guide's gfx950 correction",
       "csrc_sha256": b.source_hash(), "hbm_bytes_per_step": int((2 * fk + wk) * 1024),
       "fetch_kb_per_step_by_kernel": f, "write_kb_per_step_by_kernel": w}
json.dump(rec, open("$O/r5_train_traffic.json", "w"), indent=1)
print("train hbm bytes per step", rec["hbm_bytes_per_step"], "fetch kb", fk, "write kb", wk)
PY
timeout 500 bash scripts/pmc_train_sq.sh 2>&1 | tail -30 | cut -c1-300 | tee $O/sq.txt
