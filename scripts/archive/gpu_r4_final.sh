#!/bin/bash
# Round-4 evidence of the committed sources (builder-run): the whole -m gpu suite, the counter constants bench.py quotes
# (profiles/r4_pmc.json, r4_train_traffic.json, tied to the source hash), the default bench line, rocprofv3 kernel statistics of the
# same command, the training line and its kernel statistics, config #5 with the refinement pass and its counters.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4final; mkdir -p $O; cd $R
if [ "$1" != "notests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log | tee $O/summary.txt
fi
timeout 600 python scripts/pmc_collect.py $O/r4_pmc.json f16x3 2>&1 | tail -2 | tee -a $O/summary.txt
cp $O/r4_pmc.json profiles/r4_pmc.json
timeout 400 bash scripts/pmc_train_traffic.sh 2>&1 | tail -2 | cut -c1-200 | tee -a $O/summary.txt
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from nerf_sr_amd import build as b
f = json.load(open("$R/gpurun_out/train_traffic/FETCH_SIZE.json")); w = json.load(open("$R/gpurun_out/train_traffic/WRITE_SIZE.json"))
fk = sum(v["kb_per_step"] for v in f.values()); wk = sum(v["kb_per_step"] for v in w.values())
rec = {"how": "scripts/pmc_train_traffic.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --mode train (3 identical steps, sums / 3); hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per the guide's gfx950 correction",
       "csrc_sha256": b.source_hash(), "hbm_bytes_per_step": int((2 * fk + wk) * 1024),
       "fetch_kb_per_step_by_kernel": f, "write_kb_per_step_by_kernel": w}
json.dump(rec, open("$R/profiles/r4_train_traffic.json", "w"), indent=1); json.dump(rec, open("$R/gpurun_out/r4final/r4_train_traffic.json", "w"), indent=1)
print("train hbm bytes per step", rec["hbm_bytes_per_step"])
PY
timeout 600 python scripts/pmc_refine.py $O/r4_refine_pmc.json > $O/refine_pmc.log 2>&1; tail -2 $O/refine_pmc.log | cut -c1-300 | tee -a $O/summary.txt
cp $O/r4_refine_pmc.json profiles/r4_refine_pmc.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-240 | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline --no-config4 --no-extras > $O/bench_traced.log 2>&1)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -4 $O/kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt; rm -rf $O/trace
timeout 300 python bench.py --mode train --steps 50 --warmup 10 > $O/train_bench.json 2>> $O/bench.err; tail -1 $O/train_bench.json | cut -c1-200 | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/train_traced.log 2>&1)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv; head -5 $O/train_kernel_stats.csv | cut -c1-160 | tee -a $O/summary.txt; rm -rf $O/trace
timeout 300 python bench.py --config 5 --with-refine --no-cpu-baseline > $O/config5_refine.json 2>> $O/bench.err
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/config3.json 2>> $O/bench.err
timeout 300 python bench.py --precision fp32 --no-cpu-baseline --no-config4 --no-extras > $O/fp32_bench.json 2>> $O/bench.err
