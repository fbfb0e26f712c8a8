#!/bin/bash
# Round-6 evidence of the committed sources (builder-run): the whole -m gpu suite, the counter constants bench.py quotes
# (profiles/r6_pmc.json, r6_train_traffic.json, r6_refine_pmc.json: tied to the source hash), the default bench line, rocprofv3
# kernel statistics of the same command, the training line and its kernel statistics / counters, config #5 with the refinement
# pass, the per-round parity record and the training drift record.   usage: gpu_r6_final.sh [notests] [noparity]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6final; mkdir -p $O; cd $R
if [[ " $* " != *" notests "* ]]; then
  timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; tail -3 $O/pytest_full.log | tee $O/summary.txt
fi
timeout 600 python scripts/pmc_collect.py $O/r6_pmc.json f16x3 2>&1 | tail -2 | tee -a $O/summary.txt
cp $O/r6_pmc.json profiles/r6_pmc.json
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
json.dump(rec, open("$R/profiles/r6_train_traffic.json", "w"), indent=1); json.dump(rec, open("$O/r6_train_traffic.json", "w"), indent=1)
print("train hbm bytes per step", rec["hbm_bytes_per_step"])
PY
timeout 500 bash scripts/pmc_train_sq.sh > $O/train_sq.log 2>&1
python - <<PY
import json, sys
sys.path.insert(0, "$R")
from nerf_sr_amd import build as b
rec = {"how": "scripts/pmc_train_sq.sh: two rocprofv3 --pmc passes over bench.py --mode train (3 identical steps, sums / 3)", "csrc_sha256": b.source_hash(),
       "cycles": json.load(open("$R/gpurun_out/train_traffic/SQ.json")), "instructions": json.load(open("$R/gpurun_out/train_traffic/SQ_insts.json"))}
json.dump(rec, open("$O/r6_train_sq.json", "w"), indent=1)
print({k: (v["mfma_busy"], v["wave_cycles_split"]) for k, v in rec["cycles"].items()})
PY
timeout 600 python scripts/pmc_refine.py $O/r6_refine_pmc.json > $O/refine_pmc.log 2>&1; tail -2 $O/refine_pmc.log | cut -c1-300 | tee -a $O/summary.txt
cp $O/r6_refine_pmc.json profiles/r6_refine_pmc.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-240 | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline --no-config4 --no-extras > $O/bench_traced.log 2>&1)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -4 $O/kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt; rm -rf $O/trace
timeout 300 python bench.py --mode train --steps 50 --warmup 10 > $O/train_bench.json 2>> $O/bench.err; tail -1 $O/train_bench.json | cut -c1-200 | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/train_traced.log 2>&1)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv; head -5 $O/train_kernel_stats.csv | cut -c1-160 | tee -a $O/summary.txt; rm -rf $O/trace
timeout 300 python bench.py --config 5 --with-refine --no-cpu-baseline > $O/config5_refine.json 2>> $O/bench.err
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/config3.json 2>> $O/bench.err
timeout 300 python bench.py --precision fp32 --no-cpu-baseline --no-config4 --no-extras > $O/fp32_bench.json 2>> $O/bench.err
timeout 300 python bench.py --n-importance 128 --no-cpu-baseline --no-config4 > $O/ni128_bench.json 2>> $O/bench.err
timeout 300 python bench.py --mode train --train-precision f16x3_gemm --steps 20 --warmup 5 --no-cpu-baseline > $O/train_bench_gemm.json 2>> $O/bench.err
timeout 300 python bench.py --mode train --train-precision fp32 --steps 20 --warmup 5 --no-cpu-baseline > $O/train_bench_fp32.json 2>> $O/bench.err
bash scripts/gpu_refine_stats.sh new > $O/refine_stats.log 2>&1; cp gpurun_out/refstats/new_kernel_stats.csv $O/refine_kernel_stats.csv; tail -3 $O/refine_stats.log | cut -c1-120
timeout 600 python scripts/train_drift.py 200 > $O/r6_train_drift.json 2>> $O/bench.err
# round 6: the backward chain on 3 / 2 / mixed (the default) / 1 MFMA terms -- gradients vs the fp64 oracle, bench scale vs the fp32-gradient path, trajectory;
# the step time of each (interleaved); the multi-seed drift record
timeout 900 python scripts/bwd_terms_check.py > $O/r6_bwd_terms_gpu.json 2>> $O/bench.err
for r in 1 2 3; do for v in f16x3_bwd3 f16x3_bwd2 f16x3_bwdm f16x3_bwd1; do
  timeout 300 python bench.py --mode train --train-precision $v --steps 40 --warmup 8 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r $v  ms_per_step %.3f' % d['ms_per_step'])"
done; done | tee $O/r6_train_terms_ab.txt
timeout 300 python bench.py --mode train --train-precision f16x3_bwd1 --steps 50 --warmup 10 --no-cpu-baseline > $O/train_bench_bwd1.json 2>> $O/bench.err
timeout 600 python scripts/train_drift_seeds.py 6 > $O/r6_train_drift_seeds.json 2>> $O/bench.err
rm -rf /tmp/tl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
python scripts/train_timeline.py /tmp/tl > $O/r6_train_timeline.txt; tail -2 $O/r6_train_timeline.txt | tee -a $O/summary.txt
timeout 600 bash scripts/power_clock_probe.sh > $O/r6_power_clock_probe.txt 2>&1; grep -E "^(idle|render|zeros|random|hilo|train) " $O/r6_power_clock_probe.txt | tee -a $O/summary.txt
timeout 900 python scripts/fp32_ray_probe.py fp32 5 65536 > $O/r6_fp32_ray_probe.json 2>> $O/bench.err
if [[ " $* " != *" noparity "* ]]; then
  timeout 1500 python scripts/parity_record.py $O/r6_parity_report.json 65536 16384 4000 2>&1 | tail -7 | cut -c1-200 | tee -a $O/summary.txt
fi
for f in bench config5_refine config3 fp32_bench ni128_bench train_bench train_bench_bwd1 train_bench_gemm train_bench_fp32; do python -c "
import json; d = json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'], 3), round(d['roofline']['frac'], 4), (d.get('refine') or {}).get('refine_ms'))"; done | tee -a $O/summary.txt
