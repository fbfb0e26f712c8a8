#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_train3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_trained.py -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
for r in 1 2 3; do
  for lib in r5 r4; do
    p=nerf_sr_amd/libnsr.so; [ $lib = r4 ] && p=nerf_sr_amd/libnsr_r4.so
    [ -f $p ] || continue
    NSR_LIB_PATH=$R/$p timeout 300 python bench.py --mode train --steps 25 --warmup 5 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r lib=$lib  ms_per_step %.3f  losses %s' % (d['ms_per_step'], d['losses']))" | tee -a $O/ab.txt
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/traced.log 2>&1)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/train_kernel_stats.csv && head -16 $O/train_kernel_stats.csv | cut -c1-150
rm -rf $O/trace
timeout 500 bash scripts/pmc_train_traffic.sh 2>&1 | tail -2 | cut -c1-100
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
for k in list(f)[:6]: print("  fetch", k, int(f[k]["kb_per_step"]))
for k in list(w)[:5]: print("  write", k, int(w[k]["kb_per_step"]))
PY
