#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_bench
mkdir -p $O
cd $R
( time timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
def short(x, n=300): return json.dumps(x)[:n]
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "non_mlp", d.get("non_mlp_ms_per_step"))
for k in ("config4", "config5", "train", "config1"):
    v = d.get(k)
    print(k, short({kk: v[kk] for kk in v if kk not in ("config", "roofline", "refine", "workload")} if isinstance(v, dict) else v, 700))
print("train.cpu", short(d.get("train", {}).get("cpu_baseline"), 600))
print("c1.cpu", short(d.get("config1", {}).get("cpu_baseline"), 600))
print("train.roofline", short(d.get("train", {}).get("roofline"), 900))
print("parity", short(d.get("parity"), 600))
PY
tail -3 $O/bench.err
timeout 900 python -m pytest tests/test_gpu_frames.py -q -m gpu -k "bench_two_ranks" 2>&1 | tail -15 | cut -c1-400
