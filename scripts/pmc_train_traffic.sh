#!/bin/bash
# HBM traffic of the training step's kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots), per the guide.
# 1 warm-up + 2 timed steps = 3 identical steps; the sums below are divided by 3.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/train_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tt_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tt_$c -o run -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > /tmp/tt_$c.log 2>&1)
  python - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("/tmp/tt_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1].strip()[:40]
        acc[n] += float(r["Counter_Value"]); cnt[n] += 1
out = {n: {"kb_per_step": v / 3.0, "dispatches_per_step": cnt[n] / 3.0} for n, v in sorted(acc.items(), key=lambda x: -x[1]) if v > 3000}
json.dump(out, open("$R/gpurun_out/train_traffic/$c.json", "w"), indent=1)
print("$c", json.dumps(out)[:1500])
PY
done
