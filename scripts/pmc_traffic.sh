#!/bin/bash
# HBM traffic of the MLP launches: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots), per guide.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
prec=${1:-f16x3}
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $R/gpurun_out/traffic_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic_$c -o run -- python $R/scripts/prof_mlp.py $prec 2 > $R/gpurun_out/traffic_$c/log.txt 2>&1)
  python - <<PY
import csv, collections, glob
for f in glob.glob("$R/gpurun_out/traffic_$c/*counter_collection.csv"):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "mlp" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
    print("$c", "$prec", {k: v for k, v in acc.items()})
PY
done
