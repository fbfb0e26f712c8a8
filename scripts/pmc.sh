#!/bin/bash
# usage: scripts/pmc.sh <tag> <precision> <counter...>   (run on the GPU box through gpurun)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; prec=$2; shift 2
mkdir -p $R/gpurun_out/$tag
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/$tag -o run -- python $R/scripts/prof_mlp.py $prec 2 > $R/gpurun_out/$tag/log.txt 2>&1
tail -5 $R/gpurun_out/$tag/log.txt
ls $R/gpurun_out/$tag
python - <<PY
import csv, collections, glob
for f in glob.glob("$R/gpurun_out/$tag/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if "mlp" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d, c in acc.items():
        print(d, {k: int(v) for k, v in c.items()})
PY
