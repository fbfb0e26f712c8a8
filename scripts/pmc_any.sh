#!/bin/bash
# usage: scripts/pmc_any.sh <tag> <kernel-substring> <counter...> -- <python script and args>   (through gpurun)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; pat=$2; shift 2
ctr=()
while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
shift
mkdir -p $R/gpurun_out/$tag
cd /tmp
rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/$tag -o run -- python "$@" > $R/gpurun_out/$tag/log.txt 2>&1
PAT="$pat" python - <<PY
import csv, collections, glob, os
pat = os.environ["PAT"]
for f in glob.glob("$R/gpurun_out/$tag/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in rows:
        if pat in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[name] += 1
    for n, c in acc.items():
        print(n, "dispatches", cnt[n], {k: int(v) for k, v in c.items()})
PY
