#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2c5
mkdir -p $O
cd $R
for b in 32 64 169; do timeout 200 python scripts/prof_refine.py $b 3 2>&1 | tail -3 | tee -a $O/summary.txt; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/scripts/prof_refine.py 32 1 > $O/traced.log 2>&1)
head -12 $O/trace/run_kernel_stats.csv | cut -c1-220 | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import csv, collections
rows = list(csv.DictReader(open("$O/trace/run_kernel_trace.csv")))
rows = [r for r in rows if "gemm_f16x3" in r["Kernel_Name"]]
# second frame only (skip warm-up): group by (grid, workgroup) shape
half = len(rows) // 2
acc = collections.OrderedDict()
for r in rows[half:]:
    key = (r["Kernel_Name"][:60].split("(")[0][-34:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Workgroup_Size_X", r.get("Workgroup_Size")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
for k, (n, us) in acc.items(): print(k, n, f"{us:.0f} us total, {us / n:.0f} us each")
PY
