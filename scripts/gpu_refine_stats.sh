#!/bin/bash
# per-kernel times of one refinement pass for the library variants named on the command line ("new" = product)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/refstats; mkdir -p $O; cd $R
for v in "$@"; do
  lib=$R/ab/libnsr_$v.so; [ "$v" = "new" ] && lib=$R/nerf_sr_amd/libnsr.so
  rm -rf /tmp/rs_$v
  (cd /tmp && NSR_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rs_$v -o run -- python $R/scripts/refine_out.py $O/$v.pt 3 > $O/$v.log 2>&1)
  tail -1 $O/$v.log
  cp $(find /tmp/rs_$v -name "*kernel_stats.csv" | head -1) $O/${v}_kernel_stats.csv
  head -8 $O/${v}_kernel_stats.csv | cut -c1-150
  python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("/tmp/rs_$v/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "gemm" in r["Kernel_Name"] or "conv_" in r["Kernel_Name"]]
n = len(rows) // 4                      # 4 passes; print the last one, launch by launch
for r in rows[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    nm = r["Kernel_Name"]; nm = nm[nm.find("conv_") if "conv_" in nm else nm.find("gemm_f16x3"):][:40]
    print(f"  {nm:42s} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9s} {d:9.1f} us")
PY
done
