#!/bin/bash
# conv_halo_kernel ablations on one box: per-launch durations of one refinement pass for each library variant
# (new = product; nopatch / nobar / nobdma / noepi = -DNSR_ABL_HALO_NO_PATCH / _NO_BARRIER / _NO_BDMA / _NO_EPILOGUE: wrong results,
# timing only), one line per launch, columns = variants
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/haloabl; mkdir -p $O; cd $R
for v in "$@"; do
  lib=$R/nerf_sr_amd/libnsr_$v.so; [ "$v" = "new" ] && lib=$R/nerf_sr_amd/libnsr.so
  rm -rf /tmp/rs_$v
  (cd /tmp && NSR_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rs_$v -o run -- python $R/scripts/refine_out.py $O/$v.pt 3 > $O/$v.log 2>&1)
  grep -h "per frame" $O/$v.log
done
python - "$@" <<'PY'
import csv, glob, sys
cols = {}
for v in sys.argv[1:]:
    rows = []
    for f in glob.glob(f"/tmp/rs_{v}/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "gemm_f16x3" in r["Kernel_Name"] or "conv_halo" in r["Kernel_Name"]]
    n = len(rows) // 4
    cols[v] = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-n:]]
names = cols[sys.argv[1]]
print("%-44s" % "launch", *["%9s" % v for v in sys.argv[1:]])
for i, (nm, _) in enumerate(names):
    k = nm.find("conv_halo") if "conv_halo" in nm else nm.find("gemm_f16x3")
    print("%-44s" % nm[k:k + 42], *["%9.1f" % cols[v][i][1] for v in sys.argv[1:]])
print("%-44s" % "sum", *["%9.1f" % sum(d for _, d in cols[v]) for v in sys.argv[1:]])
PY
