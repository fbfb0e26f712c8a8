#!/bin/bash
# SQ / GRBM counters of the training step's kernels (one --pmc pass; 1 warm-up + 2 timed steps, sums / 3).
# TRAIN_PREC=<train precision> TAG=<suffix of the output files> select the path (default f16x3, no suffix)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/train_traffic
rm -rf /tmp/tsq
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/tsq -o run -- python $R/bench.py --mode train --train-precision ${TRAIN_PREC:-f16x3} --steps 2 --warmup 1 --no-cpu-baseline > /tmp/tsq.log 2>&1)
python - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/tsq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1].strip()[:40]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"]) / 3.0
out = {}
for n, c in acc.items():
    if c.get("GRBM_GUI_ACTIVE", 0) < 1e5: continue
    wc = c["SQ_WAVE_CYCLES"]
    out[n] = {"gui_active_cycles_per_step": int(c["GRBM_GUI_ACTIVE"]),
              "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * c["GRBM_GUI_ACTIVE"]), 4),
              "wave_cycles_split": {"active": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3), "issue_wait": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                                    "parked": round(c["SQ_WAIT_ANY"] / wc, 3)}}
json.dump(out, open("$R/gpurun_out/train_traffic/SQ${TAG}.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY

# second pass: instruction mix and LDS bank conflicts (the weight-gradient kernel's transposing reads)
rm -rf /tmp/tsq2
(cd /tmp && rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d /tmp/tsq2 -o run -- python $R/bench.py --mode train --train-precision ${TRAIN_PREC:-f16x3} --steps 2 --warmup 1 --no-cpu-baseline > /tmp/tsq2.log 2>&1)
python - <<PY
import csv, collections, glob, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/tsq2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1].strip()[:40]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"]) / 3.0
out = {n: {k: int(v) for k, v in c.items()} for n, c in acc.items() if c.get("SQ_INSTS_MFMA", 0) > 1e4}
for n, c in out.items():
    c["bank_conflict_cycles_per_lds_inst"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_INSTS_LDS", 1), 1), 3)
json.dump(out, open("$R/gpurun_out/train_traffic/SQ_insts${TAG}.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
