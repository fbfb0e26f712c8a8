"""One training step as the GPU saw it: every kernel of the LAST step of a rocprofv3 --kernel-trace run in start order, with its
duration and the idle gap in front of it.  usage: python scripts/train_timeline.py <dir with *_kernel_trace.csv> [steps in the run]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
# a step starts at its weight-range check (one launch per step since round 6; two back to back before)
starts = [i for i, r in enumerate(rows) if name(r).startswith("check_weights") and (i == 0 or not name(rows[i - 1]).startswith(("check_weights", "pack_")))]
i0 = starts[-2] if len(starts) > 1 else 0
i1 = starts[-1] if len(starts) > 1 else len(rows)
t_prev = None
busy = gaps = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if t_prev is None else s - t_prev
    print(f"{name(r):50s} {(e - s) / 1e3:9.1f} us   gap {gap / 1e3:7.1f} us")
    busy += e - s
    gaps += max(gap, 0)
    t_prev = e
print(f"step: kernels {busy / 1e3:.1f} us + gaps {gaps / 1e3:.1f} us = {(busy + gaps) / 1e3:.1f} us, {i1 - i0} launches")
