"""Development aid: per-kernel time of the LAST training step in a rocprofv3 kernel trace of scripts/prof_train.py.
usage: train_breakdown.py <run_kernel_trace.csv> [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = adam[-3] + 1, adam[-1] + 1          # two adam launches (coarse, fine) close a step
step = rows[lo:hi]
tot = collections.OrderedDict()
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1][:40]
    if "gemm_kernel" in r["Kernel_Name"]:
        n = "gemm" + r["Kernel_Name"].split("gemm_kernel")[1][:12]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = tot.setdefault(n, [0, 0.0]); c[0] += 1; c[1] += d
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
print(f"step wall {wall:.0f} us, kernel sum {sum(v[1] for v in tot.values()):.0f} us")
for n, (c, d) in sorted(tot.items(), key=lambda x: -x[1][1]): print(f"  {n:44s} {c:4d} calls {d:9.1f} us")
if "-v" in sys.argv:
    for r in step:
        if "gemm_kernel" in r["Kernel_Name"]:
            print("   ", r["Kernel_Name"].split("gemm_kernel")[1][:10], r["Grid_Size_X"], r["Grid_Size_Y"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
