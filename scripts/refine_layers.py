import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last refine pass: take the last 40 kernels that are gemm/im2col-ish
ks = [r for r in rows if any(t in r["Kernel_Name"] for t in ("gemm_f16x3", "im2col", "max_refs", "gather", "stitch", "gemm_kernel"))]
n = len(ks) // 4   # 4 passes (1 warm + 3)
last = ks[-n:]
tot = 0
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    print(f"{name:42s} grid {r['Grid_Size_X']:>9s} wg {r['Workgroup_Size_X']:>4s} {d:9.1f} us")
print("total", tot)
