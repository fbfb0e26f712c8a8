"""Counter picture of the refinement pass's GEMM launches (csrc/nsr_gemm_f16.hip) -- run on the GPU box through gpurun:

    python scripts/pmc_refine.py gpurun_out/r3_refine_pmc.json

Separate rocprofv3 --pmc passes (never combined with other trace domains) over scripts/prof_refine.py 256 1 (two passes over
one 800 x 800 frame of BASELINE config #5), summed per kernel instantiation over all of its dispatches:
  pass A  GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
  pass B  SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES
  pass C  GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS   (LDS-array cycles, LDS issue stalls; skipped if the names are refused)
Derived per kernel (MI355X: 8 XCDs, 256 CUs, 1,024 SIMDs; GRBM_GUI_ACTIVE comes summed over the XCDs):
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE),  lds_busy = SQ_LDS_IDX_ACTIVE / (32 x GRBM_GUI_ACTIVE),
  the wave-cycle split as in scripts/pmc_collect.py."""
import collections, csv, glob, json, os, re, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nerf_sr_amd import build as nsr_build  # noqa: E402

PASSES = {
    "A": "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES".split(),
    "B": "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES".split(),
    "C": "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS".split(),
}


def short(name):
    m = re.search(r"(gemm_f16x3_kernel<[^>]*>|[A-Za-z0-9_]+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def run_pass(counters, workdir):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", workdir, "-o", "run", "--",
           sys.executable, os.path.join(REPO, "scripts", "prof_refine.py"), "256", "1"]
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for f in glob.glob(os.path.join(workdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen[k]:
                seen[k].add(r["Dispatch_Id"])
                per[k]["_ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for k in per:
        per[k]["_dispatches"] = len(seen[k])
    return per


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3_refine_pmc.json"
    rep = {"csrc_sha256": nsr_build.source_hash(), "how": __doc__.split("\n\n")[1], "kernels": {}, "errors": {}}
    data = {}
    for tag, ctr in PASSES.items():
        try:
            with tempfile.TemporaryDirectory(dir="/tmp") as wd:
                data[tag] = run_pass(ctr, wd)
            print("pass", tag, "ok:", len(data[tag]), "kernels", flush=True)
        except Exception as e:  # a refused counter name must not lose the other passes
            rep["errors"][tag] = repr(e)[:300]
            print("pass", tag, "FAILED", rep["errors"][tag], flush=True)
    a = data.get("A", {})
    for k, c in sorted(a.items(), key=lambda kv: -kv[1]["_ms"]):
        gui, wc = c["GRBM_GUI_ACTIVE"], max(c["SQ_WAVE_CYCLES"], 1.0)
        e = {"dispatches": int(c["_dispatches"]), "ms_under_pmc": round(c["_ms"], 3),
             "mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * gui),
             "effective_clock_ghz": gui / 8.0 / (c["_ms"] * 1e-3) / 1e9,
             "wave_cycles_split": {"active": c["SQ_ACTIVE_INST_ANY"] / wc, "issue_wait": c["SQ_WAIT_INST_ANY"] / wc,
                                   "parked": c["SQ_WAIT_ANY"] / wc}}
        b = data.get("B", {}).get(k)
        if b:
            waves = max(b["SQ_WAVES"], 1.0)
            e["instructions_per_wave"] = {n.replace("SQ_INSTS_", "").lower(): round(b[n] / waves, 1) for n in b if n.startswith("SQ_INSTS_")}
            e["lds_bank_conflict_cycles_per_wave"] = round(b["SQ_LDS_BANK_CONFLICT"] / waves, 1)
        cc = data.get("C", {}).get(k)
        if cc:
            e["lds_busy"] = cc["SQ_LDS_IDX_ACTIVE"] / (32.0 * cc["GRBM_GUI_ACTIVE"])
            e["lds_issue_wait_share_of_wave_cycles"] = cc["SQ_WAIT_INST_LDS"] / wc
        rep["kernels"][k] = e
        print(k, json.dumps(e), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
