"""Collect the counter constants bench.py quotes (profiles/r4_pmc.json) -- run on the GPU box through gpurun:

    python scripts/pmc_collect.py gpurun_out/r4_pmc.json [precision ...]

Four separate rocprofv3 --pmc passes (never combined with other trace domains) over scripts/prof_mlp.py <precision> 2, i.e.
the fine-pass launch forward_rays makes (network + compositing of the tile's own rays; config #2: 190,512 rays x 128
samples), read from the LAST dispatch of the MLP kernel:
  pass A  GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
  pass B  SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES
  pass C  FETCH_SIZE          pass D  WRITE_SIZE
Derived (MI355X: 8 XCDs, 1,024 SIMDs): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE) (GRBM summed over
the XCDs), effective_clock_ghz = GRBM_GUI_ACTIVE / 8 / launch duration, hbm_bytes_per_launch = (2 x FETCH_SIZE +
WRITE_SIZE) x 1024 (the guide's gfx950 correction: FETCH_SIZE sees half of a wide coalesced read stream).
The file records the sha256 of the kernel sources (nerf_sr_amd.build.source_hash): bench.py drops the constants when the
sources have changed since."""
import collections, csv, glob, json, os, subprocess, sys, tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from nerf_sr_amd import build as nsr_build  # noqa: E402

PASSES = {
    "A": "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES".split(),
    "B": "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES".split(),
    "C": ["FETCH_SIZE"],
    "D": ["WRITE_SIZE"],
}


def run_pass(prec, counters, workdir):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", workdir, "-o", "run", "--",
           sys.executable, os.path.join(REPO, "scripts", "prof_mlp.py"), prec, "2"]
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    rows = []
    for f in glob.glob(os.path.join(workdir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in rows:
        if "mlp" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    last = max(per)
    return dict(per[last]), dur[last], len(per)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r4_pmc.json"
    precs = sys.argv[2:] or ["f16x3"]
    rep = {"csrc_sha256": nsr_build.source_hash(),
           "how": "scripts/pmc_collect.py: four separate rocprofv3 --pmc passes of scripts/prof_mlp.py (fine-pass launch: 190,512 rays "
                  "x 128 samples, network + compositing), last dispatch of the MLP kernel"}
    for prec in precs:
        c, ms = {}, {}
        for tag, ctr in PASSES.items():
            with tempfile.TemporaryDirectory(dir="/tmp") as wd:
                vals, d, n = run_pass(prec, ctr, wd)
            c.update(vals)
            ms[tag] = d
            print(prec, tag, {k: int(v) for k, v in vals.items()}, f"{d:.3f} ms, {n} dispatches", flush=True)
        gui = c["GRBM_GUI_ACTIVE"]
        waves = max(c.get("SQ_WAVES", 0.0), 1.0)
        wc = c["SQ_WAVE_CYCLES"]
        rep[prec] = {
            "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
            "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
            "hbm_note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE reads half of a wide coalesced stream on gfx950 (guide, "
                        "HBM section), WRITE_SIZE uncalibrated.  Expected from the data structures: reads z (R, 128) 97.5 MB + rays "
                        "6.1 MB + the 2.4 MB weight stream (L2-resident); writes fine_weights (R, 128) 97.5 MB + rgb / depth / opacity 3.8 MB",
            "mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * gui),
            "effective_clock_ghz": gui / 8.0 / (ms["A"] * 1e-3) / 1e9,
            "launch_ms_under_pmc": ms["A"],
            "wave_cycles_split": {"active": c["SQ_ACTIVE_INST_ANY"] / wc, "issue_wait": c["SQ_WAIT_INST_ANY"] / wc,
                                  "parked": max(0.0, 1.0 - (c["SQ_ACTIVE_INST_ANY"] + c["SQ_WAIT_INST_ANY"]) / wc)},
            "instructions_per_wave_tile": {k.replace("SQ_INSTS_", "").lower(): c[k] / waves for k in c if k.startswith("SQ_INSTS_")},
            "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
            "raw": {k: v for k, v in c.items()},
        }
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
