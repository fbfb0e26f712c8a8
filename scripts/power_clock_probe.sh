#!/bin/bash
# Round 6 (VERDICT r5 "next" #4b): is the 1.8 GHz the headline kernel runs at the POWER cap, or a thermal / DVFS artefact of its
# launch pattern?  Samples rocm-smi (socket power, shader clock, temperatures, the power cap) every 0.1 s while (1) bench.py
# renders config #2 frames back to back, (2) a bare v_mfma_f32_32x32x16_f16 loop runs on zero operands, (3) on weight-like x
# activation-like operand bits, (4) on the kernel's own (hi, lo) operand mix -- each for several seconds -- and (5) idle.
# usage (GPU box): bash scripts/power_clock_probe.sh > gpurun_out/power_clock_probe.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o /tmp/mfma_ceiling || exit 1
rocm-smi --showmaxpower --showpower --showclocks --showtemp 2>&1 | grep -v "^$" | head -40
sample() {   # $1 = label; runs until the file /tmp/pc_stop exists
  rm -f /tmp/pc_stop /tmp/pc_$1.log
  ( while [ ! -e /tmp/pc_stop ]; do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null >> /tmp/pc_$1.log; echo >> /tmp/pc_$1.log; sleep 0.1; done ) &
  SAMPLER=$!
}
stop() { touch /tmp/pc_stop; wait $SAMPLER 2>/dev/null; }
sample idle; sleep 3; stop
sample render; python bench.py --steps 80 --warmup 3 --no-cpu-baseline --no-extras --no-config4 > /tmp/pc_render.json 2>/dev/null; stop
sample zeros; /tmp/mfma_ceiling hold 0 6; stop
sample random; /tmp/mfma_ceiling hold 1 6; stop
sample hilo; /tmp/mfma_ceiling hold 2 6; stop
sample train; python bench.py --mode train --steps 1500 --warmup 10 --no-cpu-baseline > /tmp/pc_train.json 2>/dev/null; stop
python - <<'PY'
import json, re, statistics
def series(label):
    pw, sc, tp = [], [], []
    for line in open(f"/tmp/pc_{label}.log"):
        line = line.strip()
        if not line.startswith("{"): continue
        try: d = json.loads(line)
        except Exception: continue
        c = d.get("card0", {})
        for k, v in c.items():
            kl = k.lower()
            try:
                if "power" in kl and "max" not in kl and "cap" not in kl: pw.append(float(v))
                elif "sclk" in kl: sc.append(float(re.sub(r"[^0-9.]", "", str(v).split("Mhz")[0].strip("() "))))
                elif "temperature" in kl and "junction" in kl: tp.append(float(v))
            except Exception: pass
    return pw, sc, tp
for label in ("idle", "render", "zeros", "random", "hilo", "train"):
    pw, sc, tp = series(label)
    f = lambda x: ("n=%d median %.0f max %.0f" % (len(x), statistics.median(x), max(x))) if x else "none"
    print(f"{label:7s} power W: {f(pw)}   sclk MHz: {f(sc)}   junction C: {f(tp)}")
for label in ("render", "train"):
    try:
        d = json.loads([l for l in open(f"/tmp/pc_{label}.json") if l.startswith("{")][0])
        print(label, "ms_per_step", d["ms_per_step"], "value", d["value"], d.get("roofline", {}).get("frac"))
    except Exception as e: print(label, "bench line unreadable:", e)
PY
head -c 1500 /tmp/pc_render.log
