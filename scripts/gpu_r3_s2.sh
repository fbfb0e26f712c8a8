#!/bin/bash
# round 3, GPU session 2: TRAPSTS probe, training tests on the job-table weight gradients + A/B against the per-product
# launches, image (RGBA) tests, trained-field tests, oracle thread scaling
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3s2
mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O2 scripts/trapsts_probe.hip -o /tmp/trapsts_probe > /dev/null 2>&1 && timeout 60 /tmp/trapsts_probe 2>&1 | tee $O/trapsts.txt
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_image.py -q -m gpu -x > $O/train_tests.log 2>&1; echo "train+image tests rc $?" | tee $O/summary.txt
tail -4 $O/train_tests.log | cut -c1-300 | tee -a $O/summary.txt
for r in 1 2 3; do
  for v in 1 0; do
    NSR_WGRAD_JOBS=$v timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r NSR_WGRAD_JOBS=$v  ms_per_step %.3f  rays/s %.0f  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))" | tee -a $O/summary.txt
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/train_traced.log 2>&1)
head -12 $O/trace_train/run_kernel_stats.csv | cut -c1-220 | tee -a $O/summary.txt
NSR_PARITY_REPORT=$O/parity_trained_tests.json timeout 1200 python -m pytest tests/test_gpu_trained.py "tests/test_gpu_frames.py::test_sharp_field_envelope_parity" tests/test_gpu_parity.py -q -m gpu -s -x > $O/trained_tests.log 2>&1; echo "trained/sharp/parity tests rc $?" | tee -a $O/summary.txt
grep -E "^\[trained|^\.?\[config|passed|failed|Error" $O/trained_tests.log | cut -c1-400 | tail -40 | tee -a $O/summary.txt
python - <<'PY' 2>&1 | tee -a $O/summary.txt
import time, torch, numpy as np
from tests import trained_field as tf
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc
sd_c, sd_f = make_state_dict(99), make_state_dict(100)
import nerf_sr_amd.cameras as cam
rays = oc.subpixel_ray_grid(torch.from_numpy(cam.spiral_pose(0.4)), 378, 504, cam.llff_focal(504), 2, True, 0.0, 1.0).reshape(-1, 8)[90000:90000 + 4096]
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        t0 = time.time(); oc.forward_rays(oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), rays, 64, 64, False); t1 = time.time()
        oc.forward_rays(oc.to_torch_sd(sd_c, torch.float64), oc.to_torch_sd(sd_f, torch.float64), rays.double(), 64, 64, False); t2 = time.time()
    print(f"oracle 4096 rays, {th} threads: fp32 {t1 - t0:.1f} s, fp64 {t2 - t1:.1f} s")
PY
