#!/bin/bash
# development aid: interleaved A/B of the refinement GEMM's staging-row assignment (libnsr.so vs libnsr_nowrows.so, built with
# -DNSR_GEMM_NO_WROWS) and of the straight-line k-steps on full column tiles (NSR_GEMM_FULLN=0 keeps the column test), one
# 800 x 800 pass of config #5 per sample (scripts/prof_refine.py 256 3), then the refinement tests on the product library.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refine_ab
mkdir -p $O
cd $R
: > $O/summary.txt
for r in 1 2 3; do
  for lib in libnsr.so libnsr_nowrows.so; do
    for f in 1 0; do
      NSR_LIB_PATH=$R/nerf_sr_amd/$lib NSR_GEMM_FULLN=$f timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -2 | tr '\n' ' ' | sed "s/^/round $r $lib fulln=$f: /" | tee -a $O/summary.txt; echo | tee -a $O/summary.txt
    done
  done
done
timeout 600 python -m pytest tests/test_gpu_refine.py tests/test_gpu_frames.py::test_config5_composed_small_vs_oracles -q -m gpu -x 2>&1 | tail -2 | tee -a $O/summary.txt
