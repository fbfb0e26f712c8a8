#!/bin/bash
# round 3, closing session after the refinement GEMM's staging / k-step fixes: training tests (the fp32-A GEMM kernels share the
# staging code), the refinement counters on the new build, the PMC constants + bench + kernel stats of the final sources
# (scripts/gpu_r3_pmc.sh), config #5 with the refinement pass.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3close
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_refine.py tests/test_gpu_frames.py::test_config5_composed_small_vs_oracles -q -m gpu -x 2>&1 | tail -2 | tee $O/summary.txt
timeout 300 python scripts/pmc_refine.py $O/r3_refine_pmc.json > $O/pmc_refine.log 2>&1; grep -c "^gemm" $O/pmc_refine.log | tee -a $O/summary.txt
for r in 1 2 3; do timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r: /" | tee -a $O/summary.txt; done
timeout 400 python bench.py --config 5 --with-refine --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err; tail -1 $O/bench_c5.json | cut -c1-200 | tee -a $O/summary.txt
python -c "import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('refine:', {k: d['refine'][k] for k in ('warp_ms','refine_ms','tiles')}, d['refine']['roofline']['achieved'])" 2>&1 | tee -a $O/summary.txt
bash scripts/gpu_r3_pmc.sh 2>&1 | tail -8 | cut -c1-300 | tee -a $O/summary.txt
