#!/bin/bash
# development aid: interleaved A/B of the refinement GEMM's switches on ONE box (one 800 x 800 pass of config #5 per sample,
# scripts/prof_refine.py 256 3), then the refinement tests on the product library.
# usage: gpu_refine_ab.sh "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]      ("" = the product's defaults)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refine_ab
mkdir -p $O
cd $R
: > $O/summary.txt
for r in 1 2 3; do
  for v in "$@"; do
    out=$(env $v timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -2 | tr '\n' ' ')
    echo "round $r [$v]: $out" | tee -a $O/summary.txt
  done
done
[ -n "$AB_SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_gpu_refine.py tests/test_gpu_frames.py::test_config5_composed_small_vs_oracles -q -m gpu -x 2>&1 | tail -2 | tee -a $O/summary.txt
