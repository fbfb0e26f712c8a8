#!/bin/bash
# development aid: refinement tests + timing of one 800 x 800 pass (3 runs) + the per-layer trace
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/refine_check
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_refine.py tests/test_gpu_frames.py::test_config5_composed_small_vs_oracles -q -m gpu -x 2>&1 | tail -2 | tee $O/summary.txt
for r in 1 2 3; do timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed "s/^/round $r: /" | tee -a $O/summary.txt; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python $R/scripts/prof_refine.py 256 3 > $O/trace.log 2>&1)
python scripts/refine_layers.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 2>&1 | tail -32 | tee -a $O/summary.txt
rm -rf $O/trace
