#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2c6
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_refine.py tests/test_gpu_frames.py -q -m gpu -k "refine or config5 or tiler" 2>&1 | tail -5 | tee $O/summary.txt
for b in 32 256; do timeout 200 python scripts/prof_refine.py $b 3 2>&1 | tail -1 | tee -a $O/summary.txt; done
NSR_LIB_PATH=$R/nerf_sr_amd/libnsr_nodma.so timeout 200 python scripts/prof_refine.py 256 3 2>&1 | tail -1 | sed 's/^/nodma: /' | tee -a $O/summary.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/scripts/prof_refine.py 256 1 > $O/traced.log 2>&1)
head -6 $O/trace/run_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
timeout 120 python scripts/time_refine.py f16x3 2>&1 | tail -3 | tee -a $O/summary.txt
