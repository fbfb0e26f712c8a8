#!/bin/bash
# round 3, final GPU session: the GPU suite as the driver runs it (with the tests' printed statistics), smoke, every bench
# line, rocprofv3 kernel stats of the bench, the PMC constants (scripts/pmc_collect.py) and the training step's traffic.
# Output: gpurun_out/r3final/ (copied into profiles/ by hand afterwards).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
t0=$(date +%s)
NSR_PARITY_REPORT=$O/parity_trained_tests.json timeout 1500 python -m pytest tests -q -m gpu -s > $O/gpu_suite.log 2>&1; echo "suite rc $? in $(( $(date +%s) - t0 )) s" | tee $O/summary.txt
grep -E "passed|failed" $O/gpu_suite.log | tail -2 | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-260 | tee -a $O/summary.txt
timeout 300 python bench.py --precision fp32 --no-config4 > $O/bench_fp32.json 2>> $O/bench.err; tail -1 $O/bench_fp32.json | cut -c1-200 | tee -a $O/summary.txt
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench.err; tail -1 $O/bench_c3.json | cut -c1-200 | tee -a $O/summary.txt
timeout 400 python bench.py --config 5 --with-refine --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err; tail -1 $O/bench_c5.json | cut -c1-200 | tee -a $O/summary.txt
python -c "import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('refine:', {k: d['refine'][k] for k in ('warp_ms','refine_ms','tiles')}, d['refine']['roofline']['achieved'])" 2>&1 | tee -a $O/summary.txt
timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $O/bench_train.json 2>> $O/bench.err; tail -1 $O/bench_train.json | cut -c1-240 | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline --no-config4 > $O/bench_traced.log 2>&1)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -6 $O/kernel_stats.csv | cut -c1-220 | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/train_traced.log 2>&1)
cp $(find $O/trace_train -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv; head -8 $O/train_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
timeout 600 python scripts/pmc_collect.py $O/r3_pmc.json f16x3 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 400 bash scripts/pmc_train_traffic.sh 2>&1 | tail -2 | cut -c1-600 | tee -a $O/summary.txt
cp gpurun_out/train_traffic/FETCH_SIZE.json $O/train_FETCH_SIZE.json; cp gpurun_out/train_traffic/WRITE_SIZE.json $O/train_WRITE_SIZE.json
rm -rf $O/trace $O/trace_train
