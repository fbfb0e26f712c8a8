#!/bin/bash
# the counter constants of the committed kernel sources (profiles/r3_pmc.json, r3_train_traffic.json) + a bench line that quotes them
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3pmc
mkdir -p $O
cd $R
timeout 600 python scripts/pmc_collect.py $O/r3_pmc.json f16x3 2>&1 | tail -2 | tee $O/summary.txt
cp $O/r3_pmc.json profiles/r3_pmc.json
timeout 400 bash scripts/pmc_train_traffic.sh 2>&1 | tail -2 | cut -c1-200 | tee -a $O/summary.txt
cp gpurun_out/train_traffic/FETCH_SIZE.json $O/train_FETCH_SIZE.json; cp gpurun_out/train_traffic/WRITE_SIZE.json $O/train_WRITE_SIZE.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-200 | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline --no-config4 > $O/bench_traced.log 2>&1)
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -4 $O/kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
rm -rf $O/trace
