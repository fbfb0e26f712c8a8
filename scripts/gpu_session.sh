#!/bin/bash
# One GPU session through gpurun (development aid): the whole GPU suite, smoke, every bench line, rocprofv3 stats and
# the PMC passes profiles/r2_pmc.json is derived from.  Output: gpurun_out/session/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/session
mkdir -p $O
cd $R
echo "== GPU suite" | tee $O/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -s > $O/gpu_suite.log 2>&1; echo "suite rc $?" | tee -a $O/summary.txt
grep -E "^\.?\[config|passed|failed|Error" $O/gpu_suite.log | cut -c1-230 | tail -16 | tee -a $O/summary.txt
echo "== smoke" | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $O/summary.txt
echo "== bench lines" | tee -a $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300 | tee -a $O/summary.txt
timeout 400 python bench.py --config 5 --with-refine --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err; tail -1 $O/bench_c5.json | cut -c1-200 | tee -a $O/summary.txt
python -c "import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('refine:', d.get('refine'))" 2>&1 | cut -c1-600 | tee -a $O/summary.txt
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err; tail -1 $O/bench_c4.json | cut -c1-200 | tee -a $O/summary.txt
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench.err; tail -1 $O/bench_c3.json | cut -c1-200 | tee -a $O/summary.txt
timeout 300 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2>> $O/bench.err; tail -1 $O/bench_train.json | cut -c1-200 | tee -a $O/summary.txt
echo "== rocprofv3 stats of the bench" | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline > $O/bench_traced.log 2>&1)
head -8 $O/trace/run_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
echo "== PMC" | tee -a $O/summary.txt
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
C2="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
timeout 300 bash scripts/pmc.sh session/pmc_a f16x3 $C1 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 bash scripts/pmc.sh session/pmc_b f16x3 $C2 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 bash scripts/pmc.sh session/pmc_fetch f16x3 FETCH_SIZE 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 bash scripts/pmc.sh session/pmc_write f16x3 WRITE_SIZE 2>&1 | tail -2 | tee -a $O/summary.txt
