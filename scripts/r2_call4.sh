#!/bin/bash
# round-2 GPU session 2: product f16x3 kernel (round-1 schedule + scaled weights / RNE splits) and its variants,
# the whole suite, bench lines of every configuration, rocprof stats and PMC passes of the fastest variant.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2c4
mkdir -p $O
cd $R
echo "== quick f16x3 correctness" | tee $O/summary.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "f16x3 or forward_rays or mlp" -x > $O/quick.log 2>&1; echo "quick rc $?" | tee -a $O/summary.txt
tail -3 $O/quick.log | tee -a $O/summary.txt
echo "== A/B timing" | tee -a $O/summary.txt
best=""; bestms=1000000
for v in "" c2; do
  lib=$R/nerf_sr_amd/libnsr${v:+_$v}.so
  [ -f $lib ] || continue
  line=$(NSR_LIB_PATH=$lib timeout 200 python scripts/quick_time.py f16x3 2>&1 | tail -1)
  echo "variant '${v:-product}': $line" | tee -a $O/summary.txt
  ms=$(echo "$line" | sed -n 's/.*: \([0-9.]*\) ms\/image.*/\1/p')
  if [ -n "$ms" ] && python -c "import sys; sys.exit(0 if float('$ms') < float('$bestms') else 1)"; then bestms=$ms; best=$v; fi
done
echo "fastest: '${best:-product}' $bestms ms" | tee -a $O/summary.txt
BEST_LIB=$R/nerf_sr_amd/libnsr.so   # bench and profile the PRODUCT library
echo "== full GPU suite (product library)" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -s > $O/gpu_suite.log 2>&1; echo "suite rc $?" | tee -a $O/summary.txt
grep -E "^\.?\[config|passed|failed|Error" $O/gpu_suite.log | cut -c1-260 | tail -30 | tee -a $O/summary.txt
echo "== bench lines (fastest variant)" | tee -a $O/summary.txt
export NSR_LIB_PATH=$BEST_LIB
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400 | tee -a $O/summary.txt
timeout 300 python bench.py --config 4 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench.err; tail -1 $O/bench_c4.json | cut -c1-300 | tee -a $O/summary.txt
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench.err; tail -1 $O/bench_c3.json | cut -c1-300 | tee -a $O/summary.txt
timeout 400 python bench.py --config 5 --with-refine --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err; tail -1 $O/bench_c5.json | cut -c1-300 | tee -a $O/summary.txt
python -c "import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('refine:', d.get('refine'))" 2>&1 | cut -c1-900 | tee -a $O/summary.txt
timeout 300 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2>> $O/bench.err; tail -1 $O/bench_train.json | cut -c1-300 | tee -a $O/summary.txt
timeout 300 python bench.py --precision fp32 --no-cpu-baseline > $O/bench_fp32.json 2>> $O/bench.err; tail -1 $O/bench_fp32.json | cut -c1-300 | tee -a $O/summary.txt
echo "== rocprofv3 stats of the bench" | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline > $O/bench_traced.log 2>&1)
head -8 $O/trace/run_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
echo "== PMC" | tee -a $O/summary.txt
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
C2="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
timeout 300 bash scripts/pmc.sh r2c4/pmc_a f16x3 $C1 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 bash scripts/pmc.sh r2c4/pmc_b f16x3 $C2 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 bash scripts/pmc.sh r2c4/pmc_fetch f16x3 FETCH_SIZE 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 300 bash scripts/pmc.sh r2c4/pmc_write f16x3 WRITE_SIZE 2>&1 | tail -2 | tee -a $O/summary.txt
