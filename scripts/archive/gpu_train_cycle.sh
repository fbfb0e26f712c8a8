timeout 300 python scripts/chain_check.py > gpurun_out/chain_check.log 2>&1
timeout 250 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/train_tests.log 2>&1; tail -3 gpurun_out/train_tests.log
python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/train_bench.json 2>gpurun_out/train_bench.err; cut -c1-330 gpurun_out/train_bench.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python scripts/train_breakdown.py $f > gpurun_out/train_breakdown.txt 2>&1; cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) gpurun_out/train_kernel_stats.csv; head -16 gpurun_out/train_breakdown.txt
