# training-step artefacts for profiles/: bench lines of both paths, rocprofv3 kernel stats + last-step breakdown of the chain path
mkdir -p gpurun_out/trainprof
python bench.py --mode train --steps 20 --warmup 5 > gpurun_out/trainprof/train_bench.json 2>gpurun_out/trainprof/train_bench.err
python bench.py --mode train --train-precision f16x3_gemm --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/trainprof/train_bench_gemm_path.json 2>/dev/null
python bench.py --mode train --train-precision fp32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/trainprof/train_bench_fp32.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python scripts/train_breakdown.py $f > gpurun_out/trainprof/train_breakdown.txt 2>&1; cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) gpurun_out/trainprof/train_kernel_stats.csv
for f in gpurun_out/trainprof/train_bench*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
head -20 gpurun_out/trainprof/train_breakdown.txt
