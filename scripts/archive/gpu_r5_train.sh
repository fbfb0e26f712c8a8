#!/bin/bash
# round 5, training step on 2-byte panels: probe of the transposing LDS read, the training tests, tensor-by-tensor check against
# the GEMM path and the fp64 oracle, interleaved A/B of the step time against the round-4 library, kernel statistics.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5_train
mkdir -p $O
cd $R
scripts/bin/probe_tr16 2>&1 | tee $O/probe_tr16.txt
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu 2>&1 | tail -40 | tee $O/pytest_train.txt
timeout 600 python scripts/chain_check.py llff_rand 2>&1 | tail -80 > $O/chain_check.txt
head -70 $O/chain_check.txt
for r in 1 2 3; do
  for lib in r5 r4; do
    p=nerf_sr_amd/libnsr.so; [ $lib = r4 ] && p=nerf_sr_amd/libnsr_r4.so
    [ -f $p ] || continue
    NSR_LIB_PATH=$R/$p timeout 300 python bench.py --mode train --steps 25 --warmup 5 --no-cpu-baseline 2>> $O/bench.err | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r lib=$lib  ms_per_step %.3f  losses %s' % (d['ms_per_step'], d['losses']))" | tee -a $O/ab.txt
  done
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/traced.log 2>&1)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/train_kernel_stats.csv && head -12 $O/train_kernel_stats.csv | cut -c1-200
rm -rf $O/trace
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
