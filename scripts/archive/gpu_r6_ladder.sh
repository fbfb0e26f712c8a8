mkdir -p gpurun_out/r6final
for r in 1 2 3; do for v in f16x3_bwd3 f16x3_bwd2 f16x3_bwdm f16x3_bwd1; do
  timeout 300 python bench.py --mode train --train-precision $v --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r $v  ms_per_step %.3f' % d['ms_per_step'])"
done; done | tee gpurun_out/r6final/r6_train_terms_ab.txt
