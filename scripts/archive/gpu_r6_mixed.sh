mkdir -p gpurun_out/r6m
python scripts/bwd_terms_check.py > gpurun_out/r6m/bwd_terms.json 2> gpurun_out/r6m/bwd_terms.err; echo rc $?
python - <<'PY'
import json; d=json.load(open('gpurun_out/r6m/bwd_terms.json'))
for k in ('fixtures','bench_scale','trajectory'): print(k, json.dumps(d[k], indent=None)[:3000])
PY
bash scripts/gpu_r6_train_ab.sh "new:f16x3_bwd2 new:f16x3_bwdm new:f16x3_bwd1"
