#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_s5; mkdir -p $O; cd $R
timeout 600 python scripts/train_drift_seeds.py 6 > $O/train_drift_seeds.json 2> $O/drift.err; python -c "import json; d=json.load(open('$O/train_drift_seeds.json')); print(json.dumps(d['summary'], indent=0)); [print(s, {p: round(v['loss_rel_diff_max'],4) for p,v in x.items()}) for s,x in d['seeds'].items()]"
timeout 1500 python -m pytest tests/test_gpu_frames.py tests/test_gpu_trained.py tests/test_gpu_options.py -q --durations=15 2>&1 | tail -40 | tee $O/pytest.txt
