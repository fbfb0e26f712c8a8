#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_s6; mkdir -p $O; cd $R
timeout 900 python scripts/fp32_ray_probe.py fp32 5 65536 > $O/fp32_ray_probe.json 2> $O/probe.err; tail -3 $O/probe.err; python -c "
import json; d=json.load(open('$O/fp32_ray_probe.json')); print(d['violations'], d['top_rays']); [print(json.dumps(p, indent=0)) for p in d['probes']]"
timeout 600 bash scripts/power_clock_probe.sh > $O/power_clock_probe.txt 2>&1; tail -25 $O/power_clock_probe.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][0]); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print('train', d['train']['ms_per_step'], 'c1', d['config1']['ms_per_step'], 'refine', d['config5']['refine']['refine_ms']); print('arch', d.get('arch')); print('errors', d.get('extras_errors'))"
