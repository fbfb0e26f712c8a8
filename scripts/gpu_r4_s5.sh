#!/bin/bash
# round-4 session 5: persistent tile loop -- smoke steps, bit-identity, A/B against the round-3 / gap-aware / one-tile-per-workgroup builds, timeline, tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4s5; mkdir -p $O; cd $R
timeout 300 python scripts/debug_steps.py f16x3 > $O/dbg.log 2>&1 || { tail -5 $O/dbg.log; exit 1; }
tail -1 $O/dbg.log
bash scripts/gpu_ab_tl.sh r4s5 5 r3 gap np new
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_status.py tests/test_gpu_frames.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
