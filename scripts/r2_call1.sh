#!/bin/bash
# round-2 GPU session 1 (through gpurun): new pair-schedule f16x3 kernel -- correctness first, then A/B vs the
# round-1 kernel (libnsr_v1.so), the bare-MFMA ceiling, PMC of both, the whole GPU suite, a bench line.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2c1
mkdir -p $O
cd $R
echo "== quick f16x3 correctness" | tee $O/summary.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "f16x3 or forward_rays or mlp" -x > $O/quick.log 2>&1; echo "quick rc $?" | tee -a $O/summary.txt
tail -5 $O/quick.log | tee -a $O/summary.txt
echo "== A/B timing" | tee -a $O/summary.txt
timeout 200 python scripts/quick_time.py f16x3 2>&1 | tail -1 | tee -a $O/summary.txt
NSR_LIB_PATH=$R/nerf_sr_amd/libnsr_v1.so timeout 200 python scripts/quick_time.py f16x3 2>&1 | tail -1 | sed 's/^/v1: /' | tee -a $O/summary.txt
echo "== mfma ceiling" | tee -a $O/summary.txt
(hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o /tmp/mfma_ceiling 2>/dev/null && timeout 200 /tmp/mfma_ceiling) 2>&1 | tee $O/mfma_ceiling.txt | tail -14 | tee -a $O/summary.txt
echo "== PMC (new kernel, then v1)" | tee -a $O/summary.txt
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
C2="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
timeout 300 bash scripts/pmc.sh r2c1/pmc_v2_a f16x3 $C1 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 300 bash scripts/pmc.sh r2c1/pmc_v2_b f16x3 $C2 2>&1 | tail -3 | tee -a $O/summary.txt
NSR_LIB_PATH=$R/nerf_sr_amd/libnsr_v1.so timeout 300 bash scripts/pmc.sh r2c1/pmc_v1_a f16x3 $C1 2>&1 | tail -3 | tee -a $O/summary.txt
echo "== kernel trace of the bench (stats)" | tee -a $O/summary.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline > $O/bench_traced.log 2>&1)
tail -2 $O/bench_traced.log | cut -c1-600 | tee -a $O/summary.txt
echo "== full GPU suite" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -s > $O/gpu_suite.log 2>&1; echo "suite rc $?" | tee -a $O/summary.txt
grep -E "^\[config|passed|failed|Error|error" $O/gpu_suite.log | tail -40 | tee -a $O/summary.txt
echo "== bench" | tee -a $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-1500 | tee -a $O/summary.txt
