"""Record of the training trajectories behind tests/test_gpu_train.py::test_fp16_weight_gradient_operands_drift_like_fp32...
(GPU box): 200 Adam steps of the three training precisions from one start with identical batches and draws.
usage: python scripts/train_drift.py [steps] > profiles/r5_train_drift.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.trained_field import adam_trajectory, trajectory_drift
from nerf_sr_amd import build

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
runs = {p: adam_trajectory(p, steps=steps) for p in ("fp32", "f16x3", "f16x3_gemm")}
marks = [i for i in (0, 1, 4, 9, 24, 49, 99, 199, 399) if i < steps]
out = {"protocol": f"{steps} Adam steps (lr 5e-4) on the analytic forward-facing scene of tests/trained_field.py, 64 LR pixels x 4 sub-rays x "
                   "(64 + 128) samples per step, randomized sampling, noise_std 1; same start, batches and draws for every precision",
       "csrc_sha256": build.source_hash(),
       "fine_mse_at_steps": {p: {str(i + 1): float(r["fine"][i]) for i in marks} for p, r in runs.items()},
       "chain_f16x3_vs_fp32": trajectory_drift(runs["f16x3"], runs["fp32"]),
       "f16x3_gemm_vs_fp32": trajectory_drift(runs["f16x3_gemm"], runs["fp32"]),
       "chain_f16x3_vs_f16x3_gemm": trajectory_drift(runs["f16x3"], runs["f16x3_gemm"]),
       "cpu_study": "profiles/r4_train_fp16_wgrad_study.txt section 2 (fp64 with rounded operands vs exact: weights 0.149, last-40 mean 3.2e-4; fp32 oracle vs exact: 0.153, 8.4e-4)"}
print(json.dumps(out, indent=1))
