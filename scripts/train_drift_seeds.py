"""Record (round 6): the 200-step Adam trajectory statistics of tests/test_gpu_train.py's drift test over SEVERAL seeds, for every
training arithmetic against the all-fp32 run -- how noisy is each statistic from seed to seed, and does the arithmetic of the
backward chain (3 / 2 / 1 MFMA terms) move any of them?   usage: python scripts/train_drift_seeds.py [n_seeds] > profiles/r6_train_drift_seeds.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.trained_field import adam_trajectory, trajectory_drift
from nerf_sr_amd import build

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
precs = ("f16x3_gemm", "f16x3_bwd3", "f16x3_bwd2", "f16x3_bwd1")
out = {"protocol": "tests/trained_field.py::adam_trajectory(precision, steps=200, seed=s): same start, batches and draws for every precision of a seed",
       "csrc_sha256": build.source_hash(), "seeds": {}}
for s in range(n):
    ref = adam_trajectory("fp32", steps=200, seed=s)
    out["seeds"][str(s)] = {p: trajectory_drift(adam_trajectory(p, steps=200, seed=s), ref) for p in precs}
keys = ("loss_rel_diff_max", "loss_rel_diff_first10_max", "last40_mean_rel_diff", "weights_rel_distance")
out["summary"] = {p: {k: {"min": min(out["seeds"][str(s)][p][k] for s in range(n)), "median": sorted(out["seeds"][str(s)][p][k] for s in range(n))[n // 2],
                          "max": max(out["seeds"][str(s)][p][k] for s in range(n))} for k in keys} for p in precs}
print(json.dumps(out, indent=1))
