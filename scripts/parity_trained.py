"""Parity of the contract-grade precisions on TRAINED density fields (and on the 'sharp' synthetic stress field), at frame
scale: trains both networks on the analytic scenes of tests/trained_field.py with the HIP training step, then runs the
protocol of tests/test_gpu_trained.py on `N` rays per family and writes gpurun_out/r3_parity_trained.json (+ the trained
weights as gpurun_out/trained_<family>.npz so the oracle side can be re-examined without a GPU).
usage: parity_trained.py [N=16384] [steps=6000]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_sr_amd import ops
from nerf_sr_amd.weights import make_state_dict
from oracle import nerf_oracle as oc
from tests import trained_field as tf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
os.makedirs("gpurun_out", exist_ok=True)
rep = {"rays_per_family": N, "train_steps": STEPS, "families": {}}


def protocol(family, sd_c, sd_f, tag):
    t0 = time.time()
    blk, ref, ref64 = tf.oracle_block(family, sd_c, sd_f, N)
    entry = {"oracle_seconds": round(time.time() - t0, 1)}
    white = tf.FAMILIES[family][3]
    for prec in ("f16x3", "fp32"):
        nc, nf = ops.VanillaMLP(precision=prec).load_state_dict(sd_c), ops.VanillaMLP(precision=prec).load_state_dict(sd_f)
        hip = ops.forward_rays(nc, nf, blk.cuda(), 64, 64, white)
        torch.cuda.synchronize()
        st = tf.parity_stats(hip, ref, ref64)
        st["status_flags"] = [nc.status(), nf.status()]
        entry[prec] = st
        print(tag, family, prec, json.dumps(st), flush=True)
    # trained-scale activations: fine-pass points of every 8th ray of the block, fp64 on the CPU
    o, d, near, far = blk[::8, 0:3], blk[::8, 3:6], blk[::8, 6:7], blk[::8, 7:8]
    z_c, _ = oc.sample_coarse(o, d, near, far, 64)
    z_f, xyz = oc.resample_fine(o, d, z_c, ref["coarse_weights"][::8], 64)
    x = torch.cat([oc.posenc(xyz.reshape(-1, 3), 10), oc.posenc(d, 4).repeat_interleave(128, 0)], -1)
    entry["layer_abs_max_fine"] = tf.layer_abs_max(sd_f, x)
    entry["layer_abs_max_coarse"] = tf.layer_abs_max(sd_c, x)
    return entry


for family in ("llff", "blender"):
    t0 = time.time()
    res = tf.train_field(family, steps=STEPS, log=print)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    np.savez_compressed(f"gpurun_out/trained_{family}.npz", **{f"c.{k}": v for k, v in res["sd_coarse"].items()},
                        **{f"f.{k}": v for k, v in res["sd_fine"].items()})
    e = {"train_seconds": round(t_train, 1), "history": res["history"]}
    e["trained"] = protocol(family, res["sd_coarse"], res["sd_fine"], "trained")
    e["sharp"] = protocol(family, make_state_dict(99, field="sharp"), make_state_dict(100, field="sharp"), "sharp")
    rep["families"][family] = e
    json.dump(rep, open("gpurun_out/r3_parity_trained.json", "w"), indent=1)
print("done")
