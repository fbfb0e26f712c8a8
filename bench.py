#!/usr/bin/env python3
"""Headline benchmark of the render hot path: rendered rays/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32] [--config 2|3|4|5] [--scaling strong|weak]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 and no torchrun environment (WORLD_SIZE unset) launches the N ranks itself -- one process per GPU
through ``torch.distributed.run`` on a free local port, like the reference's ``mp.spawn`` (train.py:154-156) -- and rank 0's
JSON line is this process's output.

One "step" = one full pass of the hot path over one synthetic frame: sub-pixel ray generation -> coarse MLP @64
samples -> compositing -> inverse-CDF resampling -> fine MLP @128 samples -> compositing -> s^2 mean.
The workload is the SAME at every N: BASELINE config #2, the one the metric is quoted on (504x378 <- 252x189, 2x
supersampling, NDC, 190,512 rays), cut into N contiguous LR-pixel blocks; every rank generates and renders its own block
and ONE all-gather of the rendered LR pixels closes the step -- strong scaling, one metric string for the whole 1/2/4/8
curve.  BASELINE config #4 (ONE 1008x756 <- 252x189 frame, 4x supersampling, 762,048 rays: the frame BASELINE shards
over 8 GPUs) is cut and timed the same way right after, in the same process, and reported as the ``config4``
sub-object.  ``--scaling weak`` renders one whole frame per rank instead; ``--config`` picks one frame geometry
explicitly.  Weights are synthetic (nerf_sr_amd.weights, "smooth" field), inputs are generated on the device: nothing
crosses PCIe inside the timed region.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects: ``roofline`` (fine-MLP
launch, algorithmic FLOPs / HIP-event duration vs the dense MFMA peak of the dtype) and ``cpu_baseline`` (the
torch-CPU oracle port timed on this host, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from nerf_sr_amd import cameras, ops  # noqa: E402
from nerf_sr_amd import dist as nsr_dist  # noqa: E402
from nerf_sr_amd.weights import FLOP_PER_POINT, make_state_dict  # noqa: E402

IMG_WH = (504, 378)
DOWNSCALE = 2
N_COARSE, N_IMPORTANCE = 64, 64
RAYS_PER_FRAME = IMG_WH[0] * IMG_WH[1]                      # 190,512
N_LR = RAYS_PER_FRAME // DOWNSCALE ** 2                     # 47,628
FLOP_PER_RAY = FLOP_PER_POINT * (N_COARSE + N_COARSE + N_IMPORTANCE)   # 227,868,672
# render configurations of BASELINE.json; #2 is the one the headline metric is quoted on
RENDER_CONFIGS = {
    2: {"name": "LLFF-like (NDC)", "img_wh": (504, 378), "downscale": 2, "ndc": True, "white_bkgd": False},
    3: {"name": "Blender-like (near/far 2/6, white background)", "img_wh": (400, 400), "downscale": 2, "ndc": False, "white_bkgd": True},
    4: {"name": "LLFF-like (NDC)", "img_wh": (1008, 756), "downscale": 4, "ndc": True, "white_bkgd": False},
    5: {"name": "Blender-like (near/far 2/6, white background), render pass", "img_wh": (800, 800), "downscale": 4, "ndc": False, "white_bkgd": True},
}
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_TFLOPS = {"fp32": 157.3, "f16x3": 2500.0, "f16": 2500.0, "bf16": 2500.0}
DTYPE_NAME = {"fp32": "f32", "f16x3": "f16x3 (fp16 MFMA, hi+lo split operands, fp32 accumulate)",
              "f16": "f16 (fp16 MFMA operands, fp32 accumulate; fast path, not a parity path)",
              "bf16": "bf16 (bf16 MFMA operands, fp32 accumulate; fast path, not a parity path)"}


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


CPU_SWEEP_FILE = os.path.join("profiles", "r4_cpu_sweep.json")


def cpu_policy():
    """(worker threads, ATen threads per worker, source) of the CPU baseline.  The torch-CPU port is GEMM-bound on small
    matrices and stops scaling at ~32 threads per evaluation, so the honest baseline on a many-core host is SEVERAL
    evaluations side by side over disjoint ray chunks (torch releases the GIL inside its operators and OpenMP gives every
    calling thread its own team).  scripts/cpu_sweep.py measured the grid once on the GPU boxes' host
    (profiles/r4_cpu_sweep.json); its best point is used when this host has the same thread count, otherwise
    min(4, host // 32) workers x min(32, host) threads."""
    host = os.cpu_count() or 1
    try:
        with open(os.path.join(REPO, CPU_SWEEP_FILE)) as f:
            rec = json.load(f)
        if int(rec["host_threads"]) == host:
            return int(rec["best"]["workers"]), int(rec["best"]["threads_per_worker"]), CPU_SWEEP_FILE
    except Exception:
        pass
    return max(1, min(4, host // 32)), min(32, host), "default policy"


def cpu_baseline(sd_c, sd_f, rays_cpu: torch.Tensor, white_bkgd: bool = False):
    """Time the oracle port on a bounded ray sample (rank 0, N=1 only): `workers` Python threads x `threads` ATen threads
    (cpu_policy) over disjoint contiguous chunks of 32,768 rays (or what the caller hands over), one untimed 256-ray warm-up
    per worker.  Returns the record, the concatenated outputs (the parity block compares against them) and the ray count."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import nerf_oracle as oc     # checker/baseline only; never on the product path
    sdc, sdf = oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f)
    host = os.cpu_count() or 1
    workers, threads, source = cpu_policy()
    n = rays_cpu.shape[0] - rays_cpu.shape[0] % (16 * workers)
    chunks = list(rays_cpu[:n].chunk(workers))

    def one(r):
        torch.set_num_threads(threads)          # OpenMP ICV of the calling thread
        with torch.no_grad():
            return oc.forward_rays(sdc, sdf, r, N_COARSE, N_IMPORTANCE, white_bkgd)
    all_threads = torch.get_num_threads()
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(one, [c[:256] for c in chunks]))      # warm-up (allocator, MKL)
        t0 = time.perf_counter()
        parts = list(ex.map(one, chunks))
        dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    out = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    return {"value": n / dt, "unit": "rays/s", "cores": workers * threads, "cores_present": host, "kind": "port", "cpu": cpu_model(),
            "workers": workers, "threads_per_worker": threads, "policy_source": source,
            "sample": f"{n} rays of the same frame (middle rows) in {workers} contiguous chunks, torch-CPU oracle (fp32, MKL), "
                      f"{dt:.1f} s, {workers} workers x {threads} ATen threads = {workers * threads} of {host} host threads"}, out, n


PMC_FILE = os.path.join("profiles", "r6_pmc.json")


def committed_pmc(precision: str):
    """Counter-derived figures of the fine-pass MLP launch from the committed PMC passes (profiles/r6_pmc.json, written by
    scripts/pmc_collect.py from separate rocprofv3 --pmc runs): HBM bytes per launch ((2 x FETCH_SIZE + WRITE_SIZE) x
    1024, the guide's gfx950 correction), matrix-pipe busy fraction, effective clock.  They are constants of the build they
    were measured on, NOT measurements of this run: the file records the sha256 of the kernel sources
    (nerf_sr_amd.build.source_hash) and the figures are dropped (None) when the sources have changed since.
    Returns (figures or {}, pmc_source record)."""
    from nerf_sr_amd import build as nsr_build
    here = nsr_build.source_hash()
    try:
        with open(os.path.join(REPO, PMC_FILE)) as f:
            allp = json.load(f)
    except Exception:
        return {}, {"file": None, "csrc_sha256": None, "this_build_sha256": here, "matches_this_build": False}
    rec = allp.get(precision, {})
    measured_on = allp.get("csrc_sha256")
    ok = bool(rec) and measured_on == here
    return (rec if ok else {}), {"file": PMC_FILE, "csrc_sha256": measured_on, "this_build_sha256": here,
                                 "matches_this_build": ok}


TRAIN_TRAFFIC_FILE = os.path.join("profiles", "r6_train_traffic.json")


def committed_train_traffic(R):
    """HBM bytes per training step from the committed PMC run (profiles/r6_train_traffic.json, measured at 2,048 rays by
    scripts/pmc_train_traffic.sh): a constant of the build it names (sha256 of the kernel sources), not a measurement of
    this run; None when the sources have changed since."""
    from nerf_sr_amd import build as nsr_build
    try:
        with open(os.path.join(REPO, TRAIN_TRAFFIC_FILE)) as f:
            d = json.load(f)
        if d.get("csrc_sha256") != nsr_build.source_hash():
            return None
        return int(d["hbm_bytes_per_step"] * (R / 2048.0))
    except Exception:
        return None


def train_cpu_baseline(sd_c, sd_f, rays_cpu, target_cpu, s2: int, draws):
    """The training oracle (torch autograd over the oracle stages, fp32) on the WHOLE batch, run the way the host is fastest at
    it (cpu_policy: several evaluations side by side over disjoint, LR-aligned ray chunks -- their gradients would be summed,
    a 2.4 MB add), after one untimed warm-up evaluation per worker; repeated until >= 5 s of timed work."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import train_oracle as to     # checker/baseline only; never on the product path
    host = os.cpu_count() or 1
    workers, threads, source = cpu_policy()
    n_lr = target_cpu.shape[0]
    workers = max(1, min(workers, n_lr // 4))
    per = -(-n_lr // workers)
    jobs = []
    for w in range(workers):
        lo, hi = w * per, min((w + 1) * per, n_lr)
        if hi > lo:
            jobs.append((rays_cpu[lo * s2:hi * s2], target_cpu[lo:hi], {k: (None if v is None else v[lo * s2:hi * s2]) for k, v in draws.items()}))

    def one(job, warm=False):
        torch.set_num_threads(threads)
        r, tg, d = job
        if warm:
            n = 4 * s2
            r, tg, d = r[:n], tg[: n // s2], {k: (None if v is None else v[:n]) for k, v in d.items()}
        to.loss_and_grads(sd_c, sd_f, r, tg, s2, N_COARSE, N_IMPORTANCE, False, noise_std=1.0, **d)
    all_threads = torch.get_num_threads()
    reps, dt = 0, 0.0
    with ThreadPoolExecutor(len(jobs)) as ex:
        list(ex.map(lambda j: one(j, True), jobs))
        t0 = time.perf_counter()
        while dt < 5.0 and reps < 50:
            list(ex.map(one, jobs))
            reps += 1
            dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    n = rays_cpu.shape[0]
    return {"value": n * reps / dt, "unit": "rays/s", "cores": len(jobs) * threads, "cores_present": host, "kind": "port",
            "cpu": cpu_model(), "workers": len(jobs), "threads_per_worker": threads, "policy_source": source,
            "sample": f"the whole {n}-ray batch x {reps} iterations in {dt:.1f} s (after one warm-up evaluation per worker): "
                      f"torch-CPU training oracle (autograd, fp32), forward + backward, {len(jobs)} workers x {threads} ATen "
                      f"threads over disjoint LR-aligned ray chunks = {len(jobs) * threads} of {host} host threads"}


CHAIN_PRECISIONS = ("f16x3", "f16x3_bwd3", "f16x3_bwd2", "f16x3_bwd1", "f16x3_bwdm")   # the chain kernels (include/nsr_train.h)


def train_bench(args, rank, local, world, steps=None, warmup=None, cpu=True, shape="downx"):
    """Training-step benchmark (not the headline).  shape "downx": scripts/train_llff_downX.sh's batch -- 512 LR pixels x 4
    sub-rays = 2,048 rays per GPU per step, 64 + 128 samples, randomized sampling, noise_std 1 -- through
    Trainer.optimize_parameters (forward, s^2-mean MSE losses, backward, one gradient all-reduce for N > 1, Adam).
    shape "vanilla": BASELINE config #1, one iteration of the vanilla `nerf` model (scripts/train_llff.sh): 2,048 11-wide
    rays (view direction in columns 8:11, models/nerf_model.py:209-213), no supersampling (s = 1), the same sample counts."""
    from nerf_sr_amd import train as nsr_train
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    dev = torch.device("cuda", local)
    vanilla = shape == "vanilla"
    s = 1 if vanilla else DOWNSCALE
    s2 = s * s
    R = args.train_rays - args.train_rays % 4
    sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    t = nsr_train.Trainer(sd_c, sd_f, randomized=True, noise_std=1.0, downscale=s, ray_chunk=R,
                          precision=args.train_precision, device=dev)
    wh = (252, 189) if vanilla else IMG_WH
    frame = ops.subpixel_rays(cameras.spiral_pose(0.4 + 0.35 * rank), wh, cameras.llff_focal(wh[0]), s,
                              True, device=dev)                       # (N_lr, s2, 8)
    torch.manual_seed(1234 + rank)
    sel = torch.randperm(frame.shape[0], device=dev)[: R // s2]
    rays = frame[sel].reshape(-1, 8).contiguous()
    if vanilla:      # the vanilla dataset's rows carry a separate (normalised) view direction (data/llff_dataset.py:337-341)
        view = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
        rays = torch.cat([rays, view], 1).contiguous()
    target = torch.rand(R // s2, 3, device=dev)
    t.set_input(rays, target)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        t.optimize_parameters()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        t.optimize_parameters()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # after the timed region: did the data-parallel protocol deliver?  Every rank started from the same weights, scaled its
    # loss by 1 / world, summed the two flat gradient buffers with ONE all-reduce each per step and ran Adam replicated
    # (Trainer.all_reduce_grads; replaces DistributedDataParallel, models/networks.py:72-86): the weights must be
    # bit-identical on all ranks, and differ from what a rank would have reached alone on its own batch.
    collective = {"backend": None, "world": world, "op": None, "calls_per_step": 0, "bytes_per_call": 0,
                  "grad_scale": 1.0, "weights_identical_on_all_ranks": None}
    if world > 1:
        flat = torch.cat([v.reshape(-1) for p in t.params for v in p.values()]).double()
        cs = torch.stack([flat.sum(), -flat.sum(), flat.abs().max(), (flat * flat).sum(), -(flat * flat).sum()])
        cs = cs.to(dev if dist.get_backend() == "nccl" else "cpu")
        lo = cs.clone()
        dist.all_reduce(cs, op=dist.ReduceOp.MAX)          # max(x) == -max(-x) on every checksum <=> all ranks hold the same weights
        same = bool(cs[0] == -cs[1]) and bool(cs[3] == -cs[4]) and bool((cs == lo).all())
        agree = torch.tensor([1 if same else 0], dtype=torch.int32, device=cs.device)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        from nerf_sr_amd.dist import _world
        collective = {"backend": dist.get_backend(), "world": world, "op": "all_reduce(sum) of the flat gradient buffer, one per network",
                      "calls_per_step": len(t.flat_grads), "bytes_per_call": int(t.flat_grads[0].numel() * 4),
                      "grad_scale": float(t.grad_scale) if t.grad_scale is not None else 1.0 / _world(t.group)[1],
                      "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None,
                      "weights_identical_on_all_ranks": bool(agree.item() == 1), "adam_steps": int(t.step)}
    res = None
    if rank == 0:
        value = R * world * steps / dt
        flop_step = 3 * FLOP_PER_RAY * R            # forward + input gradients + weight gradients, per GPU
        achieved = flop_step / (dt / steps) / 1e12
        res = {
            "metric": ("training rays/sec (64+128 samples, no SS; forward + backward + Adam)" if vanilla else
                       "training rays/sec (64+128 samples, 2x SS; forward + backward + Adam)"), "value": value,
            "unit": "rays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (forward products split-fp16 x3, fp32-grade)" if args.train_precision.startswith("f16x3") else "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE config #1: training iteration of the vanilla nerf model (scripts/train_llff.sh "
                                    f"shape): {R} 11-wide rays of a 252x189 frame per GPU per step, no supersampling, 64 coarse + "
                                    "128 fine samples, randomized sampling, noise_std 1, MSE on coarse + fine, Adam lr 5e-4"
                                    if vanilla else
                                    f"training iteration of nerf_downX (scripts/train_llff_downX.sh shape): {R // 4} LR "
                                    f"pixels x 4 sub-rays = {R} rays per GPU per step, 64 coarse + 128 fine samples, "
                                    "randomized sampling, noise_std 1, s^2-mean MSE on coarse + fine, Adam lr 5e-4")
                                   + ("" if world == 1 else f"; data parallel x{world}, one gradient all-reduce per network per step"),
                       "rays_per_step": R * world, "parallelism": f"data-parallel x{world}"},
            "roofline": {"bound": "mfma", "kernel": "whole training step (fp32-MFMA GEMMs: forward, dgrad, wgrad)",
                         "achieved": achieved, "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS["fp32"], "traffic": None,
                         "flop_per_step": flop_step,
                         "note": "algorithmic flops = 3 x 593,408 MAC x 2 per sample point x 192 points per ray"},
            "losses": [float(x) for x in t.losses.tolist()],
            "collective": collective,
        }
        if args.train_precision in CHAIN_PRECISIONS:
            # chain path: forward and input gradients run on the split-fp16 MFMA, the weight gradients on one
            # fp16 MFMA per product, and the step is bound by the 2-byte panels it moves through HBM.  Rows of 2-byte values
            # per sample point: the forward kernel writes 2,560 (ten activation panels + the two encodings), the backward
            # chain writes 2,432, the weight-gradient kernels read 5,696 (each product reads its gradient panel and its
            # input panel once); + 4 bytes of sign words per 32 pre-activations, written once and read once.
            rows = 2560 + 2432 + 5696
            gb = (2.0 * rows + 2 * 304) * R * (N_COARSE + N_COARSE + N_IMPORTANCE) / 1e9
            # TRUE algorithmic bytes of a step (what any implementation must move): rays + targets in, and per network the
            # weights read and written, the gradients written and read, Adam's two moments read and written (8 x 595,844 x 4 B)
            true_bytes = R * rays.shape[1] * 4 + (R // s2) * 12 + 2 * 8 * 595844 * 4
            traffic = committed_train_traffic(R)
            bwd_terms = {"f16x3": 12, "f16x3_bwd3": 3, "f16x3_bwd2": 2, "f16x3_bwd1": 1, "f16x3_bwdm": 12}[args.train_precision]
            res["dtype"] = ("f32 results; forward from split-fp16 x3 MFMA products (fp32-grade), input gradients (backward chain) on "
                            f"{ {12: '2 / 1'}.get(bwd_terms, bwd_terms) } MFMA term(s) per product ({ {3: 'W_hi g_hi + W_hi g_lo + W_lo g_hi', 2: 'W_hi g_hi + W_lo g_hi', 1: 'W_hi g_hi', 12: 'W_hi g_hi + W_lo g_hi on the six layers nearest the output, W_hi g_hi below'}[bwd_terms] }), "
                            "weight gradients from fp16 operands (11 bits) on one MFMA per product, fp32 accumulation")
            res["roofline"] = {"bound": "hbm", "kernel": "whole training step, chain path (mlp_f16x3_kernel TRAIN, "
                                                          "chain_bwd_h_kernel / chain_bwd_kernel, wgrad_jobs_kernel)",
                               "achieved": gb / (dt / steps), "peak": 8000.0, "unit": "GB/s",
                               "frac": gb / (dt / steps) / 8000.0, "traffic": traffic,
                               "gbytes_per_step": gb,
                               # the same step against what it MUST do rather than what this design moves: the panels are a
                               # design choice, so both axes are restated on true figures
                               "true_algorithmic_bytes_per_step": true_bytes,
                               "traffic_over_true_algorithmic_bytes": (traffic / true_bytes) if traffic else None,
                               "mfma_frac_on_true_flops": achieved / PEAK_TFLOPS["f16x3"],
                               "mfma_tflops_true": achieved,
                               "mfma_tflops_issued": achieved * (4.0 + bwd_terms) / 3.0,
                               "note": "algorithmic bytes = 21,984 B of panel + sign traffic per sample point x 192 points per ray "
                                       "(split-K partial sums, weight streams and per-ray arrays excluded; traffic = PMC-measured HBM bytes of all kernels "
                                       "of a step, " + TRAIN_TRAFFIC_FILE + ", null if the kernel sources changed since); "
                                       f"mfma_tflops_issued = 3 fp16 MFMAs per product in the forward chain, {bwd_terms} in the backward chain, 1 in the weight gradients"}
        if world == 1 and cpu and not args.no_cpu_baseline:
            draws = {k: (None if v is None else v.cpu()) for k, v in t.draw(R).items()}
            res["cpu_baseline"] = train_cpu_baseline(sd_c, sd_f, rays.cpu(), target.cpu(), s2, draws)
        else:
            res["cpu_baseline"] = None
    return res


def self_launch(n: int) -> int:
    """`bench.py --gpus N` started plainly (no torchrun environment): start the N ranks ourselves, one process per GPU, through
    torch.distributed.run on a free local port (the reference spawns its own ranks too: train.py:154-156 mp.spawn,
    utils/distributed.py:5-18).  The children inherit stdout / stderr, so rank 0's JSON line is this process's output."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main_train(args):
    rank, local, world = nsr_dist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: inconsistent torchrun environment")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    res = train_bench(args, rank, local, world)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def refine_pass(img_wh, downscale, c2w, focal, o, dev, reps: int = 3):
    """BASELINE config #5's tail on the frame the timed steps just rendered: HR depth map -> nsr_depth_warp (ray
    distance -> camera-axis depth, include/nsr_warp.h) into a reference view 15 degrees away -> 64 x 64 tiles with up
    to 8 depth-warped reference patches each -> refinement network -> stitch.  Reported separately from `value`
    (the render pass); the reference image is synthetic noise (its content does not change the work)."""
    from nerf_sr_amd import pipeline, refine, warp
    W, H = img_wh
    net = refine.MaxPoolingModel(device=dev).load_state_dict(refine.make_refine_state_dict(7))
    ref_c2w = cameras.spheric_pose(-15.0, -30.0, 4.0)
    g = torch.Generator(device="cpu").manual_seed(5)
    ref_img = (torch.rand(3, H, W, generator=g) * 2 - 1).to(dev)
    hr = ops.unflatten_reshape(o["fine_comp_rgbs"], img_wh, downscale)
    depth = ops.unflatten_reshape(o["fine_depth"].reshape(-1, 1).contiguous(), img_wh, downscale)[..., 0].contiguous()
    sr = (hr.permute(2, 0, 1) * 2 - 1).contiguous()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_warp, t_ref = [], []
    for i in range(reps + 1):
        ev[0].record()
        locs = warp.depth_warp(depth, c2w, pipeline.world_to_camera(ref_c2w), focal, "ray")
        ev[1].record()
        out = refine.refine_image(net, sr, ref_img, locs)
        ev[2].record()
        torch.cuda.synchronize()
        if i:                    # first trip = warm-up (workspace allocation)
            t_warp.append(ev[0].elapsed_time(ev[1]))
            t_ref.append(ev[1].elapsed_time(ev[2]))
    n_tiles = -(-W // 64) * -(-H // 64)
    flop = 2 * refine.refine_macs(64, 64, 8) * n_tiles
    ms = sum(t_ref) / len(t_ref)
    busy, busy_src = refine_pmc_busy()
    inside = ((locs[..., 0] >= 0) & (locs[..., 0] < W) & (locs[..., 1] >= 0) & (locs[..., 1] < H)).double().mean()
    return {"what": f"depth warp + {n_tiles} tiles of 64 x 64 with 8 reference patches through the refinement network "
                    "(split-fp16 MFMA, fp32-grade) + stitch, one 800 x 800 frame",
            "warp_ms": sum(t_warp) / len(t_warp), "refine_ms": ms, "tiles": n_tiles,
            "pixels_warped_inside_reference_view": float(inside),
            "finite": bool(torch.isfinite(out).all()),
            "roofline": {"bound": "mfma", "kernel": "refinement network (implicit-im2col split-fp16 GEMMs)",
                         "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_TFLOPS["f16x3"], "unit": "TFLOP/s",
                         "frac": flop / (ms * 1e-3) / 1e12 / PEAK_TFLOPS["f16x3"], "traffic": None,
                         "flop_per_pass": flop, "mfma_issued": 3 * flop / (ms * 1e-3) / 1e12,
                         "mfma_busy_time_weighted": busy, "pmc_source": busy_src,
                         "note": "true convolution MACs x 2; three MFMAs are issued per product (split-fp16); mfma_busy_time_weighted "
                                 "= SQ_VALU_MFMA_BUSY_CYCLES share of the pass's GEMM kernels weighted by their time, a counter "
                                 "constant (scripts/pmc_refine.py), null unless pmc_source.matches_this_build"}}


def arch_bench(spec: str, dev, fused_rays_per_s: float, n_rays: int = 32768, reps: int = 3) -> dict:
    """`--arch D,W,skip+skip`: one NON-DEFAULT network architecture (models/networks.py:124-157) through the same model
    interface.  Such a network is outside the fused kernels' layout and runs nn.Linear by nn.Linear on the fp32-MFMA GEMM
    (ops.GenericMLP: every (P, W) activation travels through HBM) on the stage-by-stage route; this figure makes that path's
    price visible (VERDICT r5 "next" #7).  Workload: `n_rays` consecutive rays from the middle of config #2's frame, eval
    mode, 64 + 128 samples, the architecture's own synthetic weights."""
    import warnings
    from nerf_sr_amd.model import NeRFDownXModel, default_options
    from nerf_sr_amd.weights import make_state_dict_arch
    D, W, skips = spec.split(",")
    arch = {"D": int(D), "W": int(W), "skips": tuple(int(x) for x in skips.split("+") if x != ""), "deg_pos": 10, "deg_dir": 4}
    frame = ops.subpixel_rays(cameras.spiral_pose(0.4), (504, 378), cameras.llff_focal(504), 2, True, device=dev).view(-1, 8)
    r0 = (frame.shape[0] - n_rays) // 2 // 4 * 4
    rays = frame[r0:r0 + n_rays].contiguous()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    secs = {}
    for prec in ("f16x3", "fp32"):        # the route's two arithmetics: split-fp16 MFMA (three terms) and fp32 MFMA GEMMs
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)        # the slow-path warning is what this figure quantifies
            m = NeRFDownXModel(default_options(precision=prec, **{**arch, "skips": list(arch["skips"])}), device=dev)
        m.load_networks(make_state_dict_arch(99, **arch), make_state_dict_arch(100, **arch)).eval()
        ms = []
        with torch.no_grad():
            for i in range(reps + 1):
                ev[0].record()
                out = m.forward_rays(rays)
                ev[1].record()
                torch.cuda.synchronize()
                if i:
                    ms.append(ev[0].elapsed_time(ev[1]))
        secs[prec] = sum(ms) / len(ms) * 1e-3
        if prec == "f16x3":
            finite = bool(torch.isfinite(out["fine_comp_rgbs"]).all())
    in_xyz, in_dir, Wd = 3 + 6 * arch["deg_pos"], 3 + 6 * arch["deg_dir"], arch["W"]
    macs = sum((in_xyz if i == 0 else (Wd + in_xyz if i in arch["skips"] else Wd)) * Wd for i in range(arch["D"]))
    macs += Wd + Wd * Wd + (Wd + in_dir) * (Wd // 2) + (Wd // 2) * 3
    t = secs["f16x3"]
    rps = n_rays / t
    tf = rps * (N_COARSE + N_COARSE + N_IMPORTANCE) * 2 * macs / 1e12
    return {"arch": arch | {"skips": list(arch["skips"])},
            "route": "ops.GenericMLP: one GEMM launch per nn.Linear (precision f16x3: split-fp16 MFMA, three terms; fp32: fp32 MFMA), stage-by-stage forward_rays",
            "value": rps, "unit": "rays/s", "precision": "f16x3", "rays": n_rays, "ms": t * 1e3, "macs_per_point": macs,
            "achieved_tflops": tf, "frac_of_fp16_mfma_peak": tf / PEAK_TFLOPS["f16x3"],
            "fp32_value": n_rays / secs["fp32"], "fp32_frac_of_fp32_mfma_peak": (n_rays / secs["fp32"]) * (N_COARSE + N_COARSE + N_IMPORTANCE) * 2 * macs / 1e12 / PEAK_TFLOPS["fp32"],
            "finite": finite,
            "fused_default_arch_rays_per_s": fused_rays_per_s,
            "slowdown_vs_fused_default_per_ray": fused_rays_per_s / rps,
            "slowdown_vs_fused_default_per_flop": (fused_rays_per_s * 2 * 593408) / (rps * 2 * macs)}


def refine_pmc_busy():
    """Time-weighted matrix-pipe busy share of the refinement pass's GEMM kernels from profiles/r6_refine_pmc.json (rocprofv3
    --pmc passes, scripts/pmc_refine.py), valid for the build whose source hash it carries."""
    from nerf_sr_amd import build as nsr_build
    here = nsr_build.source_hash()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r6_refine_pmc.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, {"file": None, "matches_this_build": False}
    ok = rec.get("csrc_sha256") == here
    src = {"file": "profiles/r6_refine_pmc.json", "csrc_sha256": rec.get("csrc_sha256"), "this_build_sha256": here, "matches_this_build": ok}
    ks = [v for k, v in rec.get("kernels", {}).items() if ("gemm" in k or "conv_halo" in k) and v.get("ms_under_pmc", 0) > 0]
    t = sum(v["ms_under_pmc"] for v in ks)
    return (sum(v["ms_under_pmc"] * v["mfma_busy"] for v in ks) / t if ok and t > 0 else None), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default=os.environ.get("NSR_PRECISION", "f16x3"), choices=list(PEAK_TFLOPS),
                    help="MLP arithmetic: f16x3 (default; split-fp16 MFMA, fp32-grade: passes the 1e-4 RGB contract) "
                         "or fp32 (fp32 MFMA); f16 / bf16 = single 16-bit operands, fast but outside the parity contract")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="render", choices=["render", "train"],
                    help="render (default): the headline metric; train: one optimize_parameters iteration per step "
                         "(SURVEY §8f N1: forward + backward + Adam, fp32) on a 2,048-ray batch per GPU")
    ap.add_argument("--train-rays", type=int, default=2048, help="rays per GPU per training step (multiple of 4)")
    ap.add_argument("--train-precision", default="f16x3", choices=["f16x3", "fp32", "f16x3_gemm", "f16x3_bwd3", "f16x3_bwd2", "f16x3_bwd1", "f16x3_bwdm"],
                    help="training step: forward products on the split-fp16 MFMA (default, fp32-grade) or everything on the fp32 MFMA")
    ap.add_argument("--n-importance", type=int, default=64,
                    help="importance samples per ray: 64 (every script of the reference: 64 coarse + 128 fine network evaluations "
                         "per ray, the metric's '64+128') or 128 (the other reading: 64 + 192; the fine pass then takes the "
                         "network-then-compositor route, sample counts other than 64 / 128 are not composited in the MLP launch)")
    ap.add_argument("--config", type=int, default=0, choices=[0] + sorted(RENDER_CONFIGS),
                    help="BASELINE.json render configuration; default: #2 (the one the metric is quoted on) at every N, with "
                         "#4 (the 1008x756 frame BASELINE shards over 8 GPUs) timed after it as the `config4` sub-object")
    ap.add_argument("--scaling", default="", choices=["", "strong", "weak"],
                    help="strong (default, every N): ONE frame cut into N contiguous LR-pixel blocks + one all-gather; "
                         "weak (N > 1) = one whole frame per rank (an N-frame batch)")
    ap.add_argument("--no-config4", action="store_true",
                    help="skip the `config4` sub-object (config #4's frame sharded over the same ranks, timed after the headline)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `train` and `config5` (+ `refine`) sub-objects of the default N = 1 line")
    ap.add_argument("--arch", default="6,192,1+3",
                    help="D,W,skips (skips joined by +): a NON-default network architecture timed on the layer-by-layer route "
                         "(ops.GenericMLP) as the `arch` sub-object of the default N = 1 line; '' skips it")
    ap.add_argument("--with-refine", action="store_true",
                    help="config #5 only: after the timed render steps run depth -> warp -> refinement network on the "
                         "rendered frame and report that pass separately (`refine` object; not part of `value`)")
    args = ap.parse_args()
    global N_IMPORTANCE, FLOP_PER_RAY
    if args.n_importance != N_IMPORTANCE:
        if args.n_importance < 1 or N_COARSE + args.n_importance > 256:
            raise SystemExit("--n-importance must keep 64 + n <= 256 samples per ray")
        N_IMPORTANCE = args.n_importance
        FLOP_PER_RAY = FLOP_PER_POINT * (N_COARSE + N_COARSE + N_IMPORTANCE)
        args.no_extras = True      # the sub-objects are defined on the reference's own sample counts
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.mode == "train":
        return main_train(args)

    rank, local, world = nsr_dist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: inconsistent torchrun environment")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    sd_c, sd_f = make_state_dict(99), make_state_dict(100)
    net_c = ops.VanillaMLP(precision=args.precision, device=dev).load_state_dict(sd_c)
    net_f = ops.VanillaMLP(precision=args.precision, device=dev).load_state_dict(sd_f)
    weak = world > 1 and args.scaling == "weak"

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def time_config(cfg_id: int, steps: int, warmup: int, gather: str = "lr") -> dict:
        """`steps` timed passes of the hot path over config `cfg_id`'s frame, this rank rendering its LR-pixel block.
        gather: what the step's ONE collective carries (N > 1): "lr" = the s^2 means (12 B per LR pixel), "hr" = the rendered
        pixels themselves, fine_comp_rgbs, 12 B per ray -- the reference's test-time deliverable (unflatten_reshape of
        out_fine_comp_rgbs_ori, models/nerf_downX_model.py:410-450); every rank then holds the (H, W, 3) frame."""
        cfg = RENDER_CONFIGS[cfg_id]
        img_wh, s, ndc, white = cfg["img_wh"], cfg["downscale"], cfg["ndc"], cfg["white_bkgd"]
        s2 = s * s
        rays_per_frame = img_wh[0] * img_wh[1]
        n_lr = rays_per_frame // s2
        # this rank's share: a contiguous LR-pixel block of THE frame (default) or a whole frame of its own (--scaling weak)
        lo, hi = (0, n_lr) if weak else nsr_dist.shard_bounds(n_lr, world)[rank]
        frame_id = rank if weak else 0
        if ndc:
            c2w, focal, nf = cameras.spiral_pose(0.4 + 0.35 * frame_id), cameras.llff_focal(img_wh[0]), (0.0, 1.0)
        else:
            c2w, focal, nf = cameras.spheric_pose(40.0 * frame_id, -30.0, 4.0), cameras.blender_focal(img_wh[0]), (2.0, 6.0)
        my_rays = (hi - lo) * s2
        ws = torch.empty(max(ops._lib.load().nsr_forward_rays_workspace_bytes_for(net_c._prec, my_rays, N_COARSE, N_IMPORTANCE), 256),
                         dtype=torch.uint8, device=dev)
        outs = {}
        events = [ops.HipEvents(4) for _ in range(steps + warmup)]
        n_gather = n_lr * world if weak else n_lr
        width = 3 * s2 if gather == "hr" else 3          # floats per LR pixel in the collective's payload
        cap0 = nsr_dist.shard_bounds(n_gather, world)[0]
        # receive buffer of the collective, allocated once: a timed step is launches + one collective, no allocation
        recv = (torch.empty((world * (cap0[1] - cap0[0]), width), dtype=torch.float32, device=dev)
                if world > 1 and dist.get_backend() == "nccl" else None)

        def step(i):
            rays = ops.subpixel_rays(c2w, img_wh, focal, s, ndc, *nf, device=dev, lr_range=(lo, hi)).view(-1, 8)
            o = ops.forward_rays(net_c, net_f, rays, N_COARSE, N_IMPORTANCE, white, workspace=ws, outs=outs,
                                 events=events[i].handles)
            if gather == "hr":   # one row per LR pixel holding its s^2 rendered rays: the gathered buffer IS the ray-major frame
                pay = o["fine_comp_rgbs"].reshape(hi - lo, width)
            else:
                pay = ops.sr_mean(o["fine_comp_rgbs"], hi - lo, s2)
            if world > 1:      # the blocks of one frame (or, weak, the frames of the batch): ONE collective per step
                pay = nsr_dist.all_gather_pixels(pay, n_gather, out=recv)
            return rays, o, pay

        for i in range(warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            rays, o, frame = step(warmup + i)
        fence()
        dt = time.perf_counter() - t0
        gathered_ok = None
        if world > 1:
            on_dev = dist.get_backend() == "nccl"
            t = torch.tensor([dt], dtype=torch.float64, device=dev if on_dev else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # outside the timed region: did the collective deliver?  Every rank must hold the same assembled image, and
            # rank r's own block must sit at its place in it
            mine = o["fine_comp_rgbs"].reshape(hi - lo, width) if gather == "hr" else ops.sr_mean(o["fine_comp_rgbs"], hi - lo, s2)
            off = rank * n_lr if weak else lo
            ok_here = bool(torch.equal(frame[off:off + (hi - lo)], mine))
            cs = torch.tensor([float(frame.double().sum()), -float(frame.double().sum()), 0.0 if ok_here else 1.0],
                              dtype=torch.float64, device=dev if on_dev else "cpu")
            dist.all_reduce(cs, op=dist.ReduceOp.MAX)          # max(x) == -max(-x)  <=>  all ranks hold the same checksum
            gathered_ok = bool(cs[0].item() == -cs[1].item() and cs[2].item() == 0.0)
        rays_per_step = rays_per_frame * world if weak else rays_per_frame
        fine_ms = [events[warmup + i].elapsed_ms(2, 3) for i in range(steps)]
        coarse_ms = [events[warmup + i].elapsed_ms(0, 1) for i in range(steps)]
        cap = cap0
        hr_ok = None
        if gather == "hr" and world > 1 and not weak:     # outside the timed region: the gathered rows ARE the HR frame
            hr_frame = ops.unflatten_reshape(frame.reshape(n_lr * s2, 3), img_wh, s)
            hr_ok = tuple(hr_frame.shape) == (img_wh[1], img_wh[0], 3) and bool(torch.isfinite(hr_frame).all())
        return {"cfg_id": cfg_id, "cfg": cfg, "img_wh": img_wh, "s": s, "white": white, "rays_per_frame": rays_per_frame,
                "gather": gather, "hr_frame_assembled": hr_ok,
                "n_lr": n_lr, "my_rays": my_rays, "rays_per_step": rays_per_step, "dt": dt, "steps": steps, "warmup": warmup,
                "value": rays_per_step * steps / dt, "ms_per_step": dt / steps * 1e3,
                "fine_ms": sum(fine_ms) / len(fine_ms), "coarse_ms": sum(coarse_ms) / len(coarse_ms),
                "rays": rays, "o": o, "c2w": c2w, "focal": focal,
                "bytes_per_rank": (cap[1] - cap[0]) * width * 4 if world > 1 else 0, "gathered_ok": gathered_ok}

    def workload(r) -> str:
        cfg, wh, s = r["cfg"], r["img_wh"], r["s"]
        how = ""
        if world > 1 and not weak:
            how = (f"; ONE frame cut into {world} contiguous LR-pixel blocks ({r['my_rays']:,} rays on rank 0), every rank "
                   "generates and renders its own block, one all-gather of "
                   + ("the rendered HR pixels (12 B per ray)" if r["gather"] == "hr" else "LR pixels") + " per step")
        elif weak:
            how = f"; {world}-frame batch, one frame per rank, one all-gather of LR pixels per step"
        return (f"BASELINE config #{r['cfg_id']}: {cfg['name']} {wh[0]}x{wh[1]} <- {wh[0] // s}x{wh[1] // s}, {s}x supersampling, "
                f"{N_COARSE} coarse + {N_COARSE + N_IMPORTANCE} fine samples/ray, {r['rays_per_frame']:,} rays per frame" + how)

    # The headline workload is the SAME at every N: config #2's frame (the one BASELINE's metric is quoted on), cut into N
    # LR-pixel blocks.  Config #4 (the frame BASELINE shards over 8 GPUs) is timed the same way in the same process and
    # reported as the `config4` sub-object.
    cfg_id = args.config or 2
    main_r = time_config(cfg_id, args.steps, args.warmup)
    c4 = None
    # The sub-objects must never cost the headline line: a failure in one of them is recorded as {"error": ...}.
    extras_err = {}

    def guarded(name, fn):
        try:
            return fn()
        except Exception as e:      # noqa: BLE001 -- reported in the line, the run goes on
            extras_err[name] = f"{type(e).__name__}: {e}"[:400]
            return None
    if not args.config and not args.no_config4 and not weak:
        # config #4's collective carries what the reference's test loop delivers: the HR frame (12 B per ray)
        c4 = guarded("config4", lambda: time_config(4, min(args.steps, 3), 1, gather="hr")) if world == 1 else \
            time_config(4, min(args.steps, 3), 1, gather="hr")     # (N > 1: every rank must take the same collectives)
    # the other measured paths of SURVEY 8f ride in the default one-GPU line as sub-objects (VERDICT r3 "next" #2), so that the
    # driver's run records them: BASELINE config #5's render pass + its refinement tail, the training step, and BASELINE
    # config #1 (the vanilla model's training iteration)
    c5 = train_res = c1_res = None
    if not args.config and not args.no_extras and world == 1:
        c5 = guarded("config5", lambda: time_config(5, 2, 1))
        # 100 timed steps behind 20 untimed ones (0.35 s in all): the sub-run follows a CPU leg during which the GPU idles, and
        # 20 + 5 steps (70 ms) read 2 % above the steady state `--mode train` measures (2.97 vs 2.91 ms on one box, round 6)
        train_res = guarded("train", lambda: train_bench(args, rank, local, world, steps=100, warmup=20, cpu=True))
        c1_res = guarded("config1", lambda: train_bench(args, rank, local, world, steps=100, warmup=20, cpu=True, shape="vanilla"))
    fast_res = None
    if train_res is not None and args.train_precision == "f16x3":
        # the stated FAST path of the training step next to the default (include/nsr_train.h: the backward chain on ONE MFMA per
        # product -- per-tensor gradient bounds and the trajectory bound hold, the whole-gradient bound of the contract does not)
        fa = argparse.Namespace(**{**vars(args), "train_precision": "f16x3_bwd1"})
        fast_res = guarded("train.fast_path", lambda: train_bench(fa, rank, local, world, steps=100, warmup=20, cpu=False))
    arch_res = None
    if args.arch and not args.config and not args.no_extras and world == 1:
        arch_res = guarded("arch", lambda: arch_bench(args.arch, dev, main_r["value"]))

    if rank == 0:
        r = main_r
        IMG_WH, DOWNSCALE, S2, white = r["img_wh"], r["s"], r["s"] ** 2, r["white"]
        RAYS_PER_FRAME, my_rays, o, rays = r["rays_per_frame"], r["my_rays"], r["o"], r["rays"]
        value, fine_avg = r["value"], r["fine_ms"]
        fine_flop = my_rays * (N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT
        achieved = fine_flop / (fine_avg * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        pmc, pmc_source = committed_pmc(args.precision)
        if not (cfg_id == 2 and world == 1):
            pmc = {}                      # the counters were collected on config #2's unsharded fine-pass launch
        mfma_per_product = 3 if args.precision == "f16x3" else 1
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 and dist.get_backend() == "nccl" else None
        except Exception:
            rccl = None
        res = {
            "metric": f"rays/sec ({N_COARSE}+{N_COARSE + N_IMPORTANCE} samples, {DOWNSCALE}x SS)", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": workload(r),
                       "rays_per_step": r["rays_per_step"], "n_coarse": N_COARSE, "n_importance": N_IMPORTANCE,
                       "precision": args.precision,
                       "parallelism": f"frame per rank x{world}" if weak else f"LR-pixel blocks of one frame x{world}"},
            "collective": {"backend": dist.get_backend() if world > 1 else None, "world": world, "rccl_version": rccl,
                           "op": "all_gather_into_tensor" if world > 1 else None, "calls_per_step": 1 if world > 1 else 0,
                           "payload": ("fine_comp_rgbs of every ray (HR frame)" if r["gather"] == "hr" else "s^2-mean RGB per LR pixel") if world > 1 else None,
                           "bytes_per_rank": r["bytes_per_rank"], "result_identical_on_all_ranks": r["gathered_ok"]},
            # what a step spends outside the two MLP launches (per-ray kernels, ray generation, the collective, launch gaps and
            # host-side enqueue): the overhead strong scaling has to keep small as the per-rank share shrinks
            "non_mlp_ms_per_step": r["ms_per_step"] - r["fine_ms"] - r["coarse_ms"],
            "achieved_tflops_whole_path": value * FLOP_PER_RAY / 1e12,
            "roofline": {"bound": "mfma", "kernel": f"mlp kernel, fine pass ({my_rays:,} rays x {N_COARSE + N_IMPORTANCE} samples, rank 0)",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc.get("hbm_bytes_per_launch"),
                         "traffic_unit": f"HBM bytes/launch (PMC constants of the build named in pmc_source, {PMC_FILE}); algorithmic = "
                                         "SURVEY 8d: 32 B in + 40 B out per ray (+ 768 B/ray when both weights arrays are written)",
                         "algorithmic_bytes_per_launch": my_rays * (32 + 40),
                         "mfma_issued": achieved * mfma_per_product,
                         "mfma_issued_frac": achieved * mfma_per_product / peak,
                         "mfma_busy": pmc.get("mfma_busy"), "effective_clock_ghz": pmc.get("effective_clock_ghz"),
                         "pmc_source": pmc_source,
                         "launch_ms": fine_avg, "coarse_launch_ms": r["coarse_ms"],
                         "flop_per_launch": fine_flop,
                         "note": ("achieved counts ALGORITHMIC flops (2 x 593,408 MAC per point) over the HIP-event duration of "
                                  "this run's launches; f16x3 issues 3 MFMAs per product, so its matrix pipe is busy for 3x this "
                                  "figure (mfma_issued); traffic / mfma_busy / effective_clock_ghz are rocprofv3 counter "
                                  "constants (config #2, one GPU), null unless pmc_source.matches_this_build"
                                  if args.precision == "f16x3" else
                                  "algorithmic flops, exact fp32 MFMA" if args.precision == "fp32" else
                                  "algorithmic flops, one 16-bit MFMA per product; fast path outside the 1e-4 "
                                  "RGB contract (see the parity block)")},
            # sticky numerics status words of both networks after every frame of this run (include/nsr.h; 0 = clean)
            "numerics_status": [net_c.status(), net_f.status()],
        }
        if c4 is not None:
            f4 = c4["my_rays"] * (N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT
            res["config4"] = {"metric": f"rays/sec ({N_COARSE}+{N_COARSE + N_IMPORTANCE} samples, 4x SS)", "value": c4["value"], "unit": "rays/s",
                              "ms_per_step": c4["ms_per_step"], "steps": c4["steps"], "warmup": c4["warmup"],
                              "scaling": "strong", "workload": workload(c4), "rays_per_step": c4["rays_per_step"],
                              "fine_launch_ms": c4["fine_ms"], "coarse_launch_ms": c4["coarse_ms"],
                              "roofline_frac": f4 / (c4["fine_ms"] * 1e-3) / 1e12 / peak,
                              "non_mlp_ms_per_step": c4["ms_per_step"] - c4["fine_ms"] - c4["coarse_ms"],
                              "collective_payload": "fine_comp_rgbs of every ray: the HR frame, 12 B per ray (nerf_downX_model.py:410-450)",
                              "bytes_per_rank": c4["bytes_per_rank"], "result_identical_on_all_ranks": c4["gathered_ok"],
                              "hr_frame_assembled": c4["hr_frame_assembled"]}
        if c5 is not None:
            f5 = c5["my_rays"] * (N_COARSE + N_IMPORTANCE) * FLOP_PER_POINT
            res["config5"] = {"metric": "rays/sec (64+128 samples, 4x SS)", "value": c5["value"], "unit": "rays/s",
                              "ms_per_step": c5["ms_per_step"], "steps": c5["steps"], "warmup": c5["warmup"],
                              "workload": workload(c5), "rays_per_step": c5["rays_per_step"],
                              "fine_launch_ms": c5["fine_ms"], "coarse_launch_ms": c5["coarse_ms"],
                              "roofline_frac": f5 / (c5["fine_ms"] * 1e-3) / 1e12 / peak,
                              "refine": guarded("config5.refine", lambda: refine_pass(c5["img_wh"], c5["s"], c5["c2w"], c5["focal"], c5["o"], dev, reps=2))}
        keys = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "losses", "cpu_baseline")
        if train_res is not None:
            res["train"] = {k: train_res[k] for k in keys}
        if fast_res is not None and "train" in res:
            res["train"]["fast_path"] = {"train_precision": "f16x3_bwd1", "ms_per_step": fast_res["ms_per_step"], "value": fast_res["value"],
                                         "unit": "rays/s", "losses": fast_res["losses"],
                                         "note": "NOT the contract-grade default: backward chain on one fp16 MFMA per product (W_hi g_hi); every gradient "
                                                 "tensor within 2e-3 of its norm of the fp64 oracle (measured 6.6e-4), 200-step Adam trajectory like any "
                                                 "fp32-grade run, whole gradient 3.1e-4 from the fp32-gradient path (the default: 1.0e-4, bound 2e-4)"}
        if c1_res is not None:
            res["config1"] = {k: c1_res[k] for k in keys}
        if arch_res is not None:
            res["arch"] = arch_res
        for name, err in extras_err.items():
            top, _, sub = name.partition(".")
            if sub and isinstance(res.get(top), dict):
                res[top][sub] = {"error": err}
            else:
                res[name] = {"error": err}
        if world == 1 and not args.no_cpu_baseline:
            mid = (RAYS_PER_FRAME // 2) - (RAYS_PER_FRAME // 2) % S2
            base, ref, n = cpu_baseline(sd_c, sd_f, rays[mid:mid + 32768].cpu(), white_bkgd=white)
            res["cpu_baseline"] = base
            from oracle import nerf_oracle as oc
            got = o["fine_comp_rgbs"][mid:mid + n].cpu()
            d = (got - ref["fine_comp_rgbs"]).abs().max(-1)[0]
            # the contract as the -m gpu suite asserts it (tests/test_gpu_frames.py): per ray |dRGB| <= max(1e-4, 2 x the
            # oracle's own fp32-vs-fp64 gap on that ray).  The fp64 pass runs on the first 8,192 rays of the sample.
            n64 = min(n, 8192)
            all_threads = torch.get_num_threads()
            torch.set_num_threads(min(64, all_threads))
            with torch.no_grad():
                ref64 = oc.forward_rays(oc.to_torch_sd(sd_c, torch.float64), oc.to_torch_sd(sd_f, torch.float64),
                                        rays[mid:mid + n64].cpu().double(), N_COARSE, N_IMPORTANCE, white)
            torch.set_num_threads(all_threads)
            gap = (ref["fine_comp_rgbs"][:n64].double() - ref64["fine_comp_rgbs"]).abs().max(-1)[0]
            d64 = d[:n64].double()
            bound = torch.clamp_min(2.0 * gap, 1e-4)
            res["parity"] = {
                "max_abs_rgb_vs_oracle": float(d.max()),
                "rays_over_1e-4": int((d > 1e-4).sum()), "p999_abs_rgb": float(torch.quantile(d, 0.999)),
                # the restated contract, on the rays that also have an fp64 oracle evaluation
                "rays_with_fp64_oracle": n64,
                "rays_bounded_by_2x_oracle_gap": int((2.0 * gap > 1e-4).sum()),
                "rays_over_max_1e-4_or_2x_oracle_gap": int((d64 > bound).sum()),
                "rays_over_1e-4_with_oracle_gap_le_1e-4": int(((d64 > 1e-4) & (gap <= 1e-4)).sum()),
                "oracle_fp32_vs_fp64_gap_max": float(gap.max()),
                "lr_max_abs_rgb_vs_oracle": float((ops.sr_mean(o["fine_comp_rgbs"][mid:mid + n].contiguous(), n // S2, S2).cpu()
                                                   - oc.sr_mean(ref["fine_comp_rgbs"], n // S2, S2)).abs().max()),
                "psnr_build_vs_oracle_db": oc.psnr(got, ref["fine_comp_rgbs"]),
                "psnr_delta_db_vs_common_target": abs(oc.psnr(got, ref["coarse_comp_rgbs"]) -
                                                      oc.psnr(ref["fine_comp_rgbs"], ref["coarse_comp_rgbs"])),
                "rays_checked": n}
        else:
            res["cpu_baseline"] = None
        if args.with_refine:
            res["refine"] = refine_pass(IMG_WH, DOWNSCALE, r["c2w"], r["focal"], o, dev) if (cfg_id == 5 and world == 1) else None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
