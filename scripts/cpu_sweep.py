"""CPU baseline policy (VERDICT r3 "next" #7): time the torch-CPU oracle port of the render path on this host for a grid of
(worker threads x ATen threads per worker) over disjoint ray chunks of one frame, and record the best in
profiles/r4_cpu_sweep.json -- bench.py's ``cpu_baseline`` uses that configuration when the host matches.
usage: python scripts/cpu_sweep.py [out.json] [rays_per_worker]"""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nerf_oracle as oc
from nerf_sr_amd import cameras
from nerf_sr_amd.weights import make_state_dict


def run(workers, threads, rays, sdc, sdf):
    chunks = rays.chunk(workers)
    def one(r):
        torch.set_num_threads(threads)          # OpenMP ICV of the calling thread: every worker gets its own team
        with torch.no_grad():
            return oc.forward_rays(sdc, sdf, r, 64, 64, False)
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(one, [c[:256] for c in chunks]))         # warm-up (allocator, MKL)
        t0 = time.perf_counter()
        list(ex.map(one, chunks))
        return rays.shape[0] / (time.perf_counter() - t0)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    host = os.cpu_count() or 1
    sdc, sdf = oc.to_torch_sd(make_state_dict(99)), oc.to_torch_sd(make_state_dict(100))
    frame = oc.subpixel_ray_grid(torch.from_numpy(cameras.spiral_pose(0.4)), 378, 504, cameras.llff_focal(504), 2, True, 0.0, 1.0).reshape(-1, 8)
    mid = frame.shape[0] // 2
    grid = [(1, 8), (1, 16), (1, 32), (1, 64), (1, host), (2, 32), (4, 16), (4, 32), (8, 16), (16, 8), (16, 16), (32, 4), (32, 8),
            (64, 4), (64, 2)]
    grid = [(w, t) for w, t in dict.fromkeys(grid) if w * t <= host]
    res = []
    for w, t in grid:
        rays = frame[mid:mid + per * w].contiguous()
        v = run(w, t, rays, sdc, sdf)
        res.append({"workers": w, "threads_per_worker": t, "rays": int(rays.shape[0]), "rays_per_s": v})
        print(res[-1], flush=True)
    best = max(res, key=lambda r: r["rays_per_s"])
    rec = {"host_threads": host, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip(),
           "torch": torch.__version__, "what": "torch-CPU oracle port (fp32, MKL), eval forward_rays 64 + 128 samples, rays of BASELINE config #2's frame",
           "grid": res, "best": best}
    print(json.dumps(rec["best"]))
    if out:
        json.dump(rec, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
