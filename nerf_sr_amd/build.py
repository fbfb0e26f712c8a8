"""Build libnsr.so (gfx950) in-tree with hipcc.  ``python -m nerf_sr_amd.build [--force]``.

One shared object, plain C ABI (include/nsr.h), no torch in the link line.  The
ray-/render-side translation units are compiled with ``-ffp-contract=off`` so the
elementwise stages keep the reference's multiply-then-add rounding; the MLP unit
is MFMA code where fused accumulation is inherent.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnsr.so")
ARCH = "gfx950"

UNITS = [
    # (source, extra flags)
    ("nsr_rays.hip", ["-ffp-contract=off"]),
    ("nsr_render.hip", ["-ffp-contract=off"]),
    ("nsr_mlp.hip", ["-ffp-contract=off"]),
    ("nsr_mlp_f16.hip", ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("nsr_mlp_h1.hip", ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("nsr_gemm.hip", ["-ffp-contract=off"]),
    ("nsr_gemm_f16.hip", ["-ffp-contract=off"]),
    ("nsr_wgrad_f16.hip", ["-ffp-contract=off"]),
    ("nsr_train.hip", ["-ffp-contract=off"]),
    ("nsr_train_chain.hip", ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]),
    ("nsr_warp.hip", ["-ffp-contract=off"]),
    ("nsr_refine.hip", ["-ffp-contract=off"]),
    ("nsr_image.hip", ["-ffp-contract=off"]),
    ("nsr_api.hip", []),
]
VARIANT_UNITS = {}   # -D flag -> (extra source, flags): experiment kernels of an ablation build (none at present)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnsr.so cannot be built")
    return exe


def _newest_source_mtime() -> float:
    m = os.path.getmtime(os.path.abspath(__file__))      # flag changes in this file rebuild too
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for fn in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, fn)))
    return m


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h; names and contents, sorted): ties a measured
    figure (profiles/*_pmc.json) to the build it was measured on.  Needs the source tree, not the toolchain."""
    import hashlib
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for fn in sorted(os.listdir(root)):
            if fn.endswith((".hip", ".h")):
                h.update(fn.encode())
                with open(os.path.join(root, fn), "rb") as f:
                    h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=()) -> str:
    """Compile every HIP unit for gfx950 and link ``nerf_sr_amd/libnsr.so``; returns its path.

    ``variant`` / ``defines`` build an ablation library ``libnsr_<variant>.so`` with extra ``-D`` flags
    (development aid; select it at run time with ``NSR_LIB_PATH``)."""
    LIB = os.path.join(HERE, f"libnsr_{variant}.so") if variant else globals()["LIB"]
    if variant:
        force = True
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source_mtime():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", variant) if variant else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", *defines]
    objs, jobs = [], []
    newest = _newest_source_mtime()
    for src, extra in UNITS + [VARIANT_UNITS[d] for d in defines if d in VARIANT_UNITS]:
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            jobs.append([hipcc, *common, *extra, "-c", srcp, "-o", obj])
        objs.append(obj)
    # the units are independent: compile them side by side (the three MFMA chain kernels take 30-40 s each)
    from concurrent.futures import ThreadPoolExecutor

    def _compile(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs) or 1, os.cpu_count() or 1))) as pool:
        list(pool.map(_compile, jobs))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build(variant=sys.argv[i + 1], defines=[a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build(force="--force" in sys.argv))
