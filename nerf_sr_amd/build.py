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
AB_DIR = os.path.join(os.path.dirname(HERE), "ab")     # A/B variant libraries (git-ignored *.so)
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

    ``variant`` / ``defines`` build an ablation library ``<repo>/ab/libnsr_<variant>.so`` with extra ``-D`` flags
    (development aid; select it at run time with ``NSR_LIB_PATH``; ``ab/`` travels to the GPU box, stale variant
    libraries next to the product library do not: .gpurunignore)."""
    if variant:
        os.makedirs(AB_DIR, exist_ok=True)
    LIB = os.path.join(AB_DIR, f"libnsr_{variant}.so") if variant else globals()["LIB"]
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


# ---- ISA contract of the hand-written inline asm (tests/test_isa_contract.py; ADVICE r4: run it for every variant) ----------
# 1. M0 stays under the asm's control: LDS-DMA pieces 4q+1..4q+3 reuse the M0 their group's first piece wrote, so no other
#    instruction of those kernels may write M0.
# 2. A VMEM instruction issued from inline asm never takes its SGPR base straight from the compiler: hipcc restores spilled
#    SGPRs with v_readlane, a VMEM read of a VALU-written SGPR needs five wait states, and the hazard pass cannot see through
#    inline asm: every global_load_lds / global_store inside an asm block must read a pair written by an s_mov_b64 inside the
#    SAME block.
ISA_UNITS = ["nsr_mlp_f16.hip", "nsr_train_chain.hip", "nsr_gemm_f16.hip", "nsr_wgrad_f16.hip"]
ISA_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S",
             "--cuda-device-only"]


def isa_contract(text: str):
    """(violations, n_m0_writes, n_asm_vmem) of one `hipcc -S` listing."""
    import re
    bad, n_m0, n_vmem = [], 0, 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        for line in body.split("\n"):
            code = line.split(";")[0]
            if re.search(r"\bm0\b", code):
                n_m0 += 1
                if not re.match(r"\s*s_mov_b32 m0, \w+\s*$", code):
                    bad.append(f"{name}: unexpected use of M0: {code.strip()}")
        for blk in re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", body, re.S):
            copies = set(re.findall(r"s_mov_b64 (s\[\d+:\d+\])", blk))
            for v in re.finditer(r"(?:global_load_lds_dword(?:x4)? v\d+, (s\[\d+:\d+\])|global_store_dword(?:x4)? v\d+, v(?:\d+|\[\d+:\d+\]), (s\[\d+:\d+\]))", blk):
                base = v.group(1) or v.group(2)
                n_vmem += 1
                if base not in copies:
                    bad.append(f"{name}: VMEM in inline asm reads {base} without an in-statement s_mov_b64")
    return bad, n_m0, n_vmem


def compile_listing(unit: str, defines=(), out_dir=None) -> str:
    """`hipcc -S --cuda-device-only` of one translation unit (cached by source hash + flags in the temp directory)."""
    import hashlib
    import tempfile
    key = hashlib.sha256((source_hash() + unit + " ".join(defines)).encode()).hexdigest()[:20]
    out = os.path.join(out_dir or tempfile.gettempdir(), f"nsr_isa_{key}_{unit}.s")
    if not os.path.exists(out):
        subprocess.check_call([_hipcc(), *ISA_FLAGS, *defines, os.path.join(CSRC, unit), "-o", out + ".tmp"], stderr=subprocess.DEVNULL)
        os.replace(out + ".tmp", out)
    with open(out) as f:
        return f.read()


# ---- documented build variants (development aids: never defined in the product build).  One entry per switch GROUP: the
# defines, the translation units they touch, what they are for.  tests/test_variants.py compiles every one of them on the CPU
# box (and runs the ISA contract on the listing), so a variant cannot rot silently.
VARIANTS = {
    "persistent": (["-DNSR_PERSISTENT"], ["nsr_mlp_f16.hip"],
                   "ray kernels loop over tiles, one workgroup per CU, weight ring streaming across tiles (DESIGN 3.1 round 4: complete, bit-identical, not faster)"),
    "enc_overlap": (["-DNSR_PERSISTENT", "-DNSR_ENC_OVERLAP"], ["nsr_mlp_f16.hip"],
                    "persistent + the next tile's encoding in the matrix shadow (150 pinned pieces)"),
    "timeline": (["-DNSR_ABL_TIMELINE"], ["nsr_mlp_f16.hip", "nsr_train_chain.hip"],
                 "s_memtime stamps at the phase boundaries of every tile (scripts/timeline.py)"),
    "abl_chain": (["-DNSR_ABL_NO_AMAX", "-DNSR_ABL_NO_DENSITY_MMA", "-DNSR_ABL_FWD_NO_STORE", "-DNSR_ABL_BWD_NO_STORE", "-DNSR_ABL_BWD_STORE_L2"],
                  ["nsr_mlp_f16.hip", "nsr_train_chain.hip"],
                  "what the range tracking / the density block's MFMAs / the training panel stores cost (profiles/r5_headline_experiments.json)"),
    "halo_no_multi": (["-DNSR_HALO_NO_MULTI"], ["nsr_gemm_f16.hip"],
                      "refinement pass with the 8 x 8-pixel plain layers on the staged tiles (round 6 A/B; other K order: not bit-identical)"),
    "store_policy": (["-DNSR_PANEL_STORE_POLICY=\"\"", "-DNSR_ABL_BWD_DEFAULT_STORE"], ["nsr_mlp_f16.hip", "nsr_train_chain.hip"],
                     "training panel stores with the default cache policy instead of non-temporal (A/B)"),
    "abl_ring": (["-DNSR_ABL_NO_DMA", "-DNSR_ABL_NO_BARRIER", "-DNSR_ABL_NO_DRAIN", "-DNSR_ABL_NO_CONVERT"], ["nsr_mlp_f16.hip"],
                 "round 1's measurement ladder of the weight ring (profiles/r1_f16x3_pmc.txt)"),
    "abl_fenced_barrier": (["-DNSR_ABL_FENCED_BARRIER", "-DNSR_ABL_NO_TILE_LAUNDER"], ["nsr_mlp_f16.hip"],
                           "round 4's publish-point A/B (__syncthreads instead of s_barrier)"),
    "dev_switches": (["-DNSR_DEV_SWITCHES"], ["nsr_gemm_f16.hip", "nsr_refine.hip"],
                     "environment-read A/B switches of the refinement GEMMs (NSR_GEMM_TILE / _TK / _FULLN, NSR_REFINE_SEPARATE_MAX)"),
    "abl_halo": (["-DNSR_ABL_HALO_NO_PATCH", "-DNSR_ABL_HALO_NO_BDMA", "-DNSR_ABL_HALO_NO_BARRIER", "-DNSR_ABL_HALO_NO_EPILOGUE"],
                 ["nsr_gemm_f16.hip"], "ablations of conv_halo_kernel (profiles/r4_refine_halo.txt)"),
    "halo_pairs": (["-DNSR_HALO_PAIR=0", "-DNSR_HALO_PAIR_WIDE=1", "-DNSR_HALO_GROUPED_QUARTER", "-DNSR_HALO_NO_LAST"], ["nsr_gemm_f16.hip"],
                   "round 6's A/B partners of conv_halo_kernel: one workgroup per CU for the half shape; the 256-column plain layers on "
                   "paired half tiles (measured: nothing); the grouped 128-column layer as pairs of 128 x 128 workgroups (measured: nothing)"),
    "bwd_waves4": (["-DNSR_BWD_WAVES=4"], ["nsr_train_chain.hip"],
                   "reduced-term backward chains on 4-wave workgroups (one wave per SIMD for two terms; round 6: 425 vs 371 us per pass)"),
    "gemm_alt": (["-DNSR_GEMM_NO_HALO", "-DNSR_GEMM_NO_XCD", "-DNSR_GEMM_NO_WROWS", "-DNSR_GEMM_K32_ONLY=1", "-DNSR_HALO_S2_MIN_CIN=256",
                  "-DNSR_ABL_RELU_FMAX"], ["nsr_gemm_f16.hip"], "staged-kernel-only build and the other GEMM A/B partners"),
}


def check_variant(name: str):
    """Compile the variant's translation units to ISA listings and run the ISA contract on them; returns the violations."""
    defines, units, _ = VARIANTS[name]
    bad = []
    for u in units:
        v, _, _ = isa_contract(compile_listing(u, defines))
        bad += [f"{name}/{u}: {x}" for x in v]
    return bad


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build(variant=sys.argv[i + 1], defines=[a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build(force="--force" in sys.argv))
