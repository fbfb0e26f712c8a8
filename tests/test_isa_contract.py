"""Host-side check of two properties the hand-written inline asm of the f16x3 kernels (the MLP kernels, the training chain,
the refinement network's conv_halo_kernel) relies on (no GPU: hipcc -S):

1. M0 stays under the asm's control: the LDS-DMA pieces 4q+1..4q+3 reuse the M0 their group's first piece wrote, so no
   other instruction of those kernels may write M0;
2. a VMEM instruction issued from inline asm never takes its SGPR base straight from the compiler: hipcc restores spilled
   SGPRs with v_readlane, a VMEM read of a VALU-written SGPR needs five wait states, and the hazard pass cannot see
   through inline asm (round 4: a persistent-loop build faulted on exactly that).  Every global_load_lds / global_store
   inside an asm block must therefore read a pair written by an s_mov_b64 inside the SAME block.
"""
import hashlib
import os
import re
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "nerf_sr_amd", "csrc")
UNITS = ["nsr_mlp_f16.hip", "nsr_train_chain.hip", "nsr_gemm_f16.hip"]      # the last: conv_halo_kernel (round 4)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-S", "--cuda-device-only"]


def _compile(unit):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):
        if fn.endswith((".h", ".hip")):
            h.update(open(os.path.join(CSRC, fn), "rb").read())
    out = os.path.join(tempfile.gettempdir(), f"nsr_isa_{h.hexdigest()[:16]}_{unit}.s")
    if not os.path.exists(out):
        subprocess.check_call([hipcc, *FLAGS, os.path.join(CSRC, unit), "-o", out + ".tmp"], stderr=subprocess.DEVNULL)
        os.replace(out + ".tmp", out)
    return open(out).read()


@pytest.fixture(scope="module")
def isa():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(len(UNITS)) as ex:
        return dict(zip(UNITS, ex.map(_compile, UNITS)))


def _kernels(text):
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        yield m.group(1), m.group(2)


def test_m0_is_written_only_by_the_dma_statements(isa):
    n = 0
    for unit, text in isa.items():
        for name, body in _kernels(text):
            for line in body.split("\n"):
                code = line.split(";")[0]
                if re.search(r"\bm0\b", code):
                    n += 1
                    assert re.match(r"\s*s_mov_b32 m0, \w+\s*$", code), f"{unit}:{name}: unexpected use of M0: {code.strip()}"
    assert n > 100          # the kernels do stream through LDS-DMA


def test_inline_asm_vmem_reads_an_in_statement_copy_of_its_base(isa):
    n = 0
    for unit, text in isa.items():
        for name, body in _kernels(text):
            for blk in re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", body, re.S):
                copies = set(re.findall(r"s_mov_b64 (s\[\d+:\d+\])", blk))
                for m in re.finditer(r"(global_load_lds_dwordx4 v\d+, (s\[\d+:\d+\])|global_store_dword v\d+, v\d+, (s\[\d+:\d+\]))", blk):
                    base = m.group(2) or m.group(3)
                    n += 1
                    assert base in copies, f"{unit}:{name}: VMEM in inline asm reads {base} without an in-statement s_mov_b64"
    assert n > 100
