"""Host-side check of two properties the hand-written inline asm of the f16x3 kernels (the MLP kernels, the training chains,
the weight-gradient kernel, the refinement network's conv_halo_kernel) relies on (no GPU: hipcc -S): M0 stays under the asm's
control, and a VMEM instruction issued from inline asm reads an in-statement copy of its SGPR base (nerf_sr_amd/build.py,
``isa_contract``: the same check tests/test_variants.py runs on every documented build variant)."""
import shutil
import os
from concurrent.futures import ThreadPoolExecutor

import pytest

from nerf_sr_amd import build as nsr_build


@pytest.fixture(scope="module")
def isa():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(len(nsr_build.ISA_UNITS)) as ex:
        return dict(zip(nsr_build.ISA_UNITS, ex.map(nsr_build.compile_listing, nsr_build.ISA_UNITS)))


def test_m0_and_inline_asm_vmem_contract(isa):
    n_m0 = n_vmem = 0
    for unit, text in isa.items():
        bad, a, b = nsr_build.isa_contract(text)
        assert not bad, (unit, bad[:4])
        n_m0, n_vmem = n_m0 + a, n_vmem + b
    assert n_m0 > 100 and n_vmem > 100          # the kernels do stream through LDS-DMA and store from inline asm


def test_the_check_catches_violations():
    """The checker itself: a compiler-written M0 and an asm VMEM on a base the block did not copy are both reported."""
    fake = "_Z1kv:\n\ts_mov_b32 m0, s3\n\ts_add_u32 m0, m0, 4\n;;#ASMSTART\n\ts_mov_b64 s[4:5], s[8:9]\n" \
           "\tglobal_load_lds_dwordx4 v1, s[4:5] offset:0\n;;#ASMEND\n;;#ASMSTART\n\tglobal_store_dwordx4 v1, v[2:5], s[10:11] offset:0 nt\n;;#ASMEND\n.Lfunc_end0:\n"
    bad, n_m0, n_vmem = nsr_build.isa_contract(fake)
    assert n_m0 == 2 and n_vmem == 2 and len(bad) == 2, bad
