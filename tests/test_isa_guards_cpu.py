"""Guards on the generated code of the built library (disassembly of the gfx950 code objects, no GPU): properties that were found
broken by reading ISA listings in earlier rounds and would come back silently with a compiler or source change --

 * the five kernels of the headline pipeline and their n = 16384 counterparts use no scratch memory and no flat (generic address
   space) memory instructions;
 * no head / tail kernel wraps a buffer load in a waterfall loop (a descriptor that ended up in VGPRs: v_readfirstlane x 4 +
   two v_cmp_eq_u64 + s_and_saveexec per load -- round 5 found 12 of them in mulrelin_tail<13,4>, 29 in <14,8>, 8 in every ks_tail);
 * the coefficient-parallel kernels address memory through buffer instructions (no 64-bit VALU address arithmetic per access).
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "sunscreen_amd", "lib", "libhipbfv.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.fixture(scope="module")
def kernels():
    if not (os.path.exists(LIB) and os.path.exists(OBJDUMP)):
        pytest.skip("library or llvm-objdump missing")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(LIB, os.path.join(td, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=td, check=True, capture_output=True)
        for co in sorted(f for f in os.listdir(td) if f.endswith("gfx950")):
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--no-leading-addr", co], cwd=td, check=True, capture_output=True, text=True).stdout
            cur = None
            for line in text.splitlines():
                m = re.match(r"^<(_ZN6hipbfv\w+)>:$", line.strip())
                if m:
                    cur = m.group(1)
                    out[cur] = []
                elif cur and line.startswith(("\t", " ")) and line.strip():
                    out[cur].append(line.strip().split()[0])
    assert len(out) > 200, len(out)
    return out


def _pick(kernels, pattern):
    hit = {k: v for k, v in kernels.items() if re.search(pattern, k)}
    assert hit, pattern
    return hit


# the instantiations the BASELINE configurations launch: n = 8192 (every FP64 row 48-bit packed) and n = 16384 (8-byte rows; per row since r06)
HOT = (r"15mul_head_kernelILi13ELi4ELb1ELi1ELb[01]E", r"14mul_mid_kernelILi13ELb1ELb1ELb[01]E", r"20mulrelin_head_kernelILi13ELi4ELi1ELb0ELi1E",
       r"13ks_mid_kernelILi13ELb1ELi8E", r"20mulrelin_tail_kernelILi13ELi4ELi1ELb0ELb1E", r"14ks_head_kernelILi13ELi1ELb0E", r"14ks_tail_kernelILi13ELb1ELb0E",
       r"15mul_head_kernelILi14ELi8ELb1ELi0ELb0E", r"14mul_mid_kernelILi14ELb1ELb0ELb[01]E", r"20mulrelin_head_kernelILi14ELi8ELi0ELb1ELi0E",
       r"13ks_mid_kernelILi14ELb0ELi16E", r"20mulrelin_tail_kernelILi14ELi8ELi0ELb1ELb0E", r"14ks_head_kernelILi14ELi0ELb0E", r"14ks_tail_kernelILi14ELb0ELb0E",
       # r06: n = 16384 packs the multiply's rows PER ROW by default (the rows whose prime is below 2^48): head / tail <.., PACK = 2, ..> and the
       # packed middle instantiation beside the 8-byte one (its squaring form was the spilling kernel that kept this opt-in through r05)
       r"15mul_head_kernelILi14ELi8ELb1ELi2ELb0E", r"14mul_mid_kernelILi14ELb1ELb1ELb[01]E", r"20mulrelin_head_kernelILi14ELi8ELi2ELb1ELi[02]E",
       r"20mulrelin_tail_kernelILi14ELi8ELi2ELb1ELb0E", r"14ks_head_kernelILi14ELi2ELb0E", r"13ks_mid_kernelILi14ELb1ELi16ELb0E")


# The one tolerated exception: the packed 16-element key-switch middle kernel of n = 16384 (launched since r06's per-row packing of the
# key-switch rows) parks four loop-invariant dwords in scratch AROUND its digit loop: 4 stores before it, 5 loads after it, none inside
# (9 of ~9300 instructions; measured with the rest of the per-row change, HISTORY.md R6 s22).  More than that is a regression.
PARKED = {r"13ks_mid_kernelILi14ELb1ELi16ELb0E": 10}


@pytest.mark.parametrize("pattern", HOT)
def test_hot_kernels_use_no_scratch_and_no_flat_memory_instructions(kernels, pattern):
    for name, ins in _pick(kernels, pattern).items():
        flat = sorted({i for i in ins if i.startswith("flat_")})
        assert not flat, (name, flat)
        scratch = [i for i in ins if i.startswith("scratch_")]
        assert len(scratch) <= PARKED.get(pattern, 0), (name, len(scratch), sorted(set(scratch)))


def test_the_known_spilling_instantiations_are_pinned(kernels):
    """Which split kernels DO use scratch, so that a new one is noticed: the packed 16-element key-switch middle kernel of n = 16384 (four
    dwords parked around its loop: PARKED above), the integer middle kernel of n = 4096, and the integer key-switch middle kernel of n = 32768."""
    spilling = sorted(k for k, ins in kernels.items() if re.search(r"mul_|ks_|mulrelin", k) and any(i.startswith("scratch_") for i in ins))
    # r06: + the integer key-switch middle kernel of n = 32768 (one 1024-thread workgroup per CU: 128 registers per lane; measured
    # against three variants that spill less or not at all -- all slower, kernels_split.hip KS_MID_INT_EPT15)
    allowed = (r"13ks_mid_kernelILi14ELb1ELi16E", r"14mul_mid_kernelILi12ELb0ELb0ELb0E", r"17ks_mid_int_kernelILi15ELi8E")
    for k in spilling:
        assert any(re.search(a, k) for a in allowed), k


@pytest.mark.parametrize("pattern", [r"20mulrelin_tail_kernel", r"20mulrelin_head_kernel", r"14ks_tail_kernel", r"14ks_head_kernel",
                                     r"15mul_head_kernel", r"15mul_tail_kernel"])
def test_no_waterfall_loops_around_buffer_accesses(kernels, pattern):
    for name, ins in _pick(kernels, pattern).items():
        assert ins.count("v_cmp_eq_u64_e32") + ins.count("v_cmp_eq_u64_e64") <= 1, (name, ins.count("v_cmp_eq_u64_e32"))


def test_coefficient_parallel_kernels_address_memory_through_buffer_instructions(kernels):
    for pattern in (r"15mul_head_kernelILi13ELi4ELb1ELi1E", r"20mulrelin_head_kernelILi13ELi4ELi1E", r"20mulrelin_tail_kernelILi13ELi4ELi1E"):
        for name, ins in _pick(kernels, pattern).items():
            mem = [i for i in ins if i.startswith(("buffer_", "global_"))]
            assert mem and all(i.startswith("buffer_") for i in mem), (name, sorted(set(mem)))
            # the 64-bit VALU address arithmetic of the flat form (two or three instructions per access) is gone
            assert ins.count("v_lshl_add_u64") + ins.count("v_addc_co_u32_e32") <= 8, (name, ins.count("v_lshl_add_u64"), ins.count("v_addc_co_u32_e32"))
