"""Build-time invariants of the hand-written score kernel, checked on the gfx950 ISA hipcc emits (no GPU needed).

Performance tripwires only -- no correctness property of the kernel depends on what is checked here: the counts-only
instantiations of score4_kernel (the timed step's launches) keep their hot loops free of scratch traffic at the 72
registers that 7 waves per SIMD allow, and the candidate records of the exact tests arrive through compiler-tracked
scalar loads (no inline-asm memory access)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(ROOT, "ransac.jl_amd"))
    import build   # the library's own flags
    out = tmp_path_factory.mktemp("isa") / "score4.s"
    flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared", "-Wall")]
    subprocess.check_call([HIPCC] + flags + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ransac.jl_amd", "csrc"), "-x", "hip",
                           os.path.join(ROOT, "ransac.jl_amd", "csrc", "score4.hip"), "--cuda-device-only", "-S", "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernels(text):
    """(R, MASK, F32, TAIL, LIST) -> {vgpr, vgpr_spill, sgpr_spill} of every score4_kernel instantiation in the metadata"""
    out = {}
    for m in re.finditer(r"\.name:\s+(\S*score4_kernelILi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])E\S*)\n((?:.*\n)*?)\s+\.wavefront_size", text):
        body = m.group(7) + text[m.end():m.end() + 400]
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))
        out[(int(m.group(2)), m.group(3) == "1", m.group(4) == "1", m.group(5) == "1", m.group(6) == "1")] = {
            "vgpr": g("vgpr_count"), "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count")}
    return out


def test_no_inline_asm_memory_access_in_the_score_kernel(isa):
    """Round 1 prefetched the next candidate's record with an asm pair of scalar loads whose SGPR tuples the compiler
    did not know were in flight.  The records are read through the constant address space (rh_ld_prep_const,
    score_device.h): ordinary scalar loads the compiler tracks itself.  Nothing in the score kernel may load through
    inline asm."""
    lines = isa.split("\n")
    for i, line in enumerate(lines):
        if "ASMSTART" in line:
            j = i + 1
            while "ASMEND" not in lines[j]:
                assert "s_load" not in lines[j] and "s_waitcnt" not in lines[j] and "global_load" not in lines[j], lines[j]
                j += 1
    assert "s_load_dwordx" in isa      # the records of the exact tests do arrive through scalar loads


def test_score_kernel_register_budget(isa):
    """Every instantiation the dispatch can launch is there, all within 72 vector registers (7 waves per SIMD).  The
    counts-only sized launches -- the timed step -- spill at most one of them; the mask-writing and the row-walking
    (open-ended windows) forms are allowed the handful the round measured (profiles/r4/experiments.txt)."""
    ks = _kernels(isa)
    for R in (2, 4, 8, 12, 16):
        for mask in (False, True):
            for f32 in (False, True):
                for lst in (False, True):   # (LIST: the rows walk their super-tile's candidate list, round 5)
                    assert (R, mask, f32, False, lst) in ks, (R, mask, f32, lst)
    for R in (4, 8):
        for f32 in (False, True):
            assert (R, False, f32, True, False) in ks
    for key, v in ks.items():
        R, mask, f32, tail, lst = key
        assert v["vgpr"] <= 72, (key, v)
        if not mask and not tail:
            assert v["vgpr_spill"] <= 2, (key, v)
        else:
            assert v["vgpr_spill"] <= 24, (key, v)
        if not lst and not tail:   # the scalar registers of the plain sized launches stay in registers
            assert v["sgpr_spill"] <= 12, (key, v)


def test_accounting_tool_finds_the_regions_of_both_instantiations(isa):
    """tools/isa_account.py cuts the kernel's ISA into regions by LLVM's loop annotations: the four per-kind batch loops, their
    point loops, the chunk visits in front of them.  Both forms of the timed launch (rows in bin order / rows walking super-tile
    lists) must parse: the round's roofline object is built on it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_account as ia
    for rows, lists in ((12, False), (12, True), (16, True), (2, False)):
        sym, text = ia.kernel_text(isa, rows, lists=lists)
        assert ("ELb1EE" in sym) == lists
        reg = ia.regions(ia.parse_blocks(text), rows)
        for kind in ("plane", "sphere", "cylinder", "cone"):
            for part in ("visit_test_%s", "pl_%s", "batch_%s", "stage_%s"):
                assert sum(len(b.ins) for b in reg[part % kind]) > 0, (rows, lists, part % kind)
        assert sum(len(b.ins) for b in reg["prologue"]) > 0
