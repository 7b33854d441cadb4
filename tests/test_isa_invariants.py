"""Build-time invariants of the hand-written kernels, checked on the gfx950 ISA hipcc emits (no GPU needed).

Performance tripwires only -- no correctness property of the kernels depends on what is checked here: the score
kernels must not spill vector registers, and their candidate records must arrive through compiler-tracked scalar
loads (no inline-asm memory access)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-Wno-unused-function", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "ransac.jl_amd", "csrc"), "-x", "hip",
                           os.path.join(ROOT, "ransac.jl_amd", "csrc", "kernels.hip"), "--cuda-device-only", "-S", "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return out.read_text().split("\n")


def _kernel_bodies(lines):
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN\S*score_groups\S*):", lines[i])
        if m:
            j = i
            while j < len(lines) and "s_endpgm" not in lines[j]:
                j += 1
            yield m.group(1), lines[i:j]
            i = j
        i += 1


def test_no_inline_asm_memory_access_in_the_score_kernels(isa):
    """Round 1 prefetched the next candidate's record with an asm pair of scalar loads whose SGPR tuples the compiler
    did not know were in flight (correct only while it neither moved nor spilled them: an ISA grep was the guard).
    The records are now read through the constant address space (rh_ld_prep_const, score_device.h): ordinary scalar
    loads the compiler tracks itself.  Nothing in the score kernels may load through inline asm any more."""
    kernels = 0
    for name, body in _kernel_bodies(isa):
        kernels += 1
        for i, line in enumerate(body):
            if "ASMSTART" in line:
                j = i + 1
                while "ASMEND" not in body[j]:
                    assert "s_load" not in body[j] and "s_waitcnt" not in body[j], (name, body[j])
                    j += 1
        assert any("s_load_dwordx" in l for l in body)      # the candidate records do arrive through scalar loads
    assert kernels >= 8


def test_score_kernels_do_not_spill_vector_registers(isa):
    """The score kernels at their natural register count never spill.  The variants capped at 64 registers (8 waves per
    SIMD: the last template argument of score_groups_all_kernel is 8) may spill a handful -- the cone body needs ~75
    registers; plane / sphere / cylinder fit -- and are measured faster all the same."""
    text = "\n".join(isa)
    blocks = re.findall(r"\.name:\s+(\S*score_groups\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(blocks) >= 8
    capped = 0
    for name, spills in blocks:
        if "score_groups_all_kernel" in name and "ELi8EE" in name:
            capped += 1
            assert int(spills) <= 12, "%s spills %s VGPRs: more than the cone body's handful" % (name, spills)
        else:
            assert int(spills) == 0, "%s spills %s VGPRs (scratch traffic in the hot loop)" % (name, spills)
    assert capped >= 2
