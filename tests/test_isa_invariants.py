"""Build-time invariants of the hand-written kernels, checked on the gfx950 ISA hipcc emits (no GPU needed).

The culled score kernels prefetch the next candidate's 96-byte record with an asm pair: `sprefetch_issue`
starts two scalar loads whose SGPR tuples are written asynchronously, `sprefetch_wait` is the matching
s_waitcnt.  That is only correct while the compiler neither moves nor spills those SGPRs in between (it does
not know they are still in flight).  This test reads the ISA and fails if any instruction between an issue
and the next wait touches the tuples, and records the spill counts of the score kernels."""
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


def _regs(line):
    found = {int(r) for r in re.findall(r"\bs(\d+)\b", line)}
    for a, b in re.findall(r"s\[(\d+):(\d+)\]", line):
        found |= set(range(int(a), int(b) + 1))
    return found


def test_scalar_prefetch_registers_are_untouched_between_issue_and_wait(isa):
    kernels = issues = 0
    for name, body in _kernel_bodies(isa):
        kernels += 1
        i = 0
        while i < len(body):
            m = re.search(r"s_load_dwordx16 s\[(\d+):(\d+)\]", body[i])
            if m and "ASMSTART" in body[i - 1]:
                m2 = re.search(r"s_load_dwordx8 s\[(\d+):(\d+)\]", body[i + 1])
                assert m2, "sprefetch_issue is a dwordx16 + dwordx8 pair"
                live = set(range(int(m.group(1)), int(m.group(2)) + 1)) | set(range(int(m2.group(1)), int(m2.group(2)) + 1))
                issues += 1
                j = i + 3   # past the pair and ASMEND
                while j < len(body) - 1:
                    if "ASMSTART" in body[j] and ("s_waitcnt lgkmcnt(0)" in body[j + 1] or "s_load_dwordx16" in body[j + 1]):
                        break
                    touched = _regs(body[j].split(";")[0]) & live
                    assert not touched, "%s: line %d touches in-flight prefetch registers %s: %s" % (
                        name, j, sorted(touched), body[j].strip())
                    j += 1
            i += 1
    assert kernels >= 8 and issues >= 16     # per-kind and merged variants, two issue sites each


def test_score_kernels_do_not_spill_vector_registers(isa):
    text = "\n".join(isa)
    blocks = re.findall(r"\.name:\s+(\S*score_groups\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(blocks) >= 8
    for name, spills in blocks:
        assert int(spills) == 0, "%s spills %s VGPRs (scratch traffic in the hot loop)" % (name, spills)
