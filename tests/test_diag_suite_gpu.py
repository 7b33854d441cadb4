"""The GPU tests that need the DIAG build of the library (libransac_hip_diag.so, -DRH_DIAG): A/B switches of the experiments
set through RH_* environment variables, the skeleton-only launch, the fake-RCCL hook, the rh_dbg_* audits.  A process loads
ONE variant of the library (clouds are handles into it), and the product library -- what every other test, bench.py and
smoke() load -- knows none of those switches and reads no environment variable.  So the tests marked `diag` run here, in one
child process started with RH_LIB_VARIANT=diag; its log goes to gpurun_out/pytest_diag.log."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_tests_marked_diag_pass_on_the_diag_build():
    if os.environ.get("RH_LIB_VARIANT") == "diag":
        pytest.skip("already inside the diag process")
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    log = os.path.join(out_dir, "pytest_diag.log")
    env = dict(os.environ, RH_LIB_VARIANT="diag")
    with open(log, "w") as f:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-x", "-q", "-m", "gpu and diag",
                            "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=f, stderr=subprocess.STDOUT, timeout=3000)
    text = open(log).read()
    tail = "\n".join(text.strip().splitlines()[-25:])
    assert r.returncode == 0, "diag-build tests failed (gpurun_out/pytest_diag.log):\n" + tail
    m = re.search(r"(\d+) passed", text)
    assert m and int(m.group(1)) >= 40, "too few diag tests ran:\n" + tail
    sys.stderr.write("\n[diag build] %s\n" % text.strip().splitlines()[-1])
