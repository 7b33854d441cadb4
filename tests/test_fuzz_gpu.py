"""A bounded, fixed-seed slice of the two fuzzers (tools/fuzz_score.py, tools/fuzz_e2e.py) inside the suite, so that
a green GPU run means what DESIGN.md section 5 claims from the long fuzz campaigns: random clouds, sizes around the tile
/ group boundaries, coordinate scales 1 .. 1e4, random enabled patterns, thresholds from tiny to huge, candidates from
jittered ground truth to degenerate (NaN / inf / zero axes), both score kernels, masks or counts; and whole rh_ransac
runs over random shape mixes, iteration parameters, score / sphere / sampling / octree modes and driver switches --
all compared with the oracle bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# the driver's A/B switches (diag build only: the e2e fuzzer draws them; the score / refit fuzzers set product options)
_ENV = ("RH_NO_PIPELINE", "RH_NO_FUSED_SCORE", "RH_HOST_SAMPLER", "RH_NO_CREC", "RH_NO_FUSED_SAMPLER",
        "RH_LONG_WINDOW_SETS", "RH_NO_FAST_EXTRACT", "RH_NO_OCT_CHAIN", "RH_NO_MANAGED_STORE",
        "RH_NO_OCT_TAB", "RH_OCT_CHAIN_W", "RH_OCT_ONE_WINDOW", "RH_OCT_WINDOW_ITERS")
_OPTS = ("score_path", "s4_rows", "refit_path", "st_cull")


@pytest.fixture(autouse=True)
def _restore_env():
    import ransac_jl_amd as R
    old = {k: os.environ.get(k) for k in _ENV}
    yield
    for k in _OPTS:
        R.set_option(k, None)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("seed,ncases", [(20261, 100), (20262, 100), (20263, 100)])
def test_score_fuzz_slice(seed, ncases):
    import fuzz_score
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(ncases):
        ok, desc = fuzz_score.one(case + seed % 1000, rng)
        if not ok:
            bad.append(desc)
    assert not bad, bad[:5]


@pytest.mark.parametrize("seed,ncases", [(41001, 100), (41002, 100)])
def test_score_fuzz_slice_float32(seed, ncases):
    """The same fuzzer on Float32 clouds: the culled kernel's box tests and band prefilter (binary64, margins widened for
    binary32 rounding) must never drop a pair the binary32 exact test accepts -- counts and masks against orc_f32.c."""
    import fuzz_score
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(ncases):
        ok, desc = fuzz_score.one(case + seed % 1000, rng, f32=True)
        if not ok:
            bad.append(desc)
    assert not bad, bad[:5]


@pytest.mark.diag
@pytest.mark.parametrize("seed,ncases,f32", [(61001, 60, False), (61002, 60, False), (61003, 60, True)])
def test_classifier_and_box_soundness_slice(seed, ncases, f32):
    """The bit-exact counts rest on two claims about the v4 score kernel's shortcuts (csrc/score4_device.h): the binary32
    box test only skips (candidate, group) pairs without an inlier, and the binary32 classifier's "sure" verdicts -- in
    or out, all four kinds -- agree with the exact test.  rh_dbg_cls_soundness COUNTS the contradictions over every
    (candidate, point) pair, with the kernel's own records and functions: 0 over random clouds at coordinate scales
    1 .. 1e6, eps 1e-6 .. 50, alpha 0.5 .. 120 degrees, jittered / arbitrary / degenerate / far-away-point shapes."""
    import fuzz_sound
    rng = np.random.default_rng(seed)
    bad = []
    tot = np.zeros((4, 10), dtype=np.int64)
    for case in range(ncases):
        ok, desc, out = fuzz_sound.one(case + seed % 1000, rng, f32=f32)
        tot += out
        if not ok:
            bad.append(desc)
    assert not bad, bad[:5]
    # the audit looked at something: every kind was classified, decided points and skipped pairs exist
    assert (tot[:, 3] > 1e6).all() and (tot[:, 1] > 0).all(), tot
    assert (tot[:3, 4] > 0).all() and (tot[:, 5] > 0).all(), tot
    if not f32:
        assert tot[3, 4] > 0, tot           # cones are decided in binary32 too (Float32 clouds: distance half only)


@pytest.mark.parametrize("seed,ncases,f32", [(5101, 80, False), (5102, 80, False), (5103, 80, True)])
def test_refit_fuzz_slice(seed, ncases, f32):
    """refit / invalidate_indexes! against the oracle with the culled scan (Morton order + box tests, korder.hip) forced
    on clouds of every size (and the plain scan on a quarter of the cases): NaN / inf points, random enabled patterns,
    degenerate shapes, Float32 clouds."""
    import fuzz_refit
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(ncases):
        ok, desc = fuzz_refit.one(case + seed % 1000, rng, f32=f32)
        if not ok:
            bad.append(desc)
    assert not bad, bad[:5]


@pytest.mark.diag
@pytest.mark.parametrize("seed,ncases,f32", [(31, 15, False), (32, 15, False), (33, 15, True), (34, 15, True)])
def test_e2e_fuzz_slice(seed, ncases, f32):
    """f32: ransac() on Float32 clouds (octree.jl:102-109) -- fits, scoring, liveness and refit in binary32 -- against the
    oracle's binary32 loop (oracle/orc_f32.c), through the same random modes and driver switches."""
    import fuzz_e2e
    rng = np.random.default_rng(seed)
    bad = []
    for case in range(ncases):
        ok, desc = fuzz_e2e.one(case + 100 * seed, rng, f32=f32)
        if not ok:
            bad.append(desc)
    assert not bad, bad[:3]
