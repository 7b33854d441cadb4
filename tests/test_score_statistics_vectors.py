"""Known-answer vectors for the score statistics of the candidate loop -- estimatescore / hypergeomdev
(/root/reference/src/confidenceintervals.jl:53-59, 71-74) and prob (src/utilities.jl:262) -- and the margins of the
extraction decisions (src/iterations.jl:114-123).  tests/golden/make_score_statistics_vectors.py derives them (exact
rational arithmetic for the rounding-free ones; Python integers and single IEEE operations for the wrapping ones);
here the oracle AND the product's host functions (rh_estimatescore / rh_prob: no GPU needed) must reproduce them
bit for bit."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import oracle as orc  # noqa: E402

WRAP, F64 = 0, 1


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(ROOT, "tests", "golden", "score_statistics_vectors.json")) as f:
        return json.load(f)


def _oracle_ci(S, P, sigma, mode):
    L = orc.lib()
    L.orc_estimatescore.restype = orc.CI
    L.orc_estimatescore.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int]
    ci = L.orc_estimatescore(S, P, sigma, mode)
    return ci.min, ci.max, ci.E


def _product_ci(S, P, sigma, mode):
    import ransac_jl_amd as R
    lo, hi, e = C.c_double(), C.c_double(), C.c_double()
    R._lib.check(R.lib().rh_estimatescore(S, P, sigma, mode, C.byref(lo), C.byref(hi), C.byref(e)))
    return lo.value, hi.value, e.value


@pytest.mark.parametrize("side", ["oracle", "product"])
def test_estimatescore_rounding_free_vectors(vec, side):
    f = _oracle_ci if side == "oracle" else _product_ci
    assert len(vec["estimatescore_exact"]) >= 12
    for v in vec["estimatescore_exact"]:
        for mode in (WRAP, F64):     # nothing wraps and nothing is rounded: both modes give the same exact numbers
            got = f(v["S1"], v["P"], v["sigma"], mode)
            assert got == (v["min"], v["max"], v["E"]), (side, mode, v, got)
    assert any(v["radicand"] < 0 for v in vec["estimatescore_exact"])       # the `sq_ < 0 ? zero` branch
    assert any(v["radicand"] == 0 for v in vec["estimatescore_exact"])      # sigma = S1: a zero-width interval


@pytest.mark.parametrize("side", ["oracle", "product"])
def test_estimatescore_int64_wrap_vectors(vec, side):
    f = _oracle_ci if side == "oracle" else _product_ci
    assert sum(v["wrapped"] for v in vec["estimatescore_wrap"]) >= 6
    for v in vec["estimatescore_wrap"]:
        got = f(v["S1"], v["P"], v["sigma"], WRAP)
        assert got == (v["min"], v["max"], v["E"]), (side, v, got)
        if v["wrapped"]:   # ... and the Float64 mode is a different function there
            assert f(v["S1"], v["P"], v["sigma"], F64) != got


def test_int64_wrap_vectors_are_what_python_integers_give(vec):
    from make_score_statistics_vectors import wrap_estimatescore, exact_estimatescore
    for v in vec["estimatescore_wrap"]:
        assert wrap_estimatescore(v["S1"], v["P"], v["sigma"]) == v
    for v in vec["estimatescore_exact"]:
        assert exact_estimatescore(v["S1"], v["P"], v["sigma"]) == v


@pytest.mark.parametrize("side", ["oracle", "product"])
def test_prob_dyadic_vectors(vec, side):
    if side == "oracle":
        L = orc.lib()
        L.orc_prob.restype = C.c_double
        L.orc_prob.argtypes = [C.c_double, C.c_int64, C.c_int64, C.c_int64]
        f = L.orc_prob
    else:
        import ransac_jl_amd as R
        f = R.lib().rh_prob
    assert len(vec["prob_exact"]) >= 6
    for v in vec["prob_exact"]:
        assert f(v["n"], v["s"], v["N"], v["k"]) == v["value"], (side, v)


def test_extraction_decisions_keep_their_distance(vec):
    """No extraction decision of the parity runs comes near prob_det: a last-ulp difference in pow() (Julia >= 1.8
    against libm) cannot move an extraction to another iteration.  The cfg1 block is re-derived here."""
    from make_score_statistics_vectors import decisions
    by = {d["workload"]: d for d in vec["decisions"]}
    assert set(by) >= {"cfg1", "cfg3"}
    for d in vec["decisions"]:
        assert d["flips_under_1ulp_pow"] == 0 and d["min_margin"] > 1e-6, d
    assert by["cfg3"]["extractions"] == 40 and by["cfg3"]["decisions"] > 500
    again = decisions("cfg1", {}, 1234)
    assert again == by["cfg1"]
