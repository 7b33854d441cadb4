"""The unpinned part of the oracle -- StaticArrays' evaluation order of dot / norm / normalize / cross -- under
other plausible readings (oracle/ransac_oracle.c header: fma, div, scaled, pairwise, libm): how many inlier
decisions move.  tests/golden/rounding_sensitivity.json holds the committed measurement for cfg1, cfg2, cfg3 and
a cone cloud (generator: tests/golden/make_rounding_sensitivity.py); here the cfg1 block is re-derived on every
CPU run and must equal the file, and the file as a whole must say what DESIGN.md section 5 quotes from it."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_rounding_sensitivity as mrs  # noqa: E402
from oracle import oracle as orc  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "rounding_sensitivity.json")))


def test_variants_really_change_the_arithmetic():
    """Each variant must differ from the default reading in the low bits of the compared quantities (otherwise a flip
    count of zero would say nothing), and only there."""
    import ctypes as C
    rng = np.random.default_rng(5)
    dp = C.POINTER(C.c_double)
    shapes = [(orc.PLANE, [1, 2, 3, .3, .4, .85]), (orc.SPHERE, [10, 20, 30, 7.]), (orc.CYLINDER, [.6, .64, .48, 10, 20, 30, 5.]),
              (orc.CONE, [10, 20, 30, .6, .64, .48, 0.7])]
    pts = rng.uniform(0, 50, (500, 3))
    nrm = rng.normal(size=(500, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    changed = {v: 0 for v in (1, 2, 3, 4)}
    for kind, v in shapes:
        s0 = orc.make_shape(kind, True, v)
        base = np.zeros((500, 2))
        for i in range(500):
            orc.lib().orc_compat_values(C.byref(s0), pts[i].ctypes.data_as(dp), nrm[i].ctypes.data_as(dp), base[i].ctypes.data_as(dp))
        for var in changed:
            Lv = orc.variant_lib(var)
            out = np.zeros(2)
            for i in range(500):
                Lv.orc_compat_values(C.byref(s0), pts[i].ctypes.data_as(dp), nrm[i].ctypes.data_as(dp), out.ctypes.data_as(dp))
                assert np.allclose(out, base[i], rtol=1e-12, atol=1e-12)       # same quantity ...
                changed[var] += int((out != base[i]).any())                      # ... other rounding
    assert all(n > 200 for n in changed.values()), changed
    # variant 5 only swaps the cone's trig: <= 1 ulp apart from the oracle's own
    s = orc.make_shape(orc.CONE, True, [0, 0, 0, 0, 0, 1, 0.0])
    L5 = orc.variant_lib(5)
    nd = 0
    for op in rng.uniform(0.03, 3.1, 2000):
        a = orc.make_shape(orc.CONE, True, [0, 0, 0, 0, 0, 1, float(op)])
        b = orc.Shape.from_buffer_copy(bytes(a))
        L5.orc_shape_finalize(C.byref(b))
        for k in (7, 8):
            ulp = abs(int(np.float64(a.v[k]).view(np.int64)) - int(np.float64(b.v[k]).view(np.int64)))
            assert ulp <= 1
            nd += ulp
    assert nd > 0   # libm and the fdlibm restatement do differ in the last place now and then


def test_cfg1_flip_counts_match_the_committed_measurement():
    got = mrs.measure("cfg1", nthreads=min(8, os.cpu_count() or 1), e2e=True)
    want = GOLD["cfg1"]
    assert got["inliers_default"] == want["inliers_default"]
    assert got["near_threshold"]["distance_side_vs_eps"] == want["near_threshold"]["distance_side_vs_eps"]
    assert got["near_threshold"]["angle_side_vs_cos_alpha"] == want["near_threshold"]["angle_side_vs_cos_alpha"]
    assert set(got["variants"]) == set(want["variants"]) == {"fma", "div", "scaled", "pairwise", "libm"}
    for name, row in got["variants"].items():
        for k, v in row.items():
            if k == "e2e_max_rel_parameter_diff":
                assert v < 1e-12 and want["variants"][name][k] < 1e-12
            else:
                assert v == want["variants"][name][k], (name, k, v)


def test_committed_measurement_says_zero_flips():
    """What DESIGN.md quotes: no variant changes a single score-mask bit, refit index or extracted index on any config,
    and no test of any batch lies within 1e-10 of its threshold (rounding differences are ~1e-13 at most)."""
    for cfg in ("cfg1", "cfg2", "cfg3", "cones"):
        blk = GOLD[cfg]
        assert blk["candidates"] == 4096
        for name, row in blk["variants"].items():
            assert row["score_mask_bits_flipped"] == 0 and row["candidates_with_another_count"] == 0, (cfg, name)
            assert row["refit_indices_flipped"] == 0, (cfg, name)
        e = blk["edges"]
        k = e.index(1e-10)
        assert blk["near_threshold"]["distance_side_vs_eps"][k] == 0
        assert blk["near_threshold"]["angle_side_vs_cos_alpha"][k] == 0
    assert GOLD["cfg3"]["tests"] == 4096 * 312500 and GOLD["cfg3"]["refit_scans"] == 40
    assert "libm" in GOLD["cones"]["variants"] and "cone" in GOLD["cones"]["kinds"]
    assert GOLD["cfg1"]["variants"]["fma"]["e2e_indices_flipped"] == 0
