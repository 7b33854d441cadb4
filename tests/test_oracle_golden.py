"""Pins the CPU oracle against the reference's own known-answer tests
(tests/golden/reference_known_answers.json, transcribed from
/root/reference/test/*.jl -- SURVEY.md 8c)."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as orc


def test_iswithinrectangle(golden):  # test/octree.jl:10-114
    g = golden["iswithinrectangle"]
    o = np.array(g["origin"], dtype=np.float64)
    w = np.array(g["widths"], dtype=np.float64)
    L = orc.lib()
    assert len(g["cases"]) == 38
    for p, exp in g["cases"]:
        p = np.array(p, dtype=np.float64)
        got = L.orc_iswithinrectangle(orc._dp(o), orc._dp(w), orc._dp(p))
        assert bool(got) == exp, p


def _grid(g):
    n, d = g["grid_n"], g["divide_by"]
    return np.array([[i / d, j / d, k / d] for i in range(n) for j in range(n) for k in range(n)], dtype=np.float64)


def test_octree_grid(golden):  # test/octree.jl:116-140
    g = golden["octree_grid"]
    ps = _grid(g)
    t = orc.Octree(ps)
    assert t.depth() == g["expected_octree_depth"]
    d, path = t.findleaf(ps[g["query_point_1based"] - 1])
    assert d == g["expected_leaf_depth"]
    # getnthcell(l, 1) == pc.octree; parents walk (octree.jl:11-22)
    assert path[0] == 0 and len(path) == d
    # root cell holds every index (octree.jl:240)
    assert np.array_equal(t.node_points(0), np.arange(1, len(ps) + 1))


def _fitparams(sphere_eps, sphere_alpha, plane_alpha=None):
    p = orc.default_params()
    p.eps[orc.SPHERE] = sphere_eps
    p.alpha[orc.SPHERE] = sphere_alpha
    if plane_alpha is not None:
        p.alpha[orc.PLANE] = plane_alpha
    p.collin_threshold = 0.2
    orc.lib().orc_params_finalize(p)
    return p


def test_dummysphere(golden):  # test/dummyspheretest.jl:14-49
    g = golden["dummysphere"]
    alfi = math.radians(g["sphere_alpha_deg"])
    base = _fitparams(g["sphere_eps"], alfi, plane_alpha=math.pi / 2)
    for s in g["sets"]:
        fs = orc.fit(orc.SPHERE, s["v"], s["n"], base)
        fp = orc.fit(orc.PLANE, s["v"], s["n"], base)
        assert (fs is not None) == s["sphere"], s["name"]
        assert (fp is not None) == s["plane"], s["name"]
        if "sphere_eps_0.01" in s:
            assert (orc.fit(orc.SPHERE, s["v"], s["n"], _fitparams(0.01, alfi)) is not None) == s["sphere_eps_0.01"]
        if "sphere_eps10_alpha_pi2" in s:
            assert (orc.fit(orc.SPHERE, s["v"], s["n"], _fitparams(10.0, math.pi / 2)) is not None) == s["sphere_eps10_alpha_pi2"]
        if "hand_derived" in s:
            h = s["hand_derived"]
            assert np.allclose(list(fs.v[0:3]), h["center"], atol=1e-12)
            assert abs(fs.v[3] - h["radius"]) < 1e-12
            assert bool(fs.outwards) == h["outwards"]


def test_default_parameters(golden):  # test/utilitytests.jl:41-65
    g = golden["default_parameters"]
    p = orc.default_params()
    names = {"plane": orc.PLANE, "sphere": orc.SPHERE, "cylinder": orc.CYLINDER, "cone": orc.CONE}
    for nm, k in names.items():
        assert p.eps[k] == g[nm]["eps"]
        assert p.alpha[k] == math.radians(g[nm]["alpha_deg"])
        assert p.cos_alpha[k] == math.cos(math.radians(g[nm]["alpha_deg"]))
    assert p.collin_threshold == g["common"]["collin_threshold"]
    assert p.parallelthrdeg == g["common"]["parallelthrdeg"]
    assert p.sphere_par == g["sphere"]["sphere_par"]
    assert p.minconeopang == math.radians(g["cone"]["minconeopang_deg"])
    it = g["iteration"]
    assert (p.drawN, p.minsubsetN, p.prob_det, p.tau, p.itermax) == (
        it["drawN"], it["minsubsetN"], it["prob_det"], it["tau"], it["itermax"])
    assert p.extract_s == orc.S_NOFMINSET and p.terminate_s == orc.S_NOFMINSET
    assert [names[n] for n in g["default_shape_order"]] == list(p.shape_types[: p.n_shape_types])


def test_pluscrossprod(golden):  # test/utilitytests.jl:116-133
    rng = np.random.default_rng(0)
    for val in golden["pluscrossprod"]["values"]:
        A = rng.random((3, 3))
        v = rng.random(3)
        v /= np.linalg.norm(v)
        T = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        expect = A + val * T
        B = np.ascontiguousarray(A.copy())
        orc.lib().orc_pluscrossprod(orc._dp(B), val, orc._dp(v))
        assert np.array_equal(B, expect)


def test_confidence_interval(golden):  # test/confidenceintervals.jl:1-26
    g = golden["confidence_interval"]
    ci = orc.CI()
    assert orc.lib().orc_confidence_interval(g["ctor"]["a"], g["ctor"]["b"], ci) == 0
    assert (ci.min, ci.max, ci.E) == (1.0, 3.0, g["ctor"]["E"])
    assert orc.lib().orc_confidence_interval(g["ctor"]["b"], g["ctor"]["a"], ci) == -1
    n = g["notsoconfident"]
    c1 = orc.lib().orc_notsoconfident(n["b"], n["a"])
    c2 = orc.lib().orc_notsoconfident(n["a"], n["b"])
    for c in (c1, c2):
        assert (c.min, c.max, c.E) == (n["min"], n["max"], n["E"])


def build_cc_case(size, patches):
    xs, ys = size
    bm = np.zeros((xs, ys), dtype=bool)
    idx = {}
    for p in patches:
        x0, x1, sx, y0, y1, sy = p["range"]
        for x in range(x0, x1 + 1, sx):
            for y in range(y0, y1 + 1, sy):
                bm[x - 1, y - 1] = True
                idx.setdefault((x - 1, y - 1), []).extend(p["idx"])
    return bm, idx


def cc_indices(bm, idx, lin):
    xs = bm.shape[0]
    out = []
    for li in lin:
        out.extend(idx[(int(li) % xs, int(li) // xs)])
    return out


@pytest.mark.parametrize("case", ["dense", "eight"])
def test_largestconncomp(golden, case):  # test/parameterspacebitmap.jl:1-55
    g = golden["largestconncomp"]
    bm, idx = build_cc_case(g["size"], g[case]["patches"])
    for conn8, key in ((False, "expected_conn4"), (True, "expected_conn8")):
        lin = orc.largestconncomp(bm, conn8=conn8)
        exp = g[case][key]
        assert cc_indices(bm, idx, lin) == exp["idx"] * exp["repeat"]


def test_findAABB(golden):  # test/utilitytests.jl:5-28
    for case in golden["findAABB"]["cases"]:
        pts = np.ascontiguousarray(case["points"], dtype=np.float64)
        d = case["dim"]
        mn, mx = np.zeros(d), np.zeros(d)
        dp = C.POINTER(C.c_double)
        orc.lib().orc_findAABB(pts.ctypes.data_as(dp), pts.shape[0], d, mn.ctypes.data_as(dp), mx.ctypes.data_as(dp))
        assert mn.tolist() == case["expected_min"] and mx.tolist() == case["expected_max"]


def test_multithreaded_score_batch_equals_sequential():
    """bench.py's all-cores steelman (OpenMP over candidates) returns the sequential oracle's counts."""
    from ransac_jl_amd import synth
    xyz, nrm, truth = synth.make_cloud(20_000, ["plane", "sphere", "cylinder", "cone"], 0.2, seed=5)
    subs = synth.make_subsets(20_000, 4, seed=5)
    oc = orc.Cloud(xyz, nrm, subs[0])
    p = orc.default_params()
    cands = synth.jittered_candidates(truth, 64, seed=1)
    arr = (orc.Shape * len(cands))()
    kmap = {"plane": 0, "sphere": 1, "cylinder": 2, "cone": 3}
    for i, (name, outw, v) in enumerate(cands):
        arr[i].kind, arr[i].outwards = kmap[name], int(outw)
        for j, x in enumerate(v):
            arr[i].v[j] = float(x)
        orc.lib().orc_shape_finalize(C.byref(arr[i]))
    assert np.array_equal(oc.score_batch(arr, p), oc.score_batch_mt(arr, p, 4))
