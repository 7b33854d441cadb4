"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the
header declares, the host-side plugin pieces (fit, estimatescore, prob, rng, parameters)
agree bit for bit with the oracle, and device calls fail loudly without a GPU."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import ransac_jl_amd as R
from oracle import oracle as orc
from ransac_jl_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ransac_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = R.lib()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libransac_hip.so does not export %s" % s
        assert s in L.SIGNATURES, "python binding has no signature for %s" % s
    assert sorted(L.SIGNATURES) == syms
    import re
    hdr = open(os.path.join(ROOT, "include", "ransac_hip.h")).read()
    assert lib.rh_version() == int(re.search(r"#define\s+RH_VERSION\s+(\d+)", hdr).group(1))


def test_struct_layouts_match_oracle():
    assert C.sizeof(L.Shape) == C.sizeof(orc.Shape) == 88
    assert C.sizeof(L.Params) == C.sizeof(orc.Params)
    for (n1, t1), (n2, t2) in zip(L.Params._fields_, orc.Params._fields_):
        assert n1 == n2 and C.sizeof(t1) == C.sizeof(t2)
        assert getattr(L.Params, n1).offset == getattr(orc.Params, n2).offset
    p, q = L.Params(), orc.Params()
    R.lib().rh_default_params(C.byref(p))
    orc.lib().orc_default_params(C.byref(q))
    assert bytes(p) == bytes(q)


def test_default_parameters_match_reference(golden):  # test/utilitytests.jl:41-114
    g = golden["default_parameters"]
    dp = R.DEFAULT_PARAMETERS
    assert dp["common"] == g["common"]
    for nm in ("plane", "sphere", "cylinder", "cone"):
        assert dp[nm]["ϵ"] == g[nm]["eps"] and dp[nm]["α"] == math.radians(g[nm]["alpha_deg"])
    assert dp["sphere"]["sphere_par"] == 0.02 and dp["cone"]["minconeopang"] == math.radians(2)
    it = dp["iteration"]
    assert (it["drawN"], it["minsubsetN"], it["prob_det"], it["τ"], it["itermax"]) == (3, 15, 0.9, 900, 1000)
    assert [R.strt(T.__new__(T)) for T in it["shape_types"]] == g["default_shape_order"]
    # ransacparameters(; sphere=(ϵ=0.9, α=deg2rad(1),), plane=(ϵ=1.0,)) -- utilitytests.jl:89-93
    rp = R.ransacparameters(sphere={"ϵ": 0.9, "α": math.radians(1)}, plane={"ϵ": 1.0})
    assert rp["sphere"] == {"ϵ": 0.9, "α": math.radians(1), "sphere_par": 0.02}
    assert rp["plane"] == {"ϵ": 1.0, "α": math.radians(5)}
    # array method -- utilitytests.jl:96-114
    rpn = R.ransacparameters([R.FittedCone, R.FittedCylinder], cone={"ϵ": 0.9, "α": math.radians(12)}, cylinder={"ϵ": 0.001})
    assert rpn["cone"] == {"ϵ": 0.9, "α": math.radians(12), "minconeopang": math.radians(2)}
    assert rpn["cylinder"] == {"ϵ": 0.001, "α": math.radians(5)}
    assert set(rpn) == {"iteration", "common", "cone", "cylinder"}


def test_dummysphere_through_product_fit(golden):  # test/dummyspheretest.jl:14-49
    g = golden["dummysphere"]
    alfi = math.radians(g["sphere_alpha_deg"])
    defrp = R.ransacparameters(sphere={"ϵ": g["sphere_eps"], "α": alfi})
    for s in g["sets"]:
        fs = R.fit(R.FittedSphere, s["v"], s["n"], None, defrp)
        fp = R.fit(R.FittedPlane, s["v"], s["n"], None,
                   R.ransacparameters(defrp, plane={"α": math.pi / 2}, common={"collin_threshold": 0.2}))
        assert isinstance(fs, R.FittedSphere) == s["sphere"], s["name"]
        assert (fp is not None) == s["plane"]
        if "sphere_eps_0.01" in s:
            assert R.fit(R.FittedSphere, s["v"], s["n"], None, R.ransacparameters(defrp, sphere={"ϵ": 0.01})) is None
        if "sphere_eps10_alpha_pi2" in s:
            assert R.fit(R.FittedSphere, s["v"], s["n"], None,
                         R.ransacparameters(defrp, sphere={"ϵ": 10, "α": math.pi / 2})) is None


def _minimal_sets(kind, rng, n):
    """Minimal sets that mostly lie on a real primitive, so the fits succeed often."""
    from ransac_jl_amd import synth
    gen = {orc.PLANE: synth.plane_patch, orc.SPHERE: synth.sphere, orc.CYLINDER: synth.cylinder, orc.CONE: synth.cone}[kind]
    out = []
    for i in range(n):
        p, nn, _ = gen(3, rng)
        if i % 4 == 3:   # inward normals
            nn = -nn
        if i % 7 == 6:   # garbage set
            p = rng.uniform(0, 100, size=(3, 3))
        out.append((np.ascontiguousarray(p), np.ascontiguousarray(nn)))
    return out


@pytest.mark.parametrize("kind", [orc.PLANE, orc.SPHERE, orc.CYLINDER, orc.CONE])
def test_fit_bit_identical_to_oracle(kind):
    rng = np.random.default_rng(100 + kind)
    po = orc.default_params()
    pp = L.Params.from_buffer_copy(bytes(po))
    nfit = 0
    for p, n in _minimal_sets(kind, rng, 400):
        a = orc.fit(kind, p, n, po)
        out, ok = L.Shape(), C.c_int32()
        L.check(R.lib().rh_fit(kind, p.ctypes.data_as(C.POINTER(C.c_double)), n.ctypes.data_as(C.POINTER(C.c_double)),
                               3, C.byref(pp), C.byref(out), C.byref(ok)))
        assert bool(ok.value) == (a is not None)
        if a is not None:
            nfit += 1
            assert bytes(a) == bytes(out)   # every parameter bit
    assert nfit > 50


@pytest.mark.parametrize("kind", [orc.PLANE, orc.SPHERE, orc.CYLINDER, orc.CONE])
def test_fit_f32_bit_identical_to_oracle(kind):
    """fit on Float32 points (a Float32 cloud, octree.jl:102-109): rh_fit_f32 -- the binary32 instantiation of
    fit_shared.h, the same code the device fits of rh_ransac run -- against the oracle's own binary32 restatement
    (oracle/orc_f32.c), parameter for parameter; the results are Float32 shapes and differ from the Float64 fit of the
    same (rounded) points."""
    rng = np.random.default_rng(300 + kind)
    po = orc.default_params()
    pp = L.Params.from_buffer_copy(bytes(po))
    nfit = ndiff = 0
    for p, n in _minimal_sets(kind, rng, 400):
        p32, n32 = p.astype(np.float32), n.astype(np.float32)
        pd, nd = np.ascontiguousarray(p32, dtype=np.float64), np.ascontiguousarray(n32, dtype=np.float64)
        a = orc.fit32(kind, p32, n32, po)
        out, ok = L.Shape(), C.c_int32()
        L.check(R.lib().rh_fit_f32(kind, pd.ctypes.data_as(C.POINTER(C.c_double)), nd.ctypes.data_as(C.POINTER(C.c_double)),
                                   3, C.byref(pp), C.byref(out), C.byref(ok)))
        assert bool(ok.value) == (a is not None)
        if a is not None:
            nfit += 1
            assert bytes(a) == bytes(out)
            assert all(float(np.float32(x)) == x for x in list(out.v)[:7])
            b = orc.fit(kind, pd, nd, po)
            ndiff += b is None or bytes(b) != bytes(a)
    assert nfit > 50 and ndiff > nfit // 2
    # the mirror: fit() on float32 arrays is the Float32 fit
    for p, n in _minimal_sets(kind, rng, 5):
        m = R.fit({orc.PLANE: R.FittedPlane, orc.SPHERE: R.FittedSphere, orc.CYLINDER: R.FittedCylinder, orc.CONE: R.FittedCone}[kind],
                  p.astype(np.float32), n.astype(np.float32), None, pp)
        a = orc.fit32(kind, p, n, po)
        assert (m is None) == (a is None)


def test_estimatescore_prob_rng_match_oracle():
    rng = np.random.default_rng(5)
    for mode in (L.SCORE_INT64_WRAP, L.SCORE_F64):
        for S1, P in ((25000, 50000), (31250, 1_000_000), (312500, 10_000_000), (1562500, 50_000_000)):
            for sigma in [0, 1, 2, 306, 307, 308, 1000, S1 // 2, S1] + list(rng.integers(0, S1, 20)):
                ci = R.estimatescore(S1, P, int(sigma), mode)
                assert (ci.min, ci.max, ci.E) == orc.estimatescore(S1, P, int(sigma), mode) or math.isnan(ci.E)
    # no wrap at 50k points: both modes agree closely and E ~ sigma * N / S
    a, b = R.estimatescore(25000, 50000, 12000, L.SCORE_INT64_WRAP), R.estimatescore(25000, 50000, 12000, L.SCORE_F64)
    assert abs(a.E - b.E) < 1e-6 * abs(b.E) and abs(b.E - 24000) < 50
    for n, s, N, k in ((24904.0, 30, 50000, 3), (900.0, 15000, 10_000_000, 3), (0.0, 15, 1000, 3)):
        assert R.prob(n, s, N, k) == orc.prob(n, s, N, k)
    r1, r2 = L.Rng(), orc.Rng()
    R.lib().rh_rng_seed(C.byref(r1), 1234)
    orc.lib().orc_rng_seed(C.byref(r2), 1234)
    for n in (1, 2, 3, 50000, 10_000_000, 2 ** 40):
        for _ in range(50):
            a, b = R.lib().rh_rng_range(C.byref(r1), n), orc.lib().orc_rng_range(C.byref(r2), n)
            assert a == b and 1 <= a <= n


def test_confidence_interval_mirror(golden):  # test/confidenceintervals.jl
    ci = R.ConfidenceInterval(1.0, 3)
    assert (ci.min, ci.max, ci.E) == (1.0, 3.0, 2.0)
    with pytest.raises(ValueError):
        R.ConfidenceInterval(3, 1.0)
    n = golden["confidence_interval"]["notsoconfident"]
    for c in (R.notsoconfident(n["b"], n["a"]), R.notsoconfident(n["a"], n["b"])):
        assert (c.min, c.max, R.E(c)) == (n["min"], n["max"], n["E"])


def test_bitmapparameters_matches_oracle():
    rng = np.random.default_rng(3)
    prm = rng.uniform(-2, 5, size=(500, 2))
    comp = rng.random(500) < 0.8
    bm, im, beta = R.bitmapparameters(prm, comp, 0.2)
    bo, io, betao = orc.bitmapparameters(prm, comp, 0.2)
    assert beta == betao and np.array_equal(bm, bo) and np.array_equal(im, io) and bm.sum() > 100


def test_device_calls_fail_loudly_without_gpu():
    n = C.c_int()
    if R.lib().rh_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    xyz = np.zeros((4, 3))
    with pytest.raises(R.RansacHipError) as e:
        R.RANSACCloud(xyz, xyz, [np.array([1, 2], dtype=np.int64)])
    assert e.value.code == L.RH_E_NODEVICE
    with pytest.raises(R.RansacHipError):
        R.largestconncomp(np.ones((4, 4), dtype=bool))


def test_c_example_builds_and_fails_loudly_without_gpu():
    """examples/score_demo.c uses the ABI from plain C (no Python, no torch)."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "score_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "score_demo.c"), "-L", os.path.join(ROOT, "ransac.jl_amd"),
                           "-lransac_hip", "-lm", "-Wl,-rpath,$ORIGIN/../ransac.jl_amd", "-o", exe])
    n = C.c_int()
    if R.lib().rh_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present: the gpu-marked test runs the example")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "no ROCm-capable device" in r.stderr + r.stdout or "no HIP device" in r.stderr


def test_deterministic_trig_is_within_one_ulp_of_libm():
    """det_math.h (fdlibm-algorithm acos / sin / cos shared by host, device and oracle): <= 1 ulp from libm."""
    rng = np.random.default_rng(9)
    sh = orc.Shape()
    sh.kind = orc.CONE
    worst = 0
    for op in list(rng.uniform(0, 2 * math.pi, 2000)) + [0.0, 1e-9, math.pi / 2, math.pi, 3.0, 6.0, 50.0]:
        sh.v[6] = float(op)
        orc.lib().orc_shape_finalize(C.byref(sh))
        ps = L.Shape.from_buffer_copy(bytes(sh))
        ps.v[7] = ps.v[8] = 0.0
        R.lib().rh_shape_finalize(C.byref(ps))
        assert (ps.v[7], ps.v[8]) == (sh.v[7], sh.v[8])          # product == oracle, bit for bit
        for got, ref in ((sh.v[7], math.cos(-op / 2)), (sh.v[8], math.sin(-op / 2))):
            if got == ref:
                continue
            ulp = abs(int(np.float64(got).view(np.int64)) - int(np.float64(ref).view(np.int64)))
            worst = max(worst, ulp)
    assert worst <= 1


def test_iterationcandidates_bookkeeping():  # test/fitting.jl:1-18
    ic = R.IterationCandidates()
    fp = R.FittedPlane([0.5, 0.5, 0.5], [0, 0, 1.0])
    sc = R.ConfidenceInterval(0, 1)
    inds = np.array([1, 2, 3, 4, 5])
    assert len(ic) == 0
    R.recordscore(ic, fp, sc, inds)
    assert len(ic) == 1
    assert len(ic.shapes) == 1 and len(ic.scores) == 1 and len(ic.inpoints) == 1
    R.deleteat(ic, 1)
    assert len(ic) == 0
    assert len(ic.shapes) == 0 and len(ic.scores) == 0 and len(ic.inpoints) == 0
    # findhighestscore (fitting.jl:140-158): first maximum, strict >; overlap flag
    assert R.findhighestscore(ic) == {"index": 0, "overlap": False}
    for lo, hi in ((0, 1), (2, 4), (2, 4), (5, 6)):
        R.recordscore(ic, fp, R.ConfidenceInterval(lo, hi), inds)
    assert R.findhighestscore(ic) == {"index": 4, "overlap": False}
    R.deleteat(ic, [1, 4])
    assert R.findhighestscore(ic) == {"index": 1, "overlap": True}


def test_push2candidatesandlevels():  # test/utilitytests.jl:134-149
    fp = R.FittedPlane([0.5, 0.5, 0.5], [0, 0, 1.0])
    candidates, levels = [], []
    R.push2candidatesandlevels(candidates, fp, levels, 3)
    assert len(candidates) == 1 and len(levels) == 1
    R.push2candidatesandlevels(candidates, [fp, fp], levels, 0)
    assert len(candidates) == 3 and len(levels) == 3
    assert levels == [3, 0, 0]


def test_setfloattype():  # test/utilitytests.jl:151-189
    nta = {"α": 1.0, "somepar": "key1", "intpar": 1}
    ntb = {"α": 1, "otherpar": np.float32(0.145), "str": "str"}
    nt = {"α": 9, "ϵ": 0.1, "γ": np.float32(0.01), "shapea": nta, "shapeb": ntb}
    f32 = R.setfloattype(nt, np.float32)
    assert f32["α"] == 9 and isinstance(f32["α"], int)
    assert isinstance(f32["ϵ"], np.float32) and math.isclose(f32["ϵ"], np.float32(0.1))
    assert isinstance(f32["γ"], np.float32) and math.isclose(f32["γ"], np.float32(0.01))
    assert isinstance(f32["shapea"]["α"], np.float32) and f32["shapea"]["somepar"] == "key1" and f32["shapea"]["intpar"] == 1
    assert f32["shapeb"]["α"] == 1 and isinstance(f32["shapeb"]["otherpar"], np.float32) and f32["shapeb"]["str"] == "str"
    f64 = R.setfloattype(f32, np.float64)
    rt = math.sqrt(np.finfo(np.float32).eps)
    assert f64["α"] == 9 and isinstance(f64["ϵ"], np.float64) and math.isclose(f64["ϵ"], 0.1, rel_tol=rt)
    assert isinstance(f64["γ"], np.float64) and math.isclose(f64["γ"], 0.01, rel_tol=rt)
    assert isinstance(f64["shapea"]["α"], np.float64) and f64["shapea"]["intpar"] == 1
    assert isinstance(f64["shapeb"]["otherpar"], np.float64) and math.isclose(f64["shapeb"]["otherpar"], 0.145, rel_tol=rt)
    assert f64["shapeb"]["str"] == "str"


def test_findAABB_and_smallestdistance():  # test/utilitytests.jl:5-40
    rng = np.random.default_rng(0)
    for d in (3, 2):
        pts = np.vstack([rng.random((30, d)), -np.ones(d), 2 * np.ones(d)])
        mn, mx = R.findAABB(pts)
        assert mn.tolist() == [-1.0] * d and mx.tolist() == [2.0] * d
    assert math.isclose(R.smallestdistance([[0.0, 0], [1.0, 1], [2.2, 2]]), math.sqrt(2))
    assert math.isclose(R.smallestdistance([[0.0, 0, 0], [1.0, 1, 1], [2.2, 2, 2]]), math.sqrt(3))


def test_octview_searches_match_std(tmp_path):
    """The linear octree's lower_bound / select (fit_shared.h, shared by the device sampler and its host twin)
    against std::lower_bound and a plain list of the set bits."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "octview_search_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"),
                           "-I", os.path.join(root, "ransac.jl_amd", "csrc"),
                           os.path.join(root, "tests", "native", "octview_search_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.strip().endswith("0 bad")


def test_product_library_reads_no_environment_and_ships_no_diagnostics():
    """libransac_hip.so sits under someone else's process: it must not change what it computes because of a variable in
    that process's environment.  The A/B switches of the experiments (RH_NO_PIPELINE, ...), the skeleton-only launch
    (RH_G2_DBG), the fake-RCCL hook (RH_RCCL_LIB) and the rh_dbg_* audits exist in the diag build only
    (libransac_hip_diag.so, -DRH_DIAG); what a caller may tune goes through rh_set_option."""
    import subprocess
    out = subprocess.run(["strings", L.SO_PATH], capture_output=True, text=True, check=True).stdout
    rh = [ln for ln in out.splitlines() if "RH_" in ln]
    assert len(rh) <= 5, rh        # (what is left: HIP call texts of error messages that name RH_* constants of the sources)
    assert not any(re.search(r"\bRH_[A-Z0-9_]+=|getenv\(\"RH_", ln) for ln in rh), rh
    nm = subprocess.run(["nm", "-D", "--defined-only", L.SO_PATH], capture_output=True, text=True, check=True).stdout
    assert "rh_dbg_" not in nm and "rh_set_option" in nm
    # no source of ours calls getenv outside the diag build (rocPRIM's headers read a variable of their own)
    csrc = os.path.join(ROOT, "ransac.jl_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        text = re.sub(r"//[^\n]*", "", text)
        if f == "options.cpp":
            text = re.sub(r"#ifdef RH_DIAG.*?#endif", "", text, flags=re.S)
        assert "getenv" not in text, f
    lib = R.lib()
    assert lib.rh_build_variant() == 0
    # product keys work, diag keys are unknown here
    R.set_option("s4_rows", 8)
    assert R.get_option("s4_rows") == 8
    R.set_option("s4_rows", None)
    assert R.get_option("s4_rows") is None
    for bad_key, bad_val in (("no_pipeline", 1), ("g2_dbg", 1), ("s4_rows", 5), ("refit_path", 7)):
        with pytest.raises(R.RansacHipError):
            R.set_option(bad_key, bad_val)


def test_diag_library_exports_the_diag_header_and_knows_the_switches():
    src = open(os.path.join(ROOT, "include", "ransac_hip_diag.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", src)))
    assert syms == sorted(L.DIAG_SIGNATURES)
    d = L.lib("diag")
    assert d.rh_build_variant() == 1
    for s in syms + header_symbols():
        assert hasattr(d, s), s
    assert d.rh_set_option(None, b"no_pipeline", 1) == 0 and d.rh_set_option(None, b"no_pipeline", L.OPTION_UNSET) == 0
    # the diag build falls back to the environment; the product build does not
    v, st = C.c_int64(), C.c_int32()
    os.environ["RH_S4_R"] = "12"
    try:
        assert d.rh_get_option(None, b"s4_rows", C.byref(v), C.byref(st)) == 0 and st.value == 1 and v.value == 12
        assert R.lib("product").rh_get_option(None, b"s4_rows", C.byref(v), C.byref(st)) == 0 and st.value == 0
    finally:
        del os.environ["RH_S4_R"]


@pytest.mark.parametrize("f32", [False, True])
def test_fit_sets_equals_forcefitshapes_per_set(f32):
    """rh_fit_sets = forcefitshapes! (fitting.jl:165-173) over the minimal sets of an iteration in one call: the shapes that fit, in
    (set, type) order -- the reference's candidate order -- bit for bit what per-set rh_fit / rh_fit_f32 calls return."""
    from ransac_jl_amd import synth
    xyz, nrm, truth = synth.make_cloud(3000, ["plane", "sphere", "cylinder", "cone"], 0.1, seed=4)
    if f32:
        xyz, nrm = xyz.astype(np.float32).astype(np.float64), nrm.astype(np.float32).astype(np.float64)
    rs = np.random.default_rng(1)
    nsets = 2600                     # (from 512 sets on the call deals them to host threads: the order must not change)
    sets = rs.integers(1, 3001, size=(nsets, 3)).astype(np.int64)
    for j in range(0, nsets, 2):    # neighbours in the cloud lie on one primitive: these sets fit often
        b = int(rs.integers(1, 2900))
        sets[j] = [b, b + 1, b + 2]
    ok = (rs.random(nsets) < 0.9).astype(np.int32)
    types = [R.FittedPlane, R.FittedCone, R.FittedCylinder, R.FittedSphere]
    cp = R.params_to_c(R.ransacparameters(types))

    class PC:
        vertices, normals, is_f32 = xyz, nrm, f32
        vertices32, normals32 = xyz.astype(np.float32), nrm.astype(np.float32)
    shapes, so = R.fit_sets(PC, sets, ok, cp)
    ref = []
    fit = R.lib().rh_fit_f32 if f32 else R.lib().rh_fit
    for j, sd in enumerate(sets):
        if not ok[j]:
            continue
        p, n = np.ascontiguousarray(xyz[sd - 1]), np.ascontiguousarray(nrm[sd - 1])
        for T in types:
            out, okf = L.Shape(), C.c_int32()
            L.check(fit(R.api._KIND_OF[T], p.ctypes.data_as(C.POINTER(C.c_double)), n.ctypes.data_as(C.POINTER(C.c_double)), 3, C.byref(cp), C.byref(out), C.byref(okf)))
            if okf.value:
                ref.append((j, bytes(out)))
    assert len(ref) >= 10 and len(shapes) == len(ref)
    assert [(int(so[i]), bytes(shapes[i])) for i in range(len(ref))] == ref
    # capacity: the needed size comes back
    n_out = C.c_int32()
    arr = (L.Shape * 1)()
    rc = R.lib().rh_fit_sets(xyz.ctypes.data_as(C.POINTER(C.c_double)), nrm.ctypes.data_as(C.POINTER(C.c_double)), sets.ctypes.data_as(C.POINTER(C.c_int64)),
                             None, nsets, 3, C.byref(cp), 0, arr, None, 1, C.byref(n_out))
    assert rc == L.RH_E_CAPACITY and n_out.value > 1
