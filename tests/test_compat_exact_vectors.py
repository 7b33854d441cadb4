"""Hand-derived known-answer vectors for compatiblesPlane / Sphere / Cylinder / Cone, for scorecandidate's use of the
enabled bits (Q4) and for validatecone (Q11): tests/golden/compat_exact_vectors.json, derivations in
tests/golden/make_compat_exact_vectors.py.  None of them depends on how StaticArrays rounds (the "exact" ones have
no rounding at all), so they pin the part of the path the reference's own tests leave unpinned -- on the CPU for
the oracle and all of its rounding-order variants, under -m gpu for the HIP kernels through the C ABI."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import ransac_jl_amd as R
from oracle import oracle as orc
from ransac_jl_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = json.load(open(os.path.join(ROOT, "tests", "golden", "compat_exact_vectors.json")))
KIND = {"plane": 0, "sphere": 1, "cylinder": 2, "cone": 3}
dp = C.POINTER(C.c_double)


@pytest.fixture
def score_path_option(path):
    """rh_set_option(NULL, "score_path", path) for the clouds the test creates (the library reads no environment)"""
    with R.option("score_path", path):
        yield path



def orc_shape(c, lib=None):
    s = orc.Shape()
    s.kind = KIND[c["kind"]]
    s.outwards = int(c["outwards"])
    for i, x in enumerate(c["v"]):
        s.v[i] = x
    (lib or orc.lib()).orc_shape_finalize(C.byref(s))
    return s


def test_vector_file_is_what_the_generator_writes():
    import subprocess
    import sys
    before = open(os.path.join(ROOT, "tests", "golden", "compat_exact_vectors.json")).read()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_compat_exact_vectors.py")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr          # the rational re-derivation inside the generator holds
    assert open(os.path.join(ROOT, "tests", "golden", "compat_exact_vectors.json")).read() == before
    modes = [c["mode"] for c in VEC["compat"]]
    assert modes.count("exact") >= 20 and {c["kind"] for c in VEC["compat"]} == set(KIND)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
def test_oracle_and_every_rounding_variant_give_the_hand_derived_answers(variant):
    lib = orc.variant_lib(variant)
    for c in VEC["compat"]:
        s = orc_shape(c, lib)
        p, n = np.array(c["p"]), np.array(c["n"])
        got = bool(lib.orc_compatible(C.byref(s), p.ctypes.data_as(dp), n.ctypes.data_as(dp), c["eps"], c["cos_alpha"]))
        assert got == c["expect"], (variant, c["name"], c["derivation"])


def test_float32_oracle_twin_gives_the_hand_derived_answers():
    """orc_f32.c (the binary32 twin used for Float32 clouds) on every vector whose numbers are binary32 numbers: the
    exact ones have no rounding in binary32 either, the robust ones keep their margin."""
    n = 0
    for c in VEC["compat"]:
        rep = all(float(np.float32(x)) == x for x in c["v"][:7] + c["p"] + c["n"])
        if c["mode"] in ("exact", "exact_distance") and not rep:
            continue   # 2^-50 offsets do not exist in binary32
        s = orc.make_shape32(KIND[c["kind"]], c["outwards"], c["v"])
        assert orc.compatible32(s, c["p"], c["n"], c["eps"], c["cos_alpha"]) == c["expect"], (c["name"], c["derivation"])
        n += 1
    assert n >= 30


def _score_case_arrays(sc):
    pts, nrm = np.array(sc["points"], dtype=float), np.array(sc["normals"], dtype=float)
    en = np.ones(len(pts), dtype=bool)
    en[np.array(sc["disabled_1based"]) - 1] = False
    return pts, nrm, en


def test_oracle_scorecandidate_enabled_bits_q4():
    for sc in VEC["score"]:
        pts, nrm, en = _score_case_arrays(sc)
        oc = orc.Cloud(pts, nrm, np.arange(1, len(pts) + 1))
        oc.set_enabled(np.frombuffer(np.packbits(en, bitorder="little").tobytes().ljust(8, b"\0"), dtype=np.uint64))
        for mode, key in ((0, "count_reference"), (1, "count_fixed")):
            prm = orc.default_params(sphere_uses_enabled=mode)
            for k in range(4):
                prm.eps[k] = sc["eps"]
                prm.cos_alpha[k] = sc["cos_alpha"]
            for sh in sc["shapes"]:
                s = orc_shape(sh)
                # every listed point is compatible on its own, every other point is not
                comp = [i + 1 for i in range(len(pts)) if orc.lib().orc_compatible(
                    C.byref(s), pts[i].ctypes.data_as(dp), nrm[i].ctypes.data_as(dp), sc["eps"], sc["cos_alpha"])]
                assert comp == sh["compatible_1based"]
                cnt, inp = oc.scorecandidate(s, prm)
                assert cnt == sh[key], (sh["kind"], key)


def _fit_inputs(fc, which):
    p = np.array(fc["p3"] + [fc[which]], dtype=float)
    n = np.array(fc["n3"] + [fc["n4"]], dtype=float)
    return p, n


def test_validatecone_has_no_abs_q11_oracle_and_product_host_fit():
    for fc in VEC["fit"]:
        oprm = orc.default_params(drawN=4)
        prm = R.ransacparameters([R.FittedCone], iteration={"drawN": 4})
        for which, key in (("p4_outside", "expect_fit_with_outside"), ("p4_inside", "expect_fit_with_inside")):
            p, n = _fit_inputs(fc, which)
            o = orc.fit(orc.CONE, p, n, oprm)
            g = R.fit(R.FittedCone, p, n, None, prm)          # rh_fit: host side of the product, no GPU needed
            assert (o is not None) == fc[key] and (g is not None) == fc[key], (which, o, g)
            if fc[key]:
                e = fc["expect_cone"]
                assert np.allclose(o.v[0:3], e["apex"], atol=1e-9) and np.allclose(o.v[3:6], e["axis"], atol=1e-9)
                assert abs(o.v[6] - e["opang"]) < 1e-9 and o.outwards == 1
                assert bytes(g.to_c()) == bytes(o)             # product == oracle, bit for bit


# ------------------------------------------------------------------------------------------ GPU ----
def _vector_cloud():
    """All vector points, plus filler so that subset 1 is large enough for the culled score kernel (S >= 8192)."""
    rng = np.random.default_rng(77)
    vp = np.array([c["p"] for c in VEC["compat"]], dtype=float)
    vn = np.array([c["n"] for c in VEC["compat"]], dtype=float)
    fp = rng.uniform(-5, 45, size=(9000, 3))
    fn = rng.normal(size=(9000, 3))
    fn /= np.linalg.norm(fn, axis=1, keepdims=True)
    pts, nrm = np.concatenate([vp, fp]), np.concatenate([vn, fn])
    perm = rng.permutation(len(pts))
    pos = np.empty(len(pts), dtype=np.int64)
    pos[perm] = np.arange(len(pts))
    return np.ascontiguousarray(pts[perm]), np.ascontiguousarray(nrm[perm]), pos[: len(vp)]


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["groups", "brute"])
def test_hip_kernels_give_the_hand_derived_answers(path, score_path_option):
    pts, nrm, where = _vector_cloud()
    n = len(pts)
    sub = np.random.default_rng(3).permutation(n).astype(np.int64) + 1      # subset 1 = the whole cloud, shuffled
    subpos = np.empty(n, dtype=np.int64)
    subpos[sub - 1] = np.arange(n)
    pc = R.RANSACCloud(pts, nrm, [sub])
    oc = orc.Cloud(pts, nrm, sub)
    for i, c in enumerate(VEC["compat"]):
        cp = R.params_to_c(R.ransacparameters())
        k = KIND[c["kind"]]
        cp.eps[k] = c["eps"]
        cp.cos_alpha[k] = c["cos_alpha"]            # the thresholds cross the C ABI as numbers (include/ransac_hip.h)
        op = orc.Params.from_buffer_copy(bytes(cp))
        s = L.Shape.from_buffer_copy(bytes(orc_shape(c)))
        arr = (L.Shape * 1)(s)
        counts, masks = R.score_batch(pc, arr, cp, want_masks=True)
        j = int(subpos[where[i]])
        got = bool((int(masks[0, j >> 6]) >> (j & 63)) & 1)
        assert got == c["expect"], (path, c["name"], c["derivation"])
        oarr = (orc.Shape * 1)(orc.Shape.from_buffer_copy(bytes(s)))
        ocounts, omasks = oc.score_batch(oarr, op, want_masks=True)
        assert np.array_equal(counts, ocounts) and np.array_equal(masks, omasks)
        assert np.array_equal(R.score_batch(pc, arr, cp), ocounts)          # the counts-only instantiation
        # refit: the same test over the whole cloud in original order
        ex = R.refit(s, pc, cp)
        assert ((where[i] + 1) in set(ex.inpoints.tolist())) == c["expect"], (path, "refit", c["name"])
        assert np.array_equal(ex.inpoints, oc.refit(oarr[0], op))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["groups", "brute"])
def test_hip_float32_kernels_give_the_hand_derived_answers(path, score_path_option):
    """The vectors whose numbers are binary32 numbers on a Float32 cloud (rh_cloud_create_f32): both scorers and the
    refit scan give the derived answer and equal the oracle's binary32 twin on the whole cloud."""
    pts, nrm, where = _vector_cloud()
    p32, n32 = pts.astype(np.float32), nrm.astype(np.float32)      # exact for the representable vectors (filler rounds)
    n = len(pts)
    sub = np.random.default_rng(3).permutation(n).astype(np.int64) + 1
    subpos = np.empty(n, dtype=np.int64)
    subpos[sub - 1] = np.arange(n)
    pc = R.RANSACCloud(p32, n32, [sub], force_eltype=np.float32)
    oc = orc.Cloud32(p32, n32, sub)
    done = 0
    for i, c in enumerate(VEC["compat"]):
        rep = all(float(np.float32(x)) == x for x in c["v"][:7] + c["p"] + c["n"])
        if c["mode"] in ("exact", "exact_distance") and not rep:
            continue
        cp = R.params_to_c(R.ransacparameters())
        k = KIND[c["kind"]]
        cp.eps[k] = c["eps"]
        cp.cos_alpha[k] = c["cos_alpha"]
        op = orc.Params.from_buffer_copy(bytes(cp))
        os_ = orc.make_shape32(k, c["outwards"], c["v"])
        s = L.Shape.from_buffer_copy(bytes(os_))
        arr = (L.Shape * 1)(s)
        counts, masks = R.score_batch(pc, arr, cp, want_masks=True)
        j = int(subpos[where[i]])
        got = bool((int(masks[0, j >> 6]) >> (j & 63)) & 1)
        assert got == c["expect"], (path, c["name"], c["derivation"])
        oarr = (orc.Shape * 1)(os_)
        ocounts, omasks = oc.score_batch(oarr, op, want_masks=True)
        assert np.array_equal(counts, ocounts) and np.array_equal(masks, omasks)
        ex = R.refit(s, pc, cp)
        assert ((where[i] + 1) in set(ex.inpoints.tolist())) == c["expect"], (path, "refit", c["name"])
        assert np.array_equal(ex.inpoints, oc.refit(oarr[0], op))
        done += 1
    assert done >= 30


@pytest.mark.gpu
def test_hip_scorecandidate_enabled_bits_q4():
    for sc in VEC["score"]:
        pts, nrm, en = _score_case_arrays(sc)
        pc = R.RANSACCloud(pts, nrm, [np.arange(1, len(pts) + 1)])
        pc.set_enabled(en)
        for mode, key in ((False, "count_reference"), (True, "count_fixed")):
            cp = R.params_to_c(R.ransacparameters(), sphere_uses_enabled=mode)
            for k in range(4):
                cp.eps[k] = sc["eps"]
                cp.cos_alpha[k] = sc["cos_alpha"]
            arr = (L.Shape * len(sc["shapes"]))(*[L.Shape.from_buffer_copy(bytes(orc_shape(sh))) for sh in sc["shapes"]])
            counts = R.score_batch(pc, arr, cp)
            assert counts.tolist() == [sh[key] for sh in sc["shapes"]], (mode, counts)
