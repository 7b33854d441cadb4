"""Float32 clouds (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109) through the C ABI against the oracle's
binary32 twin (oracle/orc_f32.c): counts, masks, refit index lists, enabled bits -- bit-exact.  On such a cloud every
per-point operation is a binary32 operation; eps / cos(alpha) stay doubles and are compared after exact promotion."""
import ctypes as C
import math

import numpy as np
import pytest

import ransac_jl_amd as R
from oracle import oracle as orc
from ransac_jl_amd import _lib as L
from ransac_jl_amd import synth

pytestmark = pytest.mark.gpu
KMAP = {"plane": 0, "sphere": 1, "cylinder": 2, "cone": 3}


def shapes32(truth, b, seed):
    arr = (L.Shape * b)()
    for i, (name, outw, v) in enumerate(synth.jittered_candidates(truth, b, seed=seed)):
        arr[i].kind = KMAP[name]
        arr[i].outwards = int(outw)
        for j, x in enumerate(v):
            arr[i].v[j] = float(x)
        R.lib().rh_shape_finalize_f32(C.byref(arr[i]))
    return arr


def to_orc(arr, b):
    out = (orc.Shape * max(1, b))()
    C.memmove(out, arr, C.sizeof(L.Shape) * b)
    return out


@pytest.fixture(scope="module", params=["groups", "brute"])
def scene32(request):
    """both scorers: the culled kernel with the exact test in binary32, and the brute-force float kernel"""
    with R.option("score_path", request.param):
        yield _scene32()


def _scene32():
    prim = ["plane", "plane", "sphere", "sphere", "cylinder", "cylinder", "cone", "cone"]
    xyz, nrm, truth = synth.make_cloud(70_000, prim, 0.2, seed=41)
    x32, n32 = xyz.astype(np.float32), nrm.astype(np.float32)
    subs = synth.make_subsets(70_000, 3, seed=41)
    pc = R.RANSACCloud(x32, n32, subs, force_eltype=np.float32)
    oc = orc.Cloud32(x32, n32, subs[0])
    return pc, oc, truth, x32, n32, subs


def test_shape_finalize_f32_matches_the_oracle():
    rng = np.random.default_rng(3)
    for op in list(rng.uniform(0.02, 3.1, 500)) + [0.5, 1.0, math.pi / 2]:
        a = orc.make_shape32(orc.CONE, True, [1.1, 2.2, 3.3, 0.6, 0.64, 0.48, float(op)])
        b = L.Shape()
        b.kind = L.CONE
        b.outwards = 1
        for i, x in enumerate([1.1, 2.2, 3.3, 0.6, 0.64, 0.48, float(op)]):
            b.v[i] = x
        R.lib().rh_shape_finalize_f32(C.byref(b))
        assert bytes(a) == bytes(b)
        assert all(float(np.float32(b.v[i])) == b.v[i] for i in range(9))


def test_f32_counts_masks_refit_all_kinds(scene32):
    pc, oc, truth, x32, n32, subs = scene32
    cp = R.params_to_c(R.ransacparameters())
    op = orc.Params.from_buffer_copy(bytes(cp))
    arr = shapes32(truth, 203, seed=5)
    counts, masks = R.score_batch(pc, arr, cp, want_masks=True)
    ocounts, omasks = oc.score_batch(to_orc(arr, 203), op, want_masks=True)
    assert counts.sum() > 10000
    assert np.array_equal(counts, ocounts) and np.array_equal(masks, omasks)
    assert np.array_equal(R.score_batch(pc, arr, cp), ocounts)                 # counts-only instantiation
    for k in range(4):
        sel = [i for i in range(203) if arr[i].kind == k]
        assert counts[sel].max() > 300, "kind %d never scored inliers" % k
    # the float path is not the double path on the same numbers: some counts differ from a Float64 cloud of the same values
    pc64 = R.RANSACCloud(x32.astype(np.float64), n32.astype(np.float64), subs)
    c64 = R.score_batch(pc64, arr, cp)
    assert (c64 != counts).any() and np.abs(c64 - counts).max() < 50
    # refit + invalidate, every kind, enabled bits in play
    seen = np.zeros(len(x32), dtype=bool)
    tarr = shapes32(truth, 8, seed=0)
    for i, t in enumerate(truth):   # the ground-truth primitives themselves (the jittered ones miss their points)
        v = {"plane": list(t.get("point", [])) + list(t.get("normal", [])),
             "sphere": list(t.get("center", [])) + [t.get("radius", 0)],
             "cylinder": list(t.get("axis", [])) + list(t.get("center", [])) + [t.get("radius", 0)],
             "cone": list(t.get("apex", [])) + list(t.get("axis", [])) + [t.get("opang", 0)]}[t["kind"]]
        tarr[i].kind, tarr[i].outwards = KMAP[t["kind"]], 1
        for j, x in enumerate(v):
            tarr[i].v[j] = float(x)
        R.lib().rh_shape_finalize_f32(C.byref(tarr[i]))
    for i in (0, 2, 4, 6, 1):
        ex = R.refit(tarr[i], pc, cp)
        assert np.array_equal(ex.inpoints, oc.refit(to_orc(tarr, 8)[i], op))
        assert ex.inpoints.size > 500 and not seen[ex.inpoints - 1].any()
        seen[ex.inpoints - 1] = True
        R.invalidate_indexes(pc, ex.inpoints)
        oc.invalidate(ex.inpoints)
    assert np.array_equal(pc.enabled_chunks(), oc.get_enabled())
    counts2, masks2 = R.score_batch(pc, arr, cp, want_masks=True)
    oc2, om2 = oc.score_batch(to_orc(arr, 203), op, want_masks=True)
    assert np.array_equal(counts2, oc2) and np.array_equal(masks2, om2)
    # Q4: the sphere scorer ignores the enabled bits unless told otherwise
    cpf = R.params_to_c(R.ransacparameters(), sphere_uses_enabled=True)
    c3 = R.score_batch(pc, arr, cpf)
    assert np.array_equal(c3, oc.score_batch(to_orc(arr, 203), orc.Params.from_buffer_copy(bytes(cpf))))
    pc.enable_all(); oc.enable_all()


@pytest.mark.parametrize("eps,alpha_deg,seed", [(0.3, 5.0, 0), (0.01, 1.0, 1), (5.0, 60.0, 2)])
def test_f32_fuzz_arbitrary_candidates(scene32, eps, alpha_deg, seed):
    pc, oc, truth, x32, n32, subs = scene32
    pc.enable_all(); oc.enable_all()
    rng = np.random.default_rng(2000 + seed)
    kinds = {k: {"ϵ": eps, "α": math.radians(alpha_deg)} for k in ("plane", "sphere", "cylinder", "cone")}
    cp = R.params_to_c(R.ransacparameters(**kinds))
    b = 120
    arr = (L.Shape * b)()
    for i in range(b):
        s = arr[i]
        s.kind = i % 4
        s.outwards = int(rng.integers(0, 2))
        v = np.zeros(10)
        v[0:3] = rng.uniform(-20, 120, 3)
        d = rng.normal(size=3)
        scale = [1.0, 1.0, 1e-3, 7.5][int(rng.integers(0, 4))]
        if s.kind == L.PLANE:
            v[3:6] = d / np.linalg.norm(d) * scale
        elif s.kind == L.SPHERE:
            v[3] = [0.01, 1.0, 10.0, 60.0, -3.0][int(rng.integers(0, 5))]
        elif s.kind == L.CYLINDER:
            v[0:3] = d / np.linalg.norm(d) * scale
            v[3:6] = rng.uniform(-20, 120, 3)
            v[6] = [0.01, 1.0, 8.0, 80.0][int(rng.integers(0, 4))]
        else:
            v[3:6] = d / np.linalg.norm(d) * scale
            v[6] = rng.uniform(0.01, 3.1)
        if i % 31 == 30:
            v[int(rng.integers(0, 7))] = [float("nan"), float("inf"), 1e30][int(rng.integers(0, 3))]
        for j in range(10):
            s.v[j] = float(v[j])
        R.lib().rh_shape_finalize_f32(C.byref(s))
    counts, masks = R.score_batch(pc, arr, cp, want_masks=True)
    ocounts, omasks = oc.score_batch(to_orc(arr, b), orc.Params.from_buffer_copy(bytes(cp)), want_masks=True)
    assert np.array_equal(counts, ocounts) and np.array_equal(masks, omasks)


def test_f32_planes_through_the_cloud_given_by_far_away_points(scene32):
    """A plane is (point, normal); the point need not be near the cloud.  On a Float32 cloud the reference's distance
    oz . (p - p0) is a binary32 chain whose rounding grows with |p0| -- 1e6 away, p - p0 is only good to 0.03 -- while
    oz . p0 itself may cancel to nothing.  The classifier's margins have to bracket THAT chain: planes that cut the cloud
    at every orientation, defined by points 1e3 ... 1e7 away along directions inside the plane."""
    pc, oc, truth, x32, n32, subs = scene32
    pc.enable_all(); oc.enable_all()
    rng = np.random.default_rng(77)
    cp = R.params_to_c(R.ransacparameters(plane={"ϵ": 0.3, "α": math.radians(5.0)}))
    b = 96
    arr = (L.Shape * b)()
    planes = [t for t in truth if t["kind"] == "plane"]
    assert planes
    for i in range(b):
        s = arr[i]
        s.kind = L.PLANE
        s.outwards = 0
        t = planes[i % len(planes)]
        # the scene's own planes (their points have matching normals), slightly tilted and shifted; then the point that
        # defines the plane is pushed far away INSIDE the plane (two in-plane directions)
        nrm = np.asarray(t["normal"], dtype=np.float64) + rng.normal(size=3) * 0.01
        nrm /= np.linalg.norm(nrm)
        t1 = np.cross(nrm, rng.normal(size=3)); t1 /= np.linalg.norm(t1)
        t2 = np.cross(nrm, t1)
        near = np.asarray(t["point"], dtype=np.float64) + nrm * rng.uniform(-0.2, 0.2)
        far = near + t1 * 10.0 ** rng.uniform(3, 6.5) * rng.choice([-1, 1]) + t2 * 10.0 ** rng.uniform(3, 6.5) * rng.choice([-1, 1])
        v = np.zeros(10)
        v[0:3] = far
        v[3:6] = nrm
        for j in range(10):
            s.v[j] = float(v[j])
        R.lib().rh_shape_finalize_f32(C.byref(s))
    counts, masks = R.score_batch(pc, arr, cp, want_masks=True)
    ocounts, omasks = oc.score_batch(to_orc(arr, b), orc.Params.from_buffer_copy(bytes(cp)), want_masks=True)
    assert np.array_equal(counts, ocounts) and np.array_equal(masks, omasks)
    assert ocounts.sum() > 1000 and (ocounts > 0).sum() > b // 2   # the planes do cut the cloud


def test_f32_device_batch_and_unsupported_calls(scene32):
    pc, oc, truth, x32, n32, subs = scene32
    pc.enable_all(); oc.enable_all()
    from ransac_jl_amd import dist as rdist
    import torch
    cp = R.params_to_c(R.ransacparameters())
    arr = shapes32(truth, 300, seed=9)
    batch = rdist.DeviceBatch(pc, arr, 300)
    counts = torch.zeros(300, dtype=torch.int32, device="cuda")
    L.check(R.lib().rh_score_batch_dev(pc._h, batch.slice_ptr(0), 300, C.byref(cp), C.c_void_p(counts.data_ptr()), None))
    L.check(R.lib().rh_cloud_sync(pc._h))
    assert np.array_equal(counts.cpu().numpy(), oc.score_batch(to_orc(arr, 300), orc.Params.from_buffer_copy(bytes(cp))))
    batch.free()
    with pytest.raises(R.RansacHipError):
        R.refit_lsq(arr[0], pc, cp)      # (the least-squares refit stays Float64-only)


def test_f32_full_size_refit_halves_the_bytes():
    """cfg3 at full size as a Float32 cloud: refit lists equal the oracle's on a plane and a cylinder; the scan streams
    24 bytes per point."""
    c = synth.config("cfg3")
    x32, n32 = c["xyz"].astype(np.float32), c["nrm"].astype(np.float32)
    subs = synth.make_subsets(len(x32), c["r"], c["seed"])
    pc = R.RANSACCloud(x32, n32, subs, force_eltype=np.float32)
    oc = orc.Cloud32(x32, n32, subs[0])
    cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder]))
    op = orc.Params.from_buffer_copy(bytes(cp))
    arr = shapes32(c["truth"], 80, seed=8)
    for ti in (0, 30):   # a ground-truth plane and a ground-truth cylinder
        t = c["truth"][ti]
        sh = (R.FittedPlane(t["point"], t["normal"]) if t["kind"] == "plane" else
              R.FittedCylinder(t["axis"], t["center"], t["radius"], True))
        cs = R.shape_f32(sh)
        ex = R.refit(cs, pc, cp)
        assert ex.inpoints.size > 100_000
        assert np.array_equal(ex.inpoints, oc.refit(orc.Shape.from_buffer_copy(bytes(cs)), op))
    a, b = C.c_float(), C.c_float()
    L.check(R.lib().rh_last_refit_ms(pc._h, C.byref(a), C.byref(b)))
    assert a.value < 0.08          # the Float64 scan of the same cloud takes 0.081 ms
    counts = R.score_batch(pc, arr, cp)
    assert np.array_equal(counts, oc.score_batch(to_orc(arr, 80), op, nthreads=16))


# ---- ransac() on a Float32 cloud: fits, scoring, candidate liveness and refit in binary32 (octree.jl:102-109,
# utilities.jl:488-503, plane.jl:33-57, sphere.jl:29-114, cylinder.jl:34-168) against the oracle's binary32 loop
def _same_run(got, st, exp, pc, oc):
    assert exp["rc"] == 0
    assert st["iterations"] == exp["iterations"] and st["candidates_scored"] == exp["candidates_scored"]
    assert st["scored_left"] == exp["scored_left"] and st["draws"] == exp["draws"]
    assert len(got) == len(exp["shapes"])
    for g, e in zip(got, exp["shapes"]):
        assert bytes(g.c_shape) == bytes(e["shape"])                      # fitted parameters: the same binary32 numbers
        assert np.array_equal(g.inpoints, e["inpoints"])
        assert g.score_E == e["score_E"] and g.iteration == e["iteration"]
        assert all(float(np.float32(x)) == x for x in list(g.c_shape.v)[:7])   # a Float32 shape
    assert np.array_equal(pc.enabled_chunks(), oc.get_enabled())


@pytest.mark.parametrize("streams,octree", [(0, False), (1, False), (1, True)])
def test_f32_ransac_cfg1_end_to_end(streams, octree):
    c = synth.config("cfg1")
    subs = synth.make_subsets(50000, c["r"], c["seed"])
    pc = R.RANSACCloud(c["xyz"], c["nrm"], subs, force_eltype=np.float32)
    oc = orc.Cloud(c["xyz"], c["nrm"], subs[0], f32=True)
    params = R.ransacparameters([R.FittedPlane, R.FittedSphere])
    cp = R.params_to_c(params, sampling_streams=streams, octree_sampling=octree)
    got, _, st = R.ransac(pc, cp, seed=1234, return_stats=True)
    exp = oc.ransac(orc.Params.from_buffer_copy(bytes(cp)), seed=1234)
    assert len(got) == 2 and {R.strt(g.shape) for g in got} == {"plane", "sphere"}
    _same_run(got, st, exp, pc, oc)
    # and it is NOT the Float64 run on the same values: the fitted parameters differ in their low bits
    pc64 = R.RANSACCloud(pc.vertices, pc.normals, subs)
    got64, _ = R.ransac(pc64, cp, seed=1234)
    assert any(bytes(a.c_shape) != bytes(b.c_shape) for a, b in zip(got, got64))


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("env", [None] + [pytest.param(e, marks=pytest.mark.diag) for e in ("RH_HOST_SAMPLER", "RH_NO_FUSED_SCORE", "RH_NO_FAST_EXTRACT", "RH_NO_PIPELINE")])
def test_f32_ransac_multi_primitive(seed, env, monkeypatch):
    prim = ["plane", "plane", "sphere", "cylinder", "cylinder", "sphere"]
    xyz, nrm, truth = synth.make_cloud(60_000, prim, 0.2, seed=40 + seed)
    subs = synth.make_subsets(60_000, 2, seed=seed)
    if env:
        monkeypatch.setenv(env, "1")
    pc = R.RANSACCloud(xyz, nrm, subs, force_eltype=np.float32)
    oc = orc.Cloud(xyz, nrm, subs[0], f32=True)
    params = R.ransacparameters([R.FittedPlane, R.FittedCylinder, R.FittedSphere],
                                iteration={"minsubsetN": 200, "itermax": 60, "τ": 300, "prob_det": 0.9})
    cp = R.params_to_c(params, score_mode=L.SCORE_F64, sphere_uses_enabled=bool(seed & 1), sampling_streams=1, octree_sampling=(seed == 3))
    got, _, st = R.ransac(pc, cp, seed=100 + seed, return_stats=True)
    exp = oc.ransac(orc.Params.from_buffer_copy(bytes(cp)), seed=100 + seed)
    assert len(got) >= 3 and len({R.strt(g.shape) for g in got}) >= 2
    _same_run(got, st, exp, pc, oc)


@pytest.mark.parametrize("streams,octree", [(0, False), (1, False), (1, True)])
def test_f32_ransac_with_cones(streams, octree):
    """ransac() on a Float32 cloud with FittedCone in shape_types (round 5: the cone's fit in binary32 -- a one-sided Jacobi SVD
    and an LU in float for cone.jl:40-50's rank / \\, fit_shared.h fit_cone_t<float>; on the host for the sequential stream,
    on the device for per-set streams) against the oracle's own binary32 loop: shapes, index lists, draws, bit for bit."""
    prim = ["plane", "cone", "cylinder", "cone", "sphere"]
    xyz, nrm, truth = synth.make_cloud(50_000, prim, 0.15, seed=77)
    subs = synth.make_subsets(50_000, 2, seed=7)
    pc = R.RANSACCloud(xyz, nrm, subs, force_eltype=np.float32)
    oc = orc.Cloud(xyz, nrm, subs[0], f32=True)
    params = R.ransacparameters([R.FittedPlane, R.FittedCone, R.FittedCylinder, R.FittedSphere],
                                iteration={"minsubsetN": 150, "itermax": 80, "τ": 300, "prob_det": 0.8})
    cp = R.params_to_c(params, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=streams, octree_sampling=octree)
    got, _, st = R.ransac(pc, cp, seed=31, return_stats=True)
    exp = oc.ransac(orc.Params.from_buffer_copy(bytes(cp)), seed=31)
    assert exp["rc"] == 0 and len(got) >= 3
    assert any(g.c_shape.kind == L.CONE for g in got), [R.strt(g.shape) for g in got]
    _same_run(got, st, exp, pc, oc)
    for g in got:      # a Float32 shape holds binary32 numbers
        assert all(float(np.float32(x)) == x for x in list(g.c_shape.v)[:9])
