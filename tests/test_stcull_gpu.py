"""The super-tile lists of the culled score kernel (score4.hip: st_cull_kernel + the LIST instantiations; rh_set_option
"st_cull"): a candidate whose culling record rules out the box of a super-tile (16 groups of 64 points) is never shown to the
tiles of that super-tile.  Whether a launch takes the lists is a matter of size; its results must not depend on it: counts and
masks with the lists forced on equal those with the lists off and the oracle's -- ragged last super-tiles and tiles, every
kind, thresholds from tiny to huge, disabled points, non-finite points and candidates, Float32 clouds."""
import ctypes as C

import numpy as np
import pytest

import ransac_jl_amd as R
from ransac_jl_amd import _lib as L
from ransac_jl_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _orc_shapes(arr, b):
    out = (orc.Shape * max(1, b))()
    C.memmove(out, arr, C.sizeof(L.Shape) * b)
    return out


def _cands(truth, b, seed):
    import bench
    return bench.shapes_to_c(R, L, synth.jittered_candidates(truth, b, seed=seed))


@pytest.mark.parametrize("n,r,b", [(66_000, 4, 1500),      # 258 groups: 17 super-tiles, the last one holds 2 groups
                                   (131_072 * 2 + 64 * 5, 2, 700),   # 2053 groups: tiles and super-tiles both ragged
                                   (9_000 * 8, 8, 300),    # 141 groups: 9 super-tiles
                                   (400_000, 4, 4096)])
def test_lists_on_equals_lists_off_and_the_oracle(n, r, b):
    prim = ["plane", "sphere", "cylinder", "cone", "plane", "cylinder"]
    xyz, nrm, truth = synth.make_cloud(n, prim, 0.25, seed=n % 97)
    subs = synth.make_subsets(n, r, seed=7)
    pc = R.RANSACCloud(xyz, nrm, subs)
    oc = orc.Cloud(xyz, nrm, subs[0])
    rng = np.random.default_rng(n)
    for eps_scale, alpha in ((1.0, 5.0), (0.01, 1.0), (30.0, 60.0)):
        params = R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone])
        for k in ("plane", "sphere", "cylinder", "cone"):
            params[k]["ϵ"] = 0.3 * eps_scale
            params[k]["α"] = float(np.radians(alpha))
        cp = R.params_to_c(params)
        arr = _cands(truth, b, seed=int(eps_scale * 10))
        en = rng.random(n) < (0.6 if eps_scale == 1.0 else 1.0)
        pc.set_enabled(en)
        bits = np.zeros(((n + 63) // 64) * 64, dtype=np.uint8); bits[:n] = en
        oc.set_enabled(np.packbits(bits, bitorder="little").view(np.uint64))
        want_c, want_m = oc.score_batch(_orc_shapes(arr, b), orc.Params.from_buffer_copy(bytes(cp)), want_masks=True)
        for mode in (1, 2, None):
            with R.option("st_cull", mode, cloud=pc):
                got_c, got_m = R.score_batch(pc, arr, cp, want_masks=True)
                assert np.array_equal(got_c, want_c), (mode, eps_scale)
                assert np.array_equal(got_m, want_m), (mode, eps_scale)
                assert np.array_equal(R.score_batch(pc, arr, cp), want_c), (mode, eps_scale)
    assert int(want_c.sum()) > 0


def test_lists_with_non_finite_points_and_candidates():
    n = 80_000
    xyz, nrm, truth = synth.make_cloud(n, ["plane", "sphere", "cylinder", "cone"], 0.2, seed=5)
    xyz = xyz.copy(); nrm = nrm.copy()
    xyz[1000] = [np.nan, 1.0, 2.0]; xyz[2000] = [np.inf, 0.0, 0.0]; nrm[3000] = [np.nan, np.nan, np.nan]; xyz[4000:4070] = np.nan   # (a whole group and more)
    subs = [np.arange(1, n + 1, dtype=np.int64)]
    pc = R.RANSACCloud(xyz, nrm, subs)
    oc = orc.Cloud(xyz, nrm, subs[0])
    cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]))
    b = 600
    arr = _cands(truth, b, seed=3)
    for i, j in ((5, 0), (17, 3), (40, 6)):
        arr[i].v[j] = float("nan")
    arr[77].v[1] = float("inf")
    want = oc.score_batch(_orc_shapes(arr, b), orc.Params.from_buffer_copy(bytes(cp)), want_masks=True)
    for mode in (1, 2):
        with R.option("st_cull", mode, cloud=pc):
            got = R.score_batch(pc, arr, cp, want_masks=True)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), mode


def test_lists_on_a_float32_cloud():
    n = 150_000
    xyz, nrm, truth = synth.make_cloud(n, ["plane", "sphere", "cylinder", "cone", "plane"], 0.3, seed=11)
    x32, n32 = xyz.astype(np.float32), nrm.astype(np.float32)
    subs = synth.make_subsets(n, 2, seed=11)
    pc = R.RANSACCloud(x32, n32, subs, force_eltype=np.float32)
    oc = orc.Cloud32(x32, n32, subs[0])
    cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]))
    b = 900
    arr = _cands(truth, b, seed=8)
    for i in range(b):
        R.lib().rh_shape_finalize_f32(C.byref(arr[i]))
    want = oc.score_batch(_orc_shapes(arr, b), orc.Params.from_buffer_copy(bytes(cp)), want_masks=True)
    for mode in (1, 2):
        with R.option("st_cull", mode, cloud=pc):
            got = R.score_batch(pc, arr, cp, want_masks=True)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), mode


def test_lists_with_batches_in_flight_at_full_size():
    """cfg5's cloud (6104 tiles: the size at which the library takes the lists by itself), 2048 candidates with cones, two
    batches in flight: counts equal the lists-off launch's; 256 of them the oracle's."""
    import torch
    from ransac_jl_amd import dist as rdist
    c = synth.config("cfg5")
    subs = synth.make_subsets(len(c["xyz"]), c["r"], c["seed"])
    pc = R.RANSACCloud(c["xyz"], c["nrm"], subs)
    types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]
    cp = R.params_to_c(R.ransacparameters(types))
    b = 2048
    arr = _cands(c["truth"], b, seed=0)
    batch = rdist.DeviceBatch(pc, arr, b)
    lib = R.lib()
    outs = {}
    for mode in (2, None):
        with R.option("st_cull", mode, cloud=pc), R.option("batches_in_flight", 2, cloud=pc):
            ring = [torch.zeros(b, dtype=torch.int32, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()
            for k in range(4):
                L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), b, C.byref(cp), C.c_void_p(ring[k & 1].data_ptr()), None))
            L.check(lib.rh_cloud_sync(pc._h))
            assert torch.equal(ring[0], ring[1])
            outs[mode] = ring[0].cpu().numpy()
    assert np.array_equal(outs[2], outs[None]) and int(outs[2].sum()) > 1_000_000
    oc = orc.Cloud(c["xyz"], c["nrm"], subs[0])
    sel = np.arange(0, b, 8)
    sub_arr = (orc.Shape * len(sel))()
    for j, i in enumerate(sel):
        C.memmove(C.byref(sub_arr[j]), C.byref(arr[int(i)]), C.sizeof(L.Shape))
    assert np.array_equal(oc.score_batch(sub_arr, orc.Params.from_buffer_copy(bytes(cp))), outs[None][sel])
    batch.free()


def test_launch_info_and_the_list_launch_time():
    """rh_score_launch_info names the last sized launch (rows, lists, grid); rh_score_batch_dev_timed times the score launch
    alone and rh_last_list_launch_ms the list launch in front of it (0 when the launch took no lists)"""
    from ransac_jl_amd import dist as rdist
    n = 200_000
    xyz, nrm, truth = synth.make_cloud(n, ["plane", "sphere", "cylinder"], 0.2, seed=2)
    subs = synth.make_subsets(n, 2, seed=2)             # 100 000 subset points: 1563 groups, 391 tiles, 98 super-tiles
    pc = R.RANSACCloud(xyz, nrm, subs)
    cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder]))
    b = 2048
    arr = _cands(truth, b, seed=1)
    batch = rdist.DeviceBatch(pc, arr, b)
    lib = R.lib()
    d_cn = C.c_void_p()
    L.check(lib.rh_dev_alloc(pc._h, 4 * b, C.byref(d_cn)))
    got = {}
    for mode in (2, 1):
        with R.option("st_cull", mode, cloud=pc):
            ms = (C.c_float * 5)()
            ms[0] = -1.0
            L.check(lib.rh_score_batch_dev_timed(pc._h, batch.slice_ptr(0), b, C.byref(cp), d_cn, None, ms))
            info = (C.c_int32 * 4)()
            L.check(lib.rh_score_launch_info(pc._h, info))
            lm = C.c_float()
            L.check(lib.rh_last_list_launch_ms(pc._h, C.byref(lm)))
            cn = np.zeros(b, dtype=np.int32)
            L.check(lib.rh_dev_download(pc._h, cn.ctypes.data_as(C.c_void_p), d_cn, 4 * b))
            got[mode] = (list(info), lm.value, ms[4], cn)
    assert got[2][0][1] == 0 and got[1][0][1] == 1                       # lists off / on
    assert got[2][0][3] == got[1][0][3] == (100_000 + 255) // 256        # tiles
    assert got[2][0][0] in (2, 4, 8, 12, 16) and got[2][0][2] >= 1       # rows of R chunks
    assert got[2][1] == 0.0 and got[1][1] > 0.0 and got[1][2] > 0.0 and got[2][2] > 0.0
    assert np.array_equal(got[1][3], got[2][3])
    lib.rh_dev_free(pc._h, d_cn)
    batch.free()
