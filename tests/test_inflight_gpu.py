"""rh_score_batch_dev (counts, or counts and masks) with several batches in flight (rh_set_option "batches_in_flight", include/ransac_hip.h): batches take
turns on the cloud's stream and on further streams with workspaces of their own.  Whatever the number in flight, every
batch's counts must equal the counts of the same batch scored alone (and the oracle's), with other calls on the cloud --
mask batches, disabling points, host-side scoring -- cut in between: those join the streams first."""
import ctypes as C

import numpy as np
import pytest

import ransac_jl_amd as R
from ransac_jl_amd import _lib as L
from ransac_jl_amd import dist as rdist, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    import bench
    prim = ["plane", "plane", "sphere", "cylinder", "cone", "cylinder"]
    n = 400_000
    xyz, nrm, truth = synth.make_cloud(n, prim, 0.25, seed=31)
    subs = synth.make_subsets(n, 8, seed=31)       # 50 000 subset points: the culled kernel
    pc = R.RANSACCloud(xyz, nrm, subs)
    oc = orc.Cloud(xyz, nrm, subs[0])
    cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]))
    nb, b = 9, 1500
    arrs = [bench.shapes_to_c(R, L, synth.jittered_candidates(truth, b, seed=100 + i)) for i in range(nb)]
    batches = [rdist.DeviceBatch(pc, a, b) for a in arrs]
    yield pc, oc, cp, arrs, batches, b
    for bt in batches:
        bt.free()


def _alone(pc, cp, batches, b, sizes):
    import torch
    lib = R.lib()
    ref = []
    for bt, nb in zip(batches, sizes):
        out = torch.zeros(b, dtype=torch.int32, device="cuda")
        L.check(lib.rh_score_batch_dev(pc._h, bt.slice_ptr(0), nb, C.byref(cp), C.c_void_p(out.data_ptr()), None))
        L.check(lib.rh_cloud_sync(pc._h))
        ref.append(out.cpu().numpy()[:nb].copy())
    return ref


@pytest.mark.parametrize("in_flight", [2, 3, 4])
def test_batches_in_flight_equal_batches_scored_alone(scene, in_flight):
    import torch
    pc, oc, cp, arrs, batches, b = scene
    lib = R.lib()
    pc.enable_all(); oc.enable_all()
    sizes = [b, 700, b, 33, b, 1, 1200, b, 64]         # ragged: every slot sees its workspace grow and shrink
    ref = _alone(pc, cp, batches, b, sizes)
    assert sum(int(r.sum()) for r in ref) > 100_000
    # the oracle on two of them (test_parity_gpu pins the batch scored alone against the oracle at large)
    for i in (0, 6):
        sh = (orc.Shape * sizes[i])()
        C.memmove(sh, arrs[i], C.sizeof(L.Shape) * sizes[i])
        assert np.array_equal(ref[i], oc.score_batch(sh, orc.Params.from_buffer_copy(bytes(cp))))
    with R.option("batches_in_flight", in_flight, cloud=pc):
        assert R.get_option("batches_in_flight", cloud=pc) == in_flight
        ring = [torch.full((b,), 7, dtype=torch.int32, device="cuda") for _ in range(in_flight)]
        torch.cuda.synchronize()
        # one run over all nine batches without a wait, buffer k mod F for call k: the last F batches are read back after
        # the join (rh_cloud_sync); the shorter runs below read every batch back
        got = {}
        for k, (bt, nb) in enumerate(zip(batches, sizes)):
            L.check(lib.rh_score_batch_dev(pc._h, bt.slice_ptr(0), nb, C.byref(cp), C.c_void_p(ring[k % in_flight].data_ptr()), None))
        L.check(lib.rh_cloud_sync(pc._h))
        for k in range(len(sizes) - in_flight, len(sizes)):
            got[k] = ring[k % in_flight].cpu().numpy()[:sizes[k]].copy()
        for k, g in got.items():
            assert np.array_equal(g, ref[k]), (in_flight, k)
        # runs of F batches with a join after each: every batch is read back
        for base in range(0, len(sizes), in_flight):
            ks = list(range(base, min(base + in_flight, len(sizes))))
            for j, k in enumerate(ks):
                L.check(lib.rh_score_batch_dev(pc._h, batches[k].slice_ptr(0), sizes[k], C.byref(cp), C.c_void_p(ring[j].data_ptr()), None))
            L.check(lib.rh_cloud_sync(pc._h))
            for j, k in enumerate(ks):
                assert np.array_equal(ring[j].cpu().numpy()[:sizes[k]], ref[k]), (in_flight, k)
    assert R.get_option("batches_in_flight", cloud=pc) is None


def test_other_calls_join_the_batches_in_flight(scene):
    """a mask batch, rh_disable and host-side scoring between pipelined calls: each sees (and leaves) one stream's order."""
    import torch
    pc, oc, cp, arrs, batches, b = scene
    lib = R.lib()
    pc.enable_all(); oc.enable_all()
    sizes = [b] * len(batches)
    ref_all = _alone(pc, cp, batches, b, sizes)
    with R.option("batches_in_flight", 2, cloud=pc):
        ring = [torch.zeros(b, dtype=torch.int32, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        for k in (0, 1, 2):      # three in a row: slots 0, 1, 0
            L.check(lib.rh_score_batch_dev(pc._h, batches[k].slice_ptr(0), b, C.byref(cp), C.c_void_p(ring[k & 1].data_ptr()), None))
        # a host-side call with masks joins, runs on the cloud's stream, and restarts the run at slot 0
        counts_m, masks_m = R.score_batch(pc, arrs[3], cp, want_masks=True)
        assert np.array_equal(counts_m, ref_all[3])
        assert np.array_equal(np.unpackbits(masks_m.view(np.uint8), axis=1).sum(axis=1), ref_all[3])
        assert np.array_equal(ring[0].cpu().numpy(), ref_all[2]) and np.array_equal(ring[1].cpu().numpy(), ref_all[1])
        # points go away while nothing is in flight any more; the next pipelined batches see the new bits
        L.check(lib.rh_score_batch_dev(pc._h, batches[4].slice_ptr(0), b, C.byref(cp), C.c_void_p(ring[0].data_ptr()), None))
        L.check(lib.rh_score_batch_dev(pc._h, batches[5].slice_ptr(0), b, C.byref(cp), C.c_void_p(ring[1].data_ptr()), None))
        keep = np.ones(pc.size, dtype=bool)
        keep[:150_000] = False
        pc.set_enabled(keep)      # (joins: batches 4 and 5 were scored against all points)
        L.check(lib.rh_cloud_sync(pc._h))
        before = [ring[0].cpu().numpy().copy(), ring[1].cpu().numpy().copy()]
        assert np.array_equal(before[0], ref_all[4]) and np.array_equal(before[1], ref_all[5])
        L.check(lib.rh_score_batch_dev(pc._h, batches[6].slice_ptr(0), b, C.byref(cp), C.c_void_p(ring[0].data_ptr()), None))
        L.check(lib.rh_score_batch_dev(pc._h, batches[7].slice_ptr(0), b, C.byref(cp), C.c_void_p(ring[1].data_ptr()), None))
        L.check(lib.rh_cloud_sync(pc._h))
        after = [ring[0].cpu().numpy().copy(), ring[1].cpu().numpy().copy()]
    alone_after = _alone(pc, cp, batches[6:8], b, [b, b])
    assert np.array_equal(after[0], alone_after[0]) and np.array_equal(after[1], alone_after[1])
    assert int(after[0].sum()) < int(ref_all[6].sum())          # the disabled points did count before
    pc.enable_all(); oc.enable_all()


@pytest.mark.parametrize("in_flight", [2, 3])
def test_mask_batches_in_flight(scene, in_flight):
    """batches WITH mask output in flight: every batch's dense subset-order masks equal those of the host-buffer call."""
    import torch
    pc, oc, cp, arrs, batches, b = scene
    lib = R.lib()
    pc.enable_all(); oc.enable_all()
    sizes = [b, 900, b, 40, 1300, b]
    want = [R.score_batch(pc, (L.Shape * nb).from_buffer_copy(bytes(arrs[i])[: C.sizeof(L.Shape) * nb]), cp, want_masks=True)
            for i, nb in enumerate(sizes)]
    sh = (orc.Shape * sizes[1])()
    C.memmove(sh, arrs[1], C.sizeof(L.Shape) * sizes[1])
    oc_counts, oc_masks = oc.score_batch(sh, orc.Params.from_buffer_copy(bytes(cp)), want_masks=True)
    assert np.array_equal(want[1][0], oc_counts) and np.array_equal(want[1][1], oc_masks)
    w = want[0][1].shape[1]
    with R.option("batches_in_flight", in_flight, cloud=pc):
        cn = [torch.zeros(b, dtype=torch.int32, device="cuda") for _ in range(in_flight)]
        mk = [torch.zeros(b * w, dtype=torch.int64, device="cuda") for _ in range(in_flight)]
        torch.cuda.synchronize()
        for base in range(0, len(sizes), in_flight):
            ks = list(range(base, min(base + in_flight, len(sizes))))
            for j, k in enumerate(ks):
                L.check(lib.rh_score_batch_dev(pc._h, batches[k].slice_ptr(0), sizes[k], C.byref(cp), C.c_void_p(cn[j].data_ptr()),
                                               C.c_void_p(mk[j].data_ptr())))
            L.check(lib.rh_cloud_sync(pc._h))
            for j, k in enumerate(ks):
                assert np.array_equal(cn[j].cpu().numpy()[:sizes[k]], want[k][0]), (in_flight, k)
                got = mk[j].cpu().numpy().view(np.uint64)[: sizes[k] * w].reshape(sizes[k], w)
                assert np.array_equal(got, want[k][1]), (in_flight, k)
        # a long run without joins: the last F batches
        for k in range(2 * len(sizes)):
            i = k % len(sizes)
            L.check(lib.rh_score_batch_dev(pc._h, batches[i].slice_ptr(0), sizes[i], C.byref(cp), C.c_void_p(cn[k % in_flight].data_ptr()),
                                           C.c_void_p(mk[k % in_flight].data_ptr())))
        L.check(lib.rh_cloud_sync(pc._h))
        for k in range(2 * len(sizes) - in_flight, 2 * len(sizes)):
            i = k % len(sizes)
            got = mk[k % in_flight].cpu().numpy().view(np.uint64)[: sizes[i] * w].reshape(sizes[i], w)
            assert np.array_equal(got, want[i][1]) and np.array_equal(cn[k % in_flight].cpu().numpy()[:sizes[i]], want[i][0])


def test_the_same_buffer_call_after_call_is_still_right(scene):
    """a caller that ignores the buffer rule (one count buffer for every call): the library never overlaps two batches on one
    buffer -- every batch's counts are those of the batch scored alone"""
    import torch
    pc, oc, cp, arrs, batches, b = scene
    lib = R.lib()
    pc.enable_all(); oc.enable_all()
    ref = _alone(pc, cp, batches, b, [b] * len(batches))
    with R.option("batches_in_flight", 3, cloud=pc):
        one = torch.zeros(b, dtype=torch.int32, device="cuda")
        other = torch.zeros(b, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        snaps = []
        for k in range(6):
            L.check(lib.rh_score_batch_dev(pc._h, batches[k].slice_ptr(0), b, C.byref(cp), C.c_void_p(one.data_ptr()), None))
            if k == 2:   # (a second buffer in between: that batch may overlap, the next one on `one` may not)
                L.check(lib.rh_score_batch_dev(pc._h, batches[8].slice_ptr(0), b, C.byref(cp), C.c_void_p(other.data_ptr()), None))
            L.check(lib.rh_cloud_sync(pc._h))
            snaps.append(one.cpu().numpy().copy())
        for k in range(6):
            assert np.array_equal(snaps[k], ref[k]), k
        assert np.array_equal(other.cpu().numpy(), ref[8])
        # without a sync in between: the last writer wins, and it is the last call
        for k in range(6):
            L.check(lib.rh_score_batch_dev(pc._h, batches[k].slice_ptr(0), b, C.byref(cp), C.c_void_p(one.data_ptr()), None))
        L.check(lib.rh_cloud_sync(pc._h))
        assert np.array_equal(one.cpu().numpy(), ref[5])


def test_option_is_refused_out_of_range_and_host_calls_ignore_it(scene):
    pc, oc, cp, arrs, batches, b = scene
    with pytest.raises(R.RansacHipError):
        R.set_option("batches_in_flight", 5, cloud=pc)
    with R.option("batches_in_flight", 2, cloud=pc):
        c1, m1 = R.score_batch(pc, arrs[0], cp, want_masks=True)
        c2, m2 = R.score_batch(pc, arrs[0], cp, want_masks=True)
        assert np.array_equal(c1, c2) and np.array_equal(m1, m2)
