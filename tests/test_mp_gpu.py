"""rh_ransac_mp: ONE scene run by several processes (one per GPU in production; here the ranks share GPU 0, which the
shared-memory exchange does not care about).  The minimal sets of every iteration are dealt round-robin to the ranks;
every rank must return exactly what the single-process rh_ransac returns -- shapes, index sets, iteration count,
candidates scored, draws -- and leave its cloud in the same state; the oracle's sequential loop is the third witness."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _scene(which):
    from ransac_jl_amd import synth
    if which == "small":
        xyz, nrm, truth = synth.make_cloud(40_000, ["plane", "sphere", "cylinder", "plane"], 0.2, seed=21)
        return xyz, nrm, synth.make_subsets(40_000, 2, seed=21), dict(minsubsetN=37, itermax=150, tau=300, prob_det=0.8), False
    if which == "cones":
        xyz, nrm, truth = synth.make_cloud(60_000, ["plane", "cone", "cylinder", "sphere", "cone"], 0.1, seed=23)
        return xyz, nrm, synth.make_subsets(60_000, 2, seed=23), dict(minsubsetN=64, itermax=200, tau=300, prob_det=0.8), True
    c = synth.config("cfg2")
    return c["xyz"], c["nrm"], synth.make_subsets(1_000_000, 32, seed=2), dict(minsubsetN=1024, itermax=600, tau=900, prob_det=0.9), False


def _params(R, L, it, cones, octree=False):
    types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder] + ([R.FittedCone] if cones else [])
    p = R.ransacparameters(types, iteration={"minsubsetN": it["minsubsetN"], "itermax": it["itermax"], "τ": it["tau"],
                                             "prob_det": it["prob_det"]})
    return R.params_to_c(p, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1, octree_sampling=octree)


def _digest(got, stats, pc):
    import hashlib
    h = hashlib.sha256()
    for g in got:
        h.update(bytes(g.c_shape))
        h.update(np.ascontiguousarray(g.inpoints).tobytes())
        h.update(np.float64(g.score_E).tobytes())
        h.update(np.int64(g.iteration).tobytes())
    h.update(pc.enabled_chunks().tobytes())
    key = (len(got), stats["iterations"], stats["candidates_scored"], stats["scored_left"], stats["draws"])
    return key, h.hexdigest()


def _rank(rank, world, name, which, octree, q, refit_path=None):
    try:
        sys.path.insert(0, ROOT)
        import ransac_jl_amd as R
        from ransac_jl_amd import _lib as L
        R.set_option("refit_path", refit_path)
        xyz, nrm, subs, it, cones = _scene(which)
        pc = R.RANSACCloud(xyz, nrm, subs, device=0)
        cp = _params(R, L, it, cones, octree)
        grp = R.MpGroup(name, rank, world)
        got, _, st = R.ransac(pc, cp, seed=77, return_stats=True, mp=grp)
        q.put((rank, _digest(got, st, pc), None))
        grp.close()
    except Exception as e:   # noqa: BLE001
        q.put((rank, None, repr(e)))


@pytest.mark.parametrize("which,world,octree,refit_path", [("small", 2, False, None), ("small", 3, False, None),
                                                            ("cones", 2, False, "culled"), ("small", 2, True, None),
                                                            ("small", 2, True, "culled"), ("cfg2", 2, False, "culled")])
def test_ransac_mp_equals_single_process(which, world, octree, refit_path):
    # refit_path: the culled refit scan (korder.hip; by default from 2^21 points on) forced on these clouds, in this process
    # and in the ranks; with the octree it also maintains the Morton-order enabled bits the sampler reads
    import ransac_jl_amd as R
    from ransac_jl_amd import _lib as L
    xyz, nrm, subs, it, cones = _scene(which)
    pc = R.RANSACCloud(xyz, nrm, subs)
    R.set_option("refit_path", refit_path, cloud=pc)      # (rh_set_option: the library reads no environment variable)
    cp = _params(R, L, it, cones, octree)
    got, _, st = R.ransac(pc, cp, seed=77, return_stats=True)
    want = _digest(got, st, pc)
    assert want[0][0] >= 2, "the scene must give shapes for the comparison to mean anything"
    if which != "cfg2":   # third witness: the oracle's strictly sequential loop over the same per-set streams
        from oracle import oracle as orc
        oc = orc.Cloud(xyz, nrm, subs[0])
        exp = oc.ransac(orc.Params.from_buffer_copy(bytes(cp)), seed=77)
        assert len(exp["shapes"]) == len(got) and exp["draws"] == st["draws"]
        for g, e in zip(got, exp["shapes"]):
            assert bytes(g.c_shape) == bytes(e["shape"]) and np.array_equal(g.inpoints, e["inpoints"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/rh_mp_test_%d_%s_%d" % (os.getpid(), which, world)
    procs = [ctx.Process(target=_rank, args=(r, world, name, which, octree, q, refit_path)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=300) for _ in procs]
    for p_ in procs:
        p_.join(timeout=60)
    for rank, dig, err in res:
        assert err is None, "rank %d: %s" % (rank, err)
        assert dig == want, "rank %d differs from the single-process run: %s vs %s" % (rank, dig[0], want[0])
