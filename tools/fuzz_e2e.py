#!/usr/bin/env python3
"""Randomised end-to-end parity: rh_ransac against the oracle's sequential ransac() over many
(cloud, parameter, mode) combinations.  Not part of the test suite (minutes of oracle time);
run on a GPU box:  python tools/fuzz_e2e.py [n_cases] [seed]"""
import os, sys, time
if __name__ == "__main__":
    os.environ.setdefault("RH_LIB_VARIANT", "diag")   # the A/B switches / rh_dbg_* audits live in the diag build (libransac_hip_diag.so)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
from oracle import oracle as orc

KINDS = {"plane": R.FittedPlane, "sphere": R.FittedSphere, "cylinder": R.FittedCylinder, "cone": R.FittedCone}


def one(case, rng, f32=False):
    n = int(rng.choice([6_000, 20_000, 40_000, 90_000]))
    names = list(rng.choice(list(KINDS), size=int(rng.integers(2, 7))))
    out_frac = float(rng.choice([0.0, 0.1, 0.3]))
    r = int(rng.choice([1, 2, 4]))
    xyz, nrm, truth = synth.make_cloud(n, names, out_frac, seed=1000 + case)
    subs = synth.make_subsets(n, r, seed=case)
    types = [KINDS[k] for k in sorted(set(names), key=lambda k: rng.random())]
    if f32:   # a Float32 cloud: the whole loop in binary32; its cone fit is not available (the scene may still hold cone points)
        types = [t for t in types if t is not R.FittedCone] or [R.FittedPlane]
    it = {"minsubsetN": int(rng.choice([15, 40, 120])), "τ": int(rng.choice([50, 300, 900])),
          "itermax": int(rng.choice([30, 120, 400])), "prob_det": float(rng.choice([0.5, 0.8, 0.9])),
          "drawN": int(rng.choice([3, 3, 3, 4]))}
    params = R.ransacparameters(types, iteration=it)
    streams = int(rng.integers(0, 2))
    octree = bool(streams and rng.integers(0, 2) == 0)
    kw = dict(score_mode=int(rng.choice([L.SCORE_F64, L.SCORE_INT64_WRAP])), sphere_uses_enabled=bool(rng.integers(0, 2)),
              sampling_streams=streams, octree_sampling=octree)
    env = None
    if streams:
        env = rng.choice([None, None, None, "RH_NO_PIPELINE", "RH_NO_FUSED_SCORE", "RH_HOST_SAMPLER", "RH_NO_CREC", "RH_NO_FUSED_SAMPLER", "RH_NO_FAST_EXTRACT"])
    if octree:
        # the chained windows (level update, store and its compaction on the device), and each of their pieces switched off
        env = rng.choice([None, None, None, None, "RH_NO_OCT_CHAIN", "RH_NO_MANAGED_STORE", "RH_NO_OCT_TAB", "RH_HOST_SAMPLER",
                          "RH_NO_FUSED_SCORE", "RH_OCT_CHAIN_W", "RH_OCT_ONE_WINDOW", "RH_OCT_WINDOW_ITERS"])
        if rng.integers(0, 3) == 0:      # thousands of minimal sets per iteration: windows beyond their launch bounds, a large store
            it["minsubsetN"] = int(rng.choice([1500, 4000]))
            it["itermax"] = int(rng.choice([12, 30]))
            params = R.ransacparameters(types, iteration=it)
    for k in ("RH_NO_PIPELINE", "RH_NO_FUSED_SCORE", "RH_HOST_SAMPLER", "RH_NO_CREC", "RH_NO_FUSED_SAMPLER", "RH_LONG_WINDOW_SETS",
              "RH_NO_FAST_EXTRACT", "RH_NO_OCT_CHAIN", "RH_NO_MANAGED_STORE", "RH_NO_OCT_TAB", "RH_OCT_CHAIN_W", "RH_OCT_ONE_WINDOW", "RH_OCT_WINDOW_ITERS"):
        os.environ.pop(k, None)
    if env in ("RH_OCT_CHAIN_W", "RH_OCT_WINDOW_ITERS"):
        os.environ[env] = str(rng.choice([1, 2, 3, 64]))
    elif env:
        os.environ[env] = "1"
    if streams and rng.integers(0, 2):   # half the cases take the long-window sampler path whatever their size
        os.environ["RH_LONG_WINDOW_SETS"] = "0"
    if rng.integers(0, 4) == 0:          # a quarter: liveness pass after the host has seen the list lengths
        os.environ["RH_NO_FAST_EXTRACT"] = "1"
    # the culled refit scan (korder.hip) on these small clouds in two cases of three, the plain scan in the third
    R.set_option("refit_path", str(rng.choice(["culled", "culled", "scan"])))
    R.set_option("st_cull", [None, 1, 2][int(rng.integers(0, 3))])   # super-tile lists for the loop's sized score launches: by size / whenever possible / never
    if f32:
        pc = R.RANSACCloud(xyz, nrm, subs, force_eltype=np.float32)
        oc = orc.Cloud(xyz, nrm, subs[0], f32=True)
    else:
        pc = R.RANSACCloud(xyz, nrm, subs)
        oc = orc.Cloud(xyz, nrm, subs[0])
    cp = R.params_to_c(params, **kw)
    seed = int(rng.integers(1, 10_000))
    got, secs, st = R.ransac(pc, cp, seed=seed, return_stats=True)
    exp = oc.ransac(orc.Params.from_buffer_copy(bytes(cp)), seed=seed)
    ok = (exp["rc"] == 0 and st["iterations"] == exp["iterations"] and st["candidates_scored"] == exp["candidates_scored"]
          and st["scored_left"] == exp["scored_left"] and st["draws"] == exp["draws"] and len(got) == len(exp["shapes"])
          and all(bytes(g.c_shape) == bytes(e["shape"]) and np.array_equal(g.inpoints, e["inpoints"])
                  and g.score_E == e["score_E"] and g.iteration == e["iteration"] for g, e in zip(got, exp["shapes"]))
          and np.array_equal(pc.enabled_chunks(), oc.get_enabled()))
    desc = ("f32 " if f32 else "") + "n=%d r=%d prims=%s types=%s it=%s %s env=%s seed=%d -> %d shapes, %d cands, %d its" % (
        n, r, "".join(k[0] for k in names), "".join(R.strt(t)[0] if hasattr(R, "strt") and not isinstance(t, type) else t.__name__[6] for t in types),
        it, {k: int(v) for k, v in kw.items()}, env, seed, len(got), st["candidates_scored"], st["iterations"])
    return ok, desc


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    bad = 0
    t0 = time.time()
    for case in range(ncases):
        ok, desc = one(case, rng, f32=bool(os.environ.get("F32")))   # F32=1: Float32 clouds against the oracle's binary32 loop
        print("%s case %2d  %s" % ("ok  " if ok else "FAIL", case, desc), flush=True)
        bad += not ok
    print("%d cases, %d failures, %.0f s" % (ncases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
