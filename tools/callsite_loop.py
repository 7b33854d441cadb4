#!/usr/bin/env python3
"""The drop-in a RANSAC.jl maintainer would try first, timed at cfg3 scale: the reference's OWN loop on the host
(iterations.jl:35-162) with only its hot call sites swapped -- scorecandidates! (one batched launch), refit,
invalidate_indexes! -- and the iteration's minimal sets drawn either by one round trip per point (rh_rng_range +
rh_select_enabled: the per-sample enabled gather of fitting.jl:405-407) or by ONE rh_sample_sets launch per iteration
(BATCHED=1, the default).  Python stands in for the Julia host here, so the host share is an upper bound of Julia's.
    python tools/callsite_loop.py      MINSUBSET=512 ITERMAX=400 BATCHED=1|0 POINTS=10000000"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth


def main(quiet=False):
    n = int(os.environ.get("POINTS", "10000000"))
    m = int(os.environ.get("MINSUBSET", "512"))
    itermax = int(os.environ.get("ITERMAX", "400"))
    batched = os.environ.get("BATCHED", "1") != "0"
    prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
    xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=3)
    subs = synth.make_subsets(n, 32, seed=3)
    pc = R.RANSACCloud(xyz, nrm, subs)
    types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
    params = R.ransacparameters(types, iteration={"minsubsetN": m, "itermax": itermax, "τ": 900, "prob_det": 0.9})
    cp = R.params_to_c(params, score_mode=L.SCORE_F64, sphere_uses_enabled=True)
    lib = R.lib()
    rng = L.Rng()
    lib.rh_rng_seed(C.byref(rng), 1234)
    drawN = 3
    S1 = subs[0]
    stored = []          # (E, shape)
    extracted = []
    t_sample = t_fit = t_score = t_extract = 0.0
    n_en = pc.count_enabled()
    en = None
    t0 = time.perf_counter()
    cc2 = 0
    for k in range(1, itermax + 1):
        if n_en < 900:
            break
        ts = time.perf_counter()
        if batched:
            sets, ok, _ = R.sample_sets(pc, drawN, rng, m)
            sets = sets[ok]
        else:
            if en is None:
                en = pc.isenabled
            out = []
            for _ in range(m):
                first = lib.rh_rng_range(C.byref(rng), n)
                while not en[first - 1]:
                    first = lib.rh_rng_range(C.byref(rng), n)
                sd = [first]
                for _q in range(1, drawN):
                    pick = int(R.select_enabled(pc, [lib.rh_rng_range(C.byref(rng), n_en)])[0])
                    if pick == first:
                        pick = int(R.select_enabled(pc, [lib.rh_rng_range(C.byref(rng), n_en)])[0])
                    sd.append(pick)
                if len(set(sd)) == drawN:
                    out.append(sd)
            sets = np.asarray(out, dtype=np.int64).reshape(-1, drawN)
        t_sample += time.perf_counter() - ts
        ts = time.perf_counter()
        cands, _ = (R.fit_sets(pc, sets, None, cp) if len(sets) else ([], None))   # forcefitshapes! over the iteration's sets (rh_fit_sets)
        t_fit += time.perf_counter() - ts
        ts = time.perf_counter()
        if cands:
            arr = (L.Shape * len(cands))(*cands)
            counts = np.zeros(len(cands), dtype=np.int32)
            L.check(lib.rh_score_batch(pc._h, arr, len(cands), C.byref(cp), counts.ctypes.data_as(C.POINTER(C.c_int32)), None))
            for s, cnt in zip(cands, counts):
                ci = R.estimatescore(S1.size, n, int(cnt), cp.score_mode)
                stored.append((ci.E, s, int(cnt)))
        t_score += time.perf_counter() - ts
        cc2 = k * m
        ts = time.perf_counter()
        if stored:
            bi = max(range(len(stored)), key=lambda i: (stored[i][0], -i))
            if R.prob(stored[bi][0], cc2, n, drawN) > 0.9:
                ex = R.refit(stored[bi][1], pc, cp)
                R.invalidate_indexes(pc, ex.inpoints)
                extracted.append(ex)
                n_en -= ex.inpoints.size
                en = None
                # removeinvalidshapes!: the stored candidates are re-scored against the new enabled bits (a candidate that lost a
                # point of its own is dropped upstream; here: dropped when its count changed)
                keep = [s for i, s in enumerate(stored) if i != bi]
                if keep:
                    arr = (L.Shape * len(keep))(*[s[1] for s in keep])
                    cn = np.zeros(len(keep), dtype=np.int32)
                    L.check(lib.rh_score_batch(pc._h, arr, len(keep), C.byref(cp), cn.ctypes.data_as(C.POINTER(C.c_int32)), None))
                    stored = [s for s, c2 in zip(keep, cn) if c2 == s[2]]
                else:
                    stored = []
        t_extract += time.perf_counter() - ts
        if R.prob(900, cc2, n, drawN) > 0.9:
            break
    t = time.perf_counter() - t0
    stats = {"shapes": len(extracted), "seconds": t, "iterations": k, "sets_per_sec": k * m / t, "sample_s": t_sample, "fit_s": t_fit, "score_s": t_score,
             "extract_s": t_extract, "inliers": int(sum(e.inpoints.size for e in extracted))}
    if quiet:
        return stats
    print("callsite loop (%s sampling): points %d minsubsetN %d iterations %d -> %d shapes in %.3f s = %.1f shapes/s, %.0f sets/s | sample %.3f fit %.3f score %.3f extract %.3f s"
          % ("batched rh_sample_sets" if batched else "per-point select", n, m, k, len(extracted), t, len(extracted) / t, k * m / t, t_sample, t_fit, t_score, t_extract), flush=True)
    return stats


if __name__ == "__main__":
    main()
