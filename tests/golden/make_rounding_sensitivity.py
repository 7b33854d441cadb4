"""Writes rounding_sensitivity.json: how many inlier decisions of the hot path depend on the UNPINNED part of
the oracle -- the evaluation order of StaticArrays' dot / norm / normalize / cross, which cannot be checked
against Julia here -- and on the last ulp of the cone's trig.

The oracle is compiled once per reading (oracle/Makefile `variants`, ransac_oracle.c header):
    0 default   dot = (a1*b1 + a2*b2) + a3*b3, norm = sqrt of the same sum, normalize = inv(norm) * a
    1 fma       every a*b + c fused (what @muladd / FMA contraction would give)
    2 div       normalize(a) = a / norm(a)
    3 scaled    norm(a) = m * sqrt(sum((a_i/m)^2))
    4 pairwise  dot = a1*b1 + (a2*b2 + a3*b3)
    5 libm      cone acos / cos / sin from the platform libm instead of the fdlibm restatement
and every variant scores the SAME inputs as the default:
    * the bench's batch (synth.jittered_candidates(truth, 4096, seed=0)) against subset 1 of cfg1, cfg2, cfg3
      and of a 2M-point cloud with cfg5's primitive mix (cones), masks compared bit by bit;
    * the full-cloud refit scan of every ground-truth primitive (cfg3: the 40 scans of the end-to-end run);
    * the whole ransac() loop on cfg1 (fits included), extracted index sets compared.
It also counts how many (candidate, point) tests have a compared quantity within 1e-15 ... 1e-8 of its
threshold under the default reading: the population a different rounding order could flip at all.

Run:  python tests/golden/make_rounding_sensitivity.py [cfg1 cfg2 cfg3 cones]     (minutes; 8 threads)
tests/test_rounding_sensitivity.py re-derives the cfg1 block on every CPU test run and compares it with the file.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from ransac_jl_amd import synth  # noqa: E402

EDGES = [1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8]
KMAP = {"plane": orc.PLANE, "sphere": orc.SPHERE, "cylinder": orc.CYLINDER, "cone": orc.CONE}
B = 4096
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rounding_sensitivity.json")


def workload(name):
    if name == "cones":   # cfg5's mix at 2M points: the cone test is the longest rounding chain of the path
        prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12 + ["cone"] * 8
        xyz, nrm, truth = synth.make_cloud(2_000_000, prim, 0.30, seed=5, scanner=[synth.BOX / 2] * 3)
        return dict(xyz=xyz, nrm=nrm, truth=truth, r=32, seed=5)
    return synth.config(name)


def shapes_for(L, cands):
    arr = (orc.Shape * len(cands))()
    for i, (kind, outw, v) in enumerate(cands):
        arr[i].kind = KMAP[kind]
        arr[i].outwards = int(outw)
        for j, x in enumerate(v):
            arr[i].v[j] = float(x)
        L.orc_shape_finalize(arr[i])
    return arr


def truth_shapes(truth):
    out = []
    for t in truth:
        if t["kind"] == "plane":
            out.append(("plane", False, list(t["point"]) + list(t["normal"])))
        elif t["kind"] == "sphere":
            out.append(("sphere", True, list(t["center"]) + [t["radius"]]))
        elif t["kind"] == "cylinder":
            out.append(("cylinder", True, list(t["axis"]) + list(t["center"]) + [t["radius"]]))
        else:
            out.append(("cone", True, list(t["apex"]) + list(t["axis"]) + [t["opang"]]))
    return out


def popcount(a):
    return int(np.unpackbits(a.view(np.uint8)).sum()) if a.size < (1 << 24) else sum(
        int(np.unpackbits(c.view(np.uint8)).sum()) for c in np.array_split(a.reshape(-1), 64))


def measure(name, nthreads=8, b=B, refits=True, e2e=False, variants=(1, 2, 3, 4, 5)):
    c = workload(name)
    n = c["xyz"].shape[0]
    subs = synth.make_subsets(n, c["r"], c["seed"])
    kinds = sorted({t["kind"] for t in c["truth"]}, key=list(KMAP).index)
    types = [KMAP[k] for k in kinds]
    params = orc.default_params(shape_types=types)
    cands = synth.jittered_candidates(c["truth"], b, seed=0)
    tshapes = truth_shapes(c["truth"])
    res = {"points": n, "subset_points": int(subs[0].size), "candidates": b, "kinds": kinds,
           "tests": int(b * subs[0].size), "eps": 0.3, "alpha_deg": 5.0, "edges": EDGES}

    L0 = orc.lib()
    oc0 = orc.Cloud(c["xyz"], c["nrm"], subs[0])
    arr0 = shapes_for(L0, cands)
    cnt0, m0 = oc0.score_masks_mt(arr0, params, nthreads)
    res["inliers_default"] = int(cnt0.sum())
    hist = oc0.margin_census(arr0, params, EDGES, nthreads)
    res["near_threshold"] = {"distance_side_vs_eps": hist[0].tolist(), "angle_side_vs_cos_alpha": hist[1].tolist(),
                             "note": "tests of the batch whose compared quantity lies within edges[k] of its threshold "
                                     "(absolute difference, default reading); cumulative"}
    ref0 = None
    if refits:
        tarr0 = shapes_for(L0, tshapes)
        ref0 = [oc0.refit(tarr0[i], params) for i in range(len(tshapes))]
        res["refit_scans"] = len(tshapes)
        res["refit_inliers_default"] = int(sum(r.size for r in ref0))
    e0 = None
    if e2e:
        e0 = oc0.ransac(params, seed=1234)
        oc0.enable_all()
        res["e2e_shapes_default"] = len(e0["shapes"])
        res["e2e_inliers_default"] = int(sum(s["inpoints"].size for s in e0["shapes"]))

    res["variants"] = {}
    for v in variants:
        if v == 5 and "cone" not in kinds and not e2e:
            continue   # the trig only enters cone records
        Lv = orc.variant_lib(v)
        ocv = orc.Cloud(c["xyz"], c["nrm"], subs[0], L=Lv)
        arrv = shapes_for(Lv, cands)
        cntv, mv = ocv.score_masks_mt(arrv, params, nthreads)
        x = m0 ^ mv
        flips = popcount(x)
        row = {"score_mask_bits_flipped": flips,
               "candidates_with_a_flip": int((x != 0).any(axis=1).sum()),
               "candidates_with_another_count": int((cnt0 != cntv).sum())}
        del x, mv
        if refits:
            tarrv = shapes_for(Lv, tshapes)
            rf = 0
            for i in range(len(tshapes)):
                rv = ocv.refit(tarrv[i], params)
                rf += int(np.setxor1d(ref0[i], rv, assume_unique=True).size)
            row["refit_indices_flipped"] = rf
        if e2e:
            ev = ocv.ransac(params, seed=1234)
            same_n = len(ev["shapes"]) == len(e0["shapes"])
            row["e2e_same_number_of_shapes"] = bool(same_n)
            row["e2e_indices_flipped"] = int(sum(np.setxor1d(a["inpoints"], b_["inpoints"], assume_unique=True).size
                                                 for a, b_ in zip(e0["shapes"], ev["shapes"]))) if same_n else None
            row["e2e_max_rel_parameter_diff"] = float(max(
                (np.max(np.abs(np.array(a["shape"].v[:7]) - np.array(b_["shape"].v[:7])) /
                        np.maximum(1e-300, np.abs(np.array(a["shape"].v[:7]))).clip(1e-12))
                 for a, b_ in zip(e0["shapes"], ev["shapes"])), default=0.0))
        res["variants"][orc.VARIANTS[v]] = row
        del ocv
    return res


def main():
    which = sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cones"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    out["_doc"] = ("flip counts of the oracle's rounding-order variants against its default reading, on the same inputs; "
                   "generator: tests/golden/make_rounding_sensitivity.py (docstring = method)")
    for name in which:
        t0 = time.time()
        out[name] = measure(name, e2e=(name == "cfg1"))
        print(name, "%.1fs" % (time.time() - t0), json.dumps(out[name]["variants"]), flush=True)
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
