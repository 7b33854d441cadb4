#!/usr/bin/env python3
"""Event counts of ONE launch of the culled score kernel on a bench workload (WL=cfg2|cfg3|cfg5), from the diag build's
counters (rh_dbg_s4_stats: chunk visits, box-tested candidates, surviving pairs, batches, second-pass pairs, undecided
points, ring drains per kind), plus the pair census of rh_dbg_cls_soundness (pairs the box test skips, pairs whose group
holds a band point = the floor of the necessary pair work, pairs with an exact inlier).  tools/isa_account.py multiplies
these with the static instruction histogram of the kernel's regions; bench.py's `frac_necessary` uses the band pairs.
    python tools/s4_stats.py            -> gpurun_out/s4_stats_<WL>.json (and a table on stdout)"""
import ctypes as C, json, os, sys
os.environ["RH_LIB_VARIANT"] = "diag"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, dist as rdist, synth
import bench

KN = ["plane", "sphere", "cylinder", "cone"]
FIELDS = ["segments", "visits_h0", "visits_h1", "visits_h2", "visits_h3p", "candidates_box_tested", "visits_without_survivor", "surviving_pairs",
          "batches", "second_pass_pairs", "undecided_points", "ring_drains_full", "ring_drains_final", "exact_accepted", "pairs_with_count",
          "cone_compaction_rounds"]


def main():
    wl = os.environ.get("WL", "cfg3")
    prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
    types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
    n, seed, scanner, outl = 10_000_000, 3, None, 0.30
    if wl == "cfg2":
        prim = ["plane", "plane", "sphere", "sphere", "cylinder", "cylinder"]; n, seed, outl = 1_000_000, 2, 0.0
    if wl == "cfg5":
        prim += ["cone"] * 8; types += [R.FittedCone]; n, seed, scanner = 50_000_000, 5, [synth.BOX / 2] * 3
    xyz, nrm, truth = synth.make_cloud(n, prim, outl, seed=seed, scanner=scanner)
    subs = synth.make_subsets(n, 32, seed=seed)
    pc = R.RANSACCloud(xyz, nrm, subs)
    if os.environ.get("LISTS") in ("on", "off"):   # the instantiation the default line's step launches (tools/profile_round.sh)
        R.set_option("st_cull", 1 if os.environ["LISTS"] == "on" else 2, cloud=pc)
    cp = R.params_to_c(R.ransacparameters(types), score_mode=L.SCORE_F64)
    cands = synth.jittered_candidates(truth, 4096, seed=0)
    arr = bench.shapes_to_c(R, L, cands)
    batch = rdist.DeviceBatch(pc, arr, 4096)
    counts = torch.zeros(4096, dtype=torch.int32, device="cuda")
    lib = R.lib()

    def step():
        L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), 4096, C.byref(cp), C.c_void_p(counts.data_ptr()), None))
    for _ in range(20):
        step()
    L.check(lib.rh_cloud_sync(pc._h))
    ref = counts.cpu().numpy().copy()
    L.check(lib.rh_dbg_s4_stats(pc._h, 1, None))
    step()
    st = np.zeros(128, dtype=np.uint64)
    L.check(lib.rh_dbg_s4_stats(pc._h, 0, st.ctypes.data_as(C.POINTER(C.c_uint64))))
    L.check(lib.rh_dbg_s4_stats(pc._h, 2, None))
    assert np.array_equal(counts.cpu().numpy(), ref)
    snd = np.zeros(56, dtype=np.uint64)
    L.check(lib.rh_dbg_cls_soundness(pc._h, arr, 4096, C.byref(cp), snd.ctypes.data_as(C.POINTER(C.c_uint64))))
    S = int(subs[0].size)
    linfo = (C.c_int32 * 4)()
    L.check(lib.rh_score_launch_info(pc._h, linfo))
    out = {"workload": wl, "rows": int(linfo[0]), "lists": bool(linfo[1]), "subset_points": S, "groups": (S + 63) // 64, "candidates": {k: sum(1 for c in cands if c[0] == k) for k in KN},
           "global": {"blocks_with_tile": int(st[96]), "stagings": int(st[97]), "stagings_reused": int(st[98]), "blocks_tile_all_disabled": int(st[99])},
           "per_kind": {}, "census": {}}
    for k, name in enumerate(KN):
        out["per_kind"][name] = {f: int(st[24 * k + i]) for i, f in enumerate(FIELDS)}
        s10 = snd[10 * k:10 * k + 10]
        out["census"][name] = {"pairs_supertile_skips": int(snd[48 + k]), "supertile_violations": int(snd[52 + k]),"pairs": int(s10[0]), "pairs_box_skips": int(s10[1]), "violations": int(s10[2] + s10[6] + s10[7] + s10[9]),
                               "points": int(s10[3]), "surely_in": int(s10[4]), "surely_out": int(s10[5]), "exact_inliers": int(s10[8]),
                               "pairs_with_band_point": int(snd[40 + k]), "pairs_with_exact_inlier": int(snd[44 + k])}
    os.makedirs("gpurun_out", exist_ok=True)
    path = os.path.join("gpurun_out", "s4_stats_%s.json" % wl)
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["global"]))
    for name in KN:
        p, c = out["per_kind"][name], out["census"][name]
        if not c["pairs"]:
            continue
        print("%-8s pairs %10d  box-survive %6.2f%%  band %6.2f%%  exact-inlier %6.2f%% | visits %d tested %d batches %d second-pass %d (%.2f%% of pairs) undecided pts %d drains %d+%d pairs_with_count %d"
              % (name, c["pairs"], 100 * (1 - c["pairs_box_skips"] / c["pairs"]), 100 * c["pairs_with_band_point"] / c["pairs"],
                 100 * c["pairs_with_exact_inlier"] / c["pairs"], p["visits_h0"] + p["visits_h1"] + p["visits_h2"] + p["visits_h3p"],
                 p["candidates_box_tested"], p["batches"], p["second_pass_pairs"], 100 * p["second_pass_pairs"] / max(1, p["surviving_pairs"]),
                 p["undecided_points"], p["ring_drains_full"], p["ring_drains_final"], p["pairs_with_count"]))
    print("wrote", path)


if __name__ == "__main__":
    main()
