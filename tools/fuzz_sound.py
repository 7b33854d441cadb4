#!/usr/bin/env python3
"""Soundness of the v4 score kernel's shortcuts as a COUNT (rh_dbg_cls_soundness, csrc/score4.hip): over random clouds,
coordinate scales 1 .. 1e6, eps 1e-6 .. 50, alpha 0.5 .. 120 degrees, candidates from jittered ground truth to
arbitrary / degenerate / far-away-point shapes (tools/fuzz_score.py's generator), Float64 and Float32 clouds --
  (i)   (candidate, group) pairs the binary32 box test skips although the exact test finds an inlier in the group,
  (ii)  points the binary32 classifier calls surely-in that the exact test rejects,
  (iii) points it calls surely-out that the exact test accepts,
  (iv)  candidates whose classifier would count the all-zero point a disabled point is staged as
must all be 0.  python tools/fuzz_sound.py [ncases] [seed]   (F32=1: Float32 clouds)"""
import os, sys, time
if __name__ == "__main__":
    os.environ.setdefault("RH_LIB_VARIANT", "diag")   # the A/B switches / rh_dbg_* audits live in the diag build (libransac_hip_diag.so)
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
import fuzz_score

KINDS = ("plane", "sphere", "cylinder", "cone")
FIELDS = ("pairs", "pairs_skipped", "V_skipped_with_inlier", "points", "sure_in", "sure_out", "V_sure_in_rejected",
          "V_sure_out_accepted", "exact_inliers", "V_zero_point_in")
VIOL = (2, 6, 7, 9)


def one(case, rng, f32=False):
    scale = float(rng.choice([1.0, 100.0, 100.0, 1e4, 1e6]))
    r = int(rng.choice([1, 2, 3]))
    n = int(rng.choice([8192, 8193, 20_000, 65_536, 150_001])) * r
    names = list(rng.choice(list(KINDS), size=int(rng.integers(1, 6))))
    xyz, nrm, truth = synth.make_cloud(n, names, float(rng.choice([0.0, 0.2, 0.5])), seed=7000 + case)
    xyz = xyz * (scale / 100.0)
    for t in truth:
        for k in ("point", "center", "apex"):
            if k in t: t[k] = np.asarray(t[k]) * (scale / 100.0)
        if "radius" in t: t["radius"] = t["radius"] * (scale / 100.0)
    subs = synth.make_subsets(n, r, seed=case)
    if f32:
        xyz, nrm = xyz.astype(np.float32), nrm.astype(np.float32)
        pc = R.RANSACCloud(xyz, nrm, subs, force_eltype=np.float32)
    else:
        pc = R.RANSACCloud(xyz, nrm, subs)
    params = R.ransacparameters()
    for k in KINDS:
        params[k]["ϵ"] = float(rng.choice([1e-6, 1e-4, 0.05, 0.3, 2.0, 50.0])) * (scale / 100.0)
        params[k]["α"] = float(np.radians(rng.choice([0.5, 5.0, 30.0, 89.0, 120.0])))
    cp = R.params_to_c(params, score_mode=L.SCORE_F64)
    b = int(rng.choice([64, 128, 300]))
    arr = (L.Shape * b)(*[fuzz_score.rand_shape(rng, truth, scale) for _ in range(b)])
    if f32:
        for i in range(b):
            R.lib().rh_shape_finalize_f32(C.byref(arr[i]))
    out = np.zeros(56, dtype=np.uint64)
    L.check(R.lib().rh_dbg_cls_soundness(pc._h, arr, b, C.byref(cp), out.ctypes.data_as(C.POINTER(C.c_uint64))))
    st_skipped, st_viol = out[48:52].astype(np.int64), out[52:56].astype(np.int64)   # the super-tile boxes (st_cull)
    out = out[:40].reshape(4, 10).astype(np.int64)
    ok = int(out[:, VIOL].sum()) == 0 and int(st_viol.sum()) == 0
    ST_TOT[0] += st_skipped; ST_TOT[1] += st_viol
    return ok, "n=%d r=%d scale=%g b=%d f32=%d viol=%s st_viol=%s" % (n, r, scale, b, f32, out[:, VIOL].tolist(), st_viol.tolist()), out


ST_TOT = [np.zeros(4, dtype=np.int64), np.zeros(4, dtype=np.int64)]


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
    f32 = bool(os.environ.get("F32"))
    tot = np.zeros((4, 10), dtype=np.int64)
    bad, t0 = 0, time.time()
    for case in range(ncases):
        ok, desc, out = one(case, rng, f32=f32)
        tot += out
        print("%s case %3d  %s" % ("ok  " if ok else "FAIL", case, desc), flush=True)
        bad += not ok
    print("%d cases (%s clouds), %d failures, %.0f s" % (ncases, "Float32" if f32 else "Float64", bad, time.time() - t0))
    for k, name in enumerate(KINDS):
        d = dict(zip(FIELDS, tot[k].tolist()))
        dec = (d["sure_in"] + d["sure_out"]) / max(1, d["points"])
        print("%-8s %s  decided=%.4f skipped=%.4f" % (name, d, dec, d["pairs_skipped"] / max(1, d["pairs"])))
    print("super-tile boxes: pairs ruled out per kind %s, of them with an exact inlier (violations) %s" % (ST_TOT[0].tolist(), ST_TOT[1].tolist()))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
