#!/usr/bin/env python3
"""Worst binary32 error of the score kernel's classifier in units of its margin width (rh_dbg_cls_audit), on synthetic
scenes at several coordinate scales with jittered-truth and arbitrary candidates.  Sound below 0.5; expected below 0.25
(the margins carry a safety factor of 2).  python tools/cls_audit.py [cases] [seed]"""
import ctypes as C, os, sys
if __name__ == "__main__":
    os.environ.setdefault("RH_LIB_VARIANT", "diag")   # the A/B switches / rh_dbg_* audits live in the diag build (libransac_hip_diag.so)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth


def audit(pc, arr, b, cp):
    out = np.zeros(12)
    L.check(R.lib().rh_dbg_cls_audit(pc._h, arr, b, C.byref(cp), out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def one(case, rng):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import fuzz_score
    scale = float(rng.choice([1.0, 100.0, 100.0, 1e4]))
    n = int(rng.choice([20_000, 65_536, 150_001]))
    names = list(rng.choice(["plane", "sphere", "cylinder"], size=int(rng.integers(2, 6))))
    xyz, nrm, truth = synth.make_cloud(n, names, float(rng.choice([0.0, 0.2, 0.5])), seed=9000 + case)
    xyz = xyz * (scale / 100.0)
    for t in truth:
        for k in ("point", "center", "apex"):
            if k in t: t[k] = np.asarray(t[k]) * (scale / 100.0)
        if "radius" in t: t["radius"] = t["radius"] * (scale / 100.0)
    subs = synth.make_subsets(n, 2, seed=case)
    pc = R.RANSACCloud(xyz, nrm, subs)
    params = R.ransacparameters()
    for k in ("plane", "sphere", "cylinder", "cone"):
        params[k]["ϵ"] = float(rng.choice([1e-3, 0.05, 0.3, 2.0])) * (scale / 100.0)
        params[k]["α"] = float(np.radians(rng.choice([0.5, 5.0, 30.0, 80.0])))
    cp = R.params_to_c(params, score_mode=L.SCORE_F64)
    b = 96
    arr = (L.Shape * b)(*[fuzz_score.rand_shape(rng, truth, scale) for _ in range(b)])
    return audit(pc, arr, b, cp), "n=%d scale=%g" % (n, scale)


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    worst = np.zeros(6)
    pairs = np.zeros(3)
    for case in range(ncases):
        o, desc = one(case, rng)
        worst = np.maximum(worst, o[:6])
        pairs += o[8:11]
        print("case %3d %-22s plane %.4f %.4f  sphere %.4f %.4f  cylinder %.4f %.4f" % ((case, desc) + tuple(o[:6])), flush=True)
    print("worst |x32 - x64| / margin width: plane a %.4f b %.4f | sphere a %.4f b %.4f | cylinder a %.4f b %.4f  (sound < 0.5)" % tuple(worst))
    print("pairs audited: plane %.3g sphere %.3g cylinder %.3g" % tuple(pairs))
    sys.exit(0 if worst.max() < 0.5 else 1)


if __name__ == "__main__":
    main()
