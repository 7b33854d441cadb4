#!/usr/bin/env python3
"""The refit scan alone on the bench's cloud (cfg3, or WL=cfg5): every ground-truth primitive through rh_refit, scan time
from HIP events (rh_last_refit_ms) per kind, with the culled scan (default from 2^21 points on) or RH_REFIT_PATH=scan.
RH_KREFIT_DBG=1 prints how many groups survive the box test."""
import ctypes as C, os, sys
_DIAG_ENV = [k for k in os.environ if k.startswith("RH_") and k not in ("RH_LIB_VARIANT", "RH_EXTRA_FLAGS", "RH_TYPES", "RH_SYSTEM_HIP") and not k.startswith("RH_BENCH")]
if _DIAG_ENV:   # RH_* switches exist in the diag build only (the product library reads no environment variable)
    os.environ.setdefault("RH_LIB_VARIANT", "diag")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
import bench

wl = os.environ.get("WL", "cfg3")
prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
n, seed, scanner = 10_000_000, 3, None
if wl == "cfg5":
    prim += ["cone"] * 8; types += [R.FittedCone]; n, seed, scanner = 50_000_000, 5, [synth.BOX / 2] * 3
xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=seed, scanner=scanner)
subs = synth.make_subsets(n, 32, seed=seed)
f32 = bool(os.environ.get("F32"))
pc = R.RANSACCloud(xyz.astype(np.float32), nrm.astype(np.float32), subs, force_eltype=np.float32) if f32 else R.RANSACCloud(xyz, nrm, subs)
cp = R.params_to_c(R.ransacparameters(types), score_mode=L.SCORE_F64)
cands = synth.jittered_candidates(truth, len(truth), seed=0, jitter=0.0)
arr = bench.shapes_to_c(R, L, [(k, o, v) for (k, o, v) in cands])
lib = R.lib()
idx = np.zeros(n, dtype=np.int64); nout = C.c_int64()
per = {}
for rep in range(int(os.environ.get("REPS", "3"))):
    for i in range(len(truth)):
        cs = arr[i]
        if f32: lib.rh_shape_finalize_f32(C.byref(cs))
        L.check(lib.rh_refit(pc._h, C.byref(cs), C.byref(cp), idx.ctypes.data_as(C.POINTER(C.c_int64)), n, C.byref(nout)))
        a, b = C.c_float(), C.c_float(); L.check(lib.rh_last_refit_ms(pc._h, C.byref(a), C.byref(b)))
        if rep: per.setdefault(cs.kind, []).append((a.value, b.value, nout.value))
for k, v in sorted(per.items()):
    v = np.array(v)
    print("kind %d: scan %.4f ms (min %.4f max %.4f), compaction %.4f ms, mean inliers %.0f" % (k, v[:, 0].mean(), v[:, 0].min(), v[:, 0].max(), v[:, 1].mean(), v[:, 2].mean()))
