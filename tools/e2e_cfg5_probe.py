#!/usr/bin/env python3
"""The bounded cfg5 end-to-end leg of bench.py alone (50M points, cones, itermax 4096), several rh_ransac calls in a row
with the driver's own breakdown (RH_DRIVER_PROF=1): what the first calls on a cloud pay that later ones do not."""
import os, sys, time
_DIAG_ENV = [k for k in os.environ if k.startswith("RH_") and k not in ("RH_LIB_VARIANT", "RH_EXTRA_FLAGS", "RH_TYPES", "RH_SYSTEM_HIP") and not k.startswith("RH_BENCH")]
if _DIAG_ENV:   # RH_* switches exist in the diag build only (the product library reads no environment variable)
    os.environ.setdefault("RH_LIB_VARIANT", "diag")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12 + ["cone"] * 8
types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]
n = int(os.environ.get("POINTS", 50_000_000))
xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=5, scanner=[synth.BOX / 2] * 3)
subs = synth.make_subsets(n, 32, seed=5)
pc = R.RANSACCloud(xyz, nrm, subs)
e2e = R.ransacparameters(types, iteration={"minsubsetN": 4096, "itermax": int(os.environ.get("ITERS", "4096")), "τ": 900, "prob_det": 0.9})
cp = R.params_to_c(e2e, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1)
for r in range(int(os.environ.get("RUNS", "5"))):
    pc.enable_all()
    t0 = time.perf_counter()
    got, _, st = R.ransac(pc, cp, seed=1234, return_stats=True)
    t = time.perf_counter() - t0
    print("run %d: %.4f s wall, %.4f s in rh_ransac, %d shapes, sample_fit %.4f extract %.4f" % (r, t, st["seconds"], len(got), st["seconds_host"], st["seconds_extract"]), flush=True)
    if os.environ.get("DROP"):
        del got
