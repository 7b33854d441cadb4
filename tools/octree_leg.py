#!/usr/bin/env python3
"""The bench's octree-sampling end-to-end leg alone (cfg3 cloud, minsubsetN = 4096, ITERS iterations, default 256): three timed
rh_ransac calls after a warm-up, with the driver's own breakdown (RH_DRIVER_PROF=1 prints it on stderr).  For rocprofv3
--kernel-trace --stats runs of the leg and A/B runs of the window logic.   ITERS=256 python tools/octree_leg.py"""
import os, sys, time
_DIAG_ENV = [k for k in os.environ if k.startswith("RH_") and k not in ("RH_LIB_VARIANT", "RH_LIB_PATH", "RH_EXTRA_FLAGS", "RH_TYPES", "RH_SYSTEM_HIP") and not k.startswith("RH_BENCH")]
if _DIAG_ENV:   # RH_* switches exist in the diag build only (the product library reads no environment variable)
    os.environ.setdefault("RH_LIB_VARIANT", "diag")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth

prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
n = int(os.environ.get("POINTS", 10_000_000))
xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=3)
subs = synth.make_subsets(n, 32, seed=3)
pc = R.RANSACCloud(xyz, nrm, subs)
if os.environ.get("LISTS") in ("on", "off"):   # super-tile lists for the windows' score launches (rh_set_option st_cull)
    R.set_option("st_cull", 1 if os.environ["LISTS"] == "on" else 2, cloud=pc)
iters = int(os.environ.get("ITERS", "256"))
e2e = R.ransacparameters(types, iteration={"minsubsetN": 4096, "itermax": iters, "τ": 900, "prob_det": 0.9})
ocp = R.params_to_c(e2e, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1, octree_sampling=not os.environ.get("ROOT_CELL"))
ocp.itermax = 4
R.ransac(pc, ocp, seed=99)
ocp.itermax = iters
import hashlib
for r in range(int(os.environ.get("RUNS", "3"))):
    pc.enable_all()
    t0 = time.perf_counter()
    got, _, st = R.ransac(pc, ocp, seed=1234, return_stats=True)
    t = time.perf_counter() - t0
    print("run %d: %.4f s, %d shapes, %d iterations, %d candidates scored, sample_fit %.4f score %.4f extract %.4f" %
          (r, t, len(got), st["iterations"], st["candidates_scored"], st["seconds_host"], st["seconds_score"], st["seconds_extract"]), flush=True)
    h = hashlib.sha256()
    for g in got:
        h.update(np.asarray(g.inpoints, dtype=np.int64).tobytes())
    print("   result digest", h.hexdigest()[:16], [(g.iteration, len(g.inpoints)) for g in got][:6], "rng draws", st.get("draws"), "scored_left", st.get("scored_left"), flush=True)
if os.environ.get("OCT_TIMING"):   # only with a library built with RH_EXTRA_FLAGS=-DRH_OCT_TIMING (ransac.jl_amd/build.py): phase stamps of the sampler
    import ctypes as C
    if not hasattr(R.lib(), "rh_dbg_oct_timing"):
        raise SystemExit("OCT_TIMING needs a library built with RH_EXTRA_FLAGS=-DRH_OCT_TIMING")
    t = (C.c_ulonglong * 16)()
    R.lib().rh_dbg_oct_timing.argtypes = [C.POINTER(C.c_ulonglong)]
    R.lib().rh_dbg_oct_timing(t)
    n = max(1, t[8])
    print("octree sampler phases, mean per thread (us; 100 MHz clock):", ["%.2f" % (t[i] / n / 100.0) for i in range(6)], "threads", t[8])
