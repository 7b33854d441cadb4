#!/usr/bin/env python3
"""The bench's timed step alone (cfg3 cloud, 4096 jittered candidates, rh_score_batch_dev) for kernel A/B runs under
rocprofv3: prints ms per step from HIP events.  RH_* switches are read by the library."""
import ctypes as C, os, sys, time
_DIAG_ENV = [k for k in os.environ if k.startswith("RH_") and k not in ("RH_LIB_VARIANT", "RH_EXTRA_FLAGS", "RH_TYPES", "RH_SYSTEM_HIP") and not k.startswith("RH_BENCH")]
if _DIAG_ENV:   # RH_* switches exist in the diag build only (the product library reads no environment variable)
    os.environ.setdefault("RH_LIB_VARIANT", "diag")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, dist as rdist, synth
import bench

wl = os.environ.get("WL", "cfg3")
prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder]
n, seed, scanner = 10_000_000, 3, None
if wl == "cfg2":
    prim = ["plane", "plane", "sphere", "sphere", "cylinder", "cylinder"]; n, seed = 1_000_000, 2
if wl == "cfg5":
    prim += ["cone"] * 8; types += [R.FittedCone]; n, seed, scanner = 50_000_000, 5, [synth.BOX / 2] * 3
xyz, nrm, truth = synth.make_cloud(n, prim, 0.0 if wl == "cfg2" else 0.30, seed=seed, scanner=scanner)
subs = synth.make_subsets(n, 32, seed=seed)
f32 = bool(os.environ.get("F32"))   # a Float32 cloud (binary32 exact tests; shapes rounded to binary32)
pc = R.RANSACCloud(xyz.astype(np.float32), nrm.astype(np.float32), subs, force_eltype=np.float32) if f32 else R.RANSACCloud(xyz, nrm, subs)
cp = R.params_to_c(R.ransacparameters(types), score_mode=L.SCORE_F64)
cands = synth.jittered_candidates(truth, 4096, seed=0)
if os.environ.get("KINDS"):   # e.g. KINDS=plane or KINDS=sphere,cylinder: the other candidates of the batch are dropped (per-region counters)
    keep = set(os.environ["KINDS"].split(","))
    cands = [c for c in cands if c[0] in keep]
NB = len(cands)
arr = bench.shapes_to_c(R, L, cands)
if f32:
    for i in range(NB):
        R.lib().rh_shape_finalize_f32(C.byref(arr[i]))
batch = rdist.DeviceBatch(pc, arr, NB)
counts = torch.zeros(NB, dtype=torch.int32, device="cuda")
lib = R.lib()
def step():
    L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), NB, C.byref(cp), C.c_void_p(counts.data_ptr()), None))
for _ in range(int(os.environ.get("PRE", "200"))): step()
L.check(lib.rh_cloud_sync(pc._h))
steps = int(os.environ.get("STEPS", "100"))
L.check(lib.rh_timer_start(pc._h))
for _ in range(steps): step()
ms = C.c_float(); L.check(lib.rh_timer_stop(pc._h, C.byref(ms)))
print("ms_per_step %.4f  sum(counts) %d" % (ms.value / steps, int(counts.sum().item())))
if os.environ.get("REFIT"):
    t = truth[0]
    sh = R.FittedPlane(t["point"], t["normal"])
    cs = R.shape_f32(sh) if f32 else sh.to_c()
    idx = np.zeros(n, dtype=np.int64); nout = C.c_int64(); acc = []
    for _ in range(6):
        L.check(lib.rh_refit(pc._h, C.byref(cs), C.byref(cp), idx.ctypes.data_as(C.POINTER(C.c_int64)), n, C.byref(nout)))
        a, b = C.c_float(), C.c_float(); L.check(lib.rh_last_refit_ms(pc._h, C.byref(a), C.byref(b))); acc.append(a.value)
    print("refit scan ms %.4f (%d inliers), %.0f GB/s on %s bytes/point" % (min(acc[1:]), nout.value, n * (24.125 if f32 else 48.125) / min(acc[1:]) / 1e6, "24.125" if f32 else "48.125"))
