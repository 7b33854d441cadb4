#!/usr/bin/env python3
"""Latency of the drop-in call the reference's loop makes every iteration: rh_score_batch with HOST buffers
(upload, score, read back, one wait) for batches of the size scorecandidates! sees (<= 60 at the defaults)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth
from bench import shapes_to_c
lib = R.lib()
n = int(os.environ.get("N", 10_000_000))
prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=3)
subs = synth.make_subsets(n, 32, seed=3)
pc = R.RANSACCloud(xyz, nrm, subs, device=0)
cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder]), score_mode=L.SCORE_F64)
cands = synth.jittered_candidates(truth, 4096, seed=0)
arr = shapes_to_c(R, L, cands)
counts = (C.c_int32 * 4096)()
for b in (1, 8, 15, 60, 256, 1024, 4096):
    for _ in range(20):
        L.check(lib.rh_score_batch(pc._h, arr, b, C.byref(cp), counts, None))
    reps = 300 if b <= 256 else 100
    t0 = time.perf_counter()
    for _ in range(reps):
        L.check(lib.rh_score_batch(pc._h, arr, b, C.byref(cp), counts, None))
    dt = (time.perf_counter() - t0) / reps
    print("b=%5d  %.1f us per rh_score_batch call (host buffers in and out), checksum %d" % (b, 1e6 * dt, sum(counts[:b])), flush=True)

# the enabled gather of samplepointcloud4! (fitting.jl:405-422): k-th enabled points, one call per minimal set
import numpy as np
total = pc.count_enabled()
rng = np.random.default_rng(1)
for k in (2, 64, 4096):
    ranks = np.ascontiguousarray(rng.integers(1, total + 1, size=k), dtype=np.int64)
    out = np.zeros(k, dtype=np.int64)
    rp, op = ranks.ctypes.data_as(C.POINTER(C.c_int64)), out.ctypes.data_as(C.POINTER(C.c_int64))
    for _ in range(20):
        L.check(lib.rh_select_enabled(pc._h, rp, k, op))
    t0 = time.perf_counter()
    for _ in range(300):
        L.check(lib.rh_select_enabled(pc._h, rp, k, op))
    dt = (time.perf_counter() - t0) / 300
    print("k=%5d  %.1f us per rh_select_enabled call, checksum %d" % (k, 1e6 * dt, int(out.sum() % 1000003)), flush=True)
