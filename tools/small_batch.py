#!/usr/bin/env python3
"""Floor of the score launch for tiny batches (cfg3 cloud): ms per rh_score_batch_dev call."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, dist as rdist, synth
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
batch = rdist.DeviceBatch(pc, arr, 4096)
counts = torch.zeros(4096, dtype=torch.int32, device="cuda")
for b in (1, 3, 8, 32, 64, 256, 1024, 4096):
    for path in ("groups", "brute"):
        for _ in range(3):
            L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), b, C.byref(cp), C.c_void_p(counts.data_ptr()), None))
        L.check(lib.rh_cloud_sync(pc._h))
        ms = C.c_float()
        L.check(lib.rh_timer_start(pc._h))
        for _ in range(50):
            L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), b, C.byref(cp), C.c_void_p(counts.data_ptr()), None))
        L.check(lib.rh_timer_stop(pc._h, C.byref(ms)))
        print("b=%5d  %.4f ms/call (merged culled kernel, back-to-back)" % (b, ms.value / 50))
        break
