#!/usr/bin/env python3
"""rh_cloud_create alone: two clouds of the same scene in one process (the first pays the library's one-time costs), with the
stage times of RH_CREATE_PROF=1 on stderr.   python tools/cloud_create_time.py [cfg3|cfg5]"""
import os, sys, time
_DIAG_ENV = [k for k in os.environ if k.startswith("RH_") and k not in ("RH_LIB_VARIANT", "RH_EXTRA_FLAGS", "RH_TYPES", "RH_SYSTEM_HIP") and not k.startswith("RH_BENCH")]
if _DIAG_ENV:   # RH_* switches exist in the diag build only (the product library reads no environment variable)
    os.environ.setdefault("RH_LIB_VARIANT", "diag")
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()                       # the HIP context exists before the library is asked for anything (like in bench.py)
torch.zeros(1, device="cuda")
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth

w = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
n, seed, scanner = 10_000_000, 3, None
if w == "cfg5":
    prim, n, seed, scanner = prim + ["cone"] * 8, 50_000_000, 5, [synth.BOX / 2] * 3
xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=seed, scanner=scanner)
subs = synth.make_subsets(n, 32, seed=seed)
for k in range(2):
    t0 = time.perf_counter()
    pc = R.RANSACCloud(xyz, nrm, subs)
    t = time.perf_counter() - t0
    ms = (C.c_double * 4)()
    L.check(R.lib().rh_cloud_create_ms(pc._h, ms))
    print("%s cloud %d: RANSACCloud() %.1f ms; rh_cloud_create %.1f ms (subset order %.1f, before %.1f, after %.1f)" % (w, k, 1e3 * t, ms[0], ms[1], ms[2], ms[3]), flush=True)
    del pc
