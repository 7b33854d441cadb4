import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, dist as rdist, synth
import bench
prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12
xyz, nrm, truth = synth.make_cloud(10_000_000, prim, 0.30, seed=3)
subs = synth.make_subsets(10_000_000, 32, seed=3)
pc = R.RANSACCloud(xyz, nrm, subs)
cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder]), score_mode=L.SCORE_F64)
cands = synth.jittered_candidates(truth, 4096, seed=0)
arr = bench.shapes_to_c(R, L, cands)
batch = rdist.DeviceBatch(pc, arr, 4096)
lib = R.lib()
comm = rdist.LibComm(pc, 0, 1)
bufs = [torch.zeros(4096, dtype=torch.int32, device="cuda") for _ in range(2)]
def run(kind, n=300):
    torch.cuda.synchronize(); L.check(lib.rh_cloud_sync(pc._h)); comm.sync()
    t0 = time.perf_counter()
    for i in range(n):
        b = bufs[i & 1]
        if kind == "lib":
            comm.score_allreduce(batch.slice_ptr(0), 4096, 0, 4096, cp, b.data_ptr())
        else:
            L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), 4096, C.byref(cp), C.c_void_p(b.data_ptr()), None))
    t1 = time.perf_counter()
    comm.sync(); L.check(lib.rh_cloud_sync(pc._h)); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-6s host enqueue %.1f us/step, total %.1f us/step" % (kind, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
for k in ("plain", "lib", "plain", "lib"):
    run(k)
