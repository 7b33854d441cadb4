import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, dist as rdist, synth
import bench
prim = ["plane"] * 16 + ["sphere"] * 12 + ["cylinder"] * 12 + ["cone"] * 8
types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]
n = 50_000_000
xyz, nrm, truth = synth.make_cloud(n, prim, 0.30, seed=5, scanner=[synth.BOX / 2] * 3)
subs = synth.make_subsets(n, 32, seed=5)
pc = R.RANSACCloud(xyz, nrm, subs)
cp = R.params_to_c(R.ransacparameters(types), score_mode=L.SCORE_F64)
cands = synth.jittered_candidates(truth, 4096, seed=0)
arr = bench.shapes_to_c(R, L, cands)
batch = rdist.DeviceBatch(pc, arr, 4096)
S = subs[0].size; sw = (S + 63) // 64
counts = torch.zeros(4096, dtype=torch.int32, device="cuda")
dmask = torch.empty(4096 * sw, dtype=torch.int64, device="cuda")
lib = R.lib()
def step():
    L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), 4096, C.byref(cp), C.c_void_p(counts.data_ptr()), C.c_void_p(dmask.data_ptr())))
for _ in range(5): step()
L.check(lib.rh_cloud_sync(pc._h))
L.check(lib.rh_timer_start(pc._h))
for _ in range(20): step()
ms = C.c_float(); L.check(lib.rh_timer_stop(pc._h, C.byref(ms)))
print("cfg5 masks ms_per_step %.4f" % (ms.value / 20))
