#!/usr/bin/env python3
"""Repeated cloud creation / scoring / rh_ransac: device and host memory must stay flat."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth

xyz, nrm, truth = synth.make_cloud(60_000, ["plane", "sphere", "cylinder", "cone"], 0.2, seed=3)
subs = synth.make_subsets(60_000, 2, seed=3)
params = R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone],
                            iteration={"minsubsetN": 64, "itermax": 64, "τ": 300, "prob_det": 0.7})
shapes = [R.FittedPlane(truth[0]["point"], truth[0]["normal"])] * 50


def snapshot():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024

for rnd in range(6):
    for i in range(50):
        pc = R.RANSACCloud(xyz, nrm, subs)
        R.score_batch(pc, shapes, params, want_masks=(i % 2 == 0))
        for mode in (0, 1):
            got, _ = R.ransac(pc, params, seed=i, sampling_streams=mode, octree_sampling=bool(mode and i % 3 == 0))
        if i % 10 == 0:   # a Float32 cloud's loop, the reference octree and its device gather as well
            pc32 = R.RANSACCloud(xyz, nrm, subs, force_eltype=np.float32)
            p32 = R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder], iteration={"minsubsetN": 64, "itermax": 32, "τ": 300, "prob_det": 0.7})
            R.ransac(pc32, p32, seed=i, sampling_streams=1)
            R.cell_enabled_points(pc, pc.octree.children[0])
            del pc32
        del pc, got
    dev, host = snapshot()
    print("after %3d clouds / %3d ransac calls: device %.0f MiB in use, host max RSS %.0f MiB" % ((rnd + 1) * 50, (rnd + 1) * 100, dev, host), flush=True)
