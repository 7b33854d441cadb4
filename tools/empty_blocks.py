#!/usr/bin/env python3
"""What blocks without work cost the culled score launch: a batch of 4096 planes far outside the cloud (every super-tile box
rules every candidate out), scored with the super-tile lists on -- every block of the score launch returns after reading its
super-tile's four list lengths -- and off (every block stages its tile and walks its row's chunks to find no pair).
    WL=cfg3|cfg5 python tools/empty_blocks.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L, synth

wl = os.environ.get("WL", "cfg3")
c = synth.config(wl)
subs = synth.make_subsets(len(c["xyz"]), c["r"], c["seed"])
pc = R.RANSACCloud(c["xyz"], c["nrm"], subs)
cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder]))
b = 4096
rng = np.random.default_rng(0)
shapes = []
for i in range(b):
    nv = rng.normal(size=3); nv /= np.linalg.norm(nv)
    shapes.append(R.FittedPlane(nv * 1e5, nv))        # a plane 1e5 away from the origin
arr = (L.Shape * b)(*[s.to_c() for s in shapes])
lib = R.lib()
d_sh, d_cn = C.c_void_p(), C.c_void_p()
L.check(lib.rh_dev_alloc(pc._h, C.sizeof(L.Shape) * b, C.byref(d_sh)))
L.check(lib.rh_dev_alloc(pc._h, 4 * b, C.byref(d_cn)))
L.check(lib.rh_dev_upload(pc._h, d_sh, C.cast(arr, C.c_void_p), C.sizeof(L.Shape) * b))
for mode, name in ((2, "lists off"), (1, "lists on")):
    R.set_option("st_cull", mode, cloud=pc)
    acc, lst = 0.0, 0.0
    msk = (C.c_float * 5)()
    for rep in range(40):
        msk[0] = -1.0
        L.check(lib.rh_score_batch_dev_timed(pc._h, d_sh, b, C.byref(cp), d_cn, None, msk))
        lm = C.c_float()
        L.check(lib.rh_last_list_launch_ms(pc._h, C.byref(lm)))
        if rep >= 10:
            acc += msk[4]; lst += lm.value
    info = (C.c_int32 * 4)()
    L.check(lib.rh_score_launch_info(pc._h, info))
    cn = np.zeros(b, dtype=np.int32)
    L.check(lib.rh_dev_download(pc._h, cn.ctypes.data_as(C.c_void_p), d_cn, 4 * b))
    print("%s %-9s score launch %.4f ms (R = %d, %d rows x %d tiles = %d blocks), list launch %.4f ms, inliers %d"
          % (wl, name, acc / 30, info[0], info[2], info[3], info[2] * info[3], lst / 30, int(cn.sum())), flush=True)
