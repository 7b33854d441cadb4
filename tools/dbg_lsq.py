import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ransac_jl_amd as R
from ransac_jl_amd import synth
from oracle import oracle as orc
prim = ["plane", "sphere", "cylinder", "cone"]
xyz, nrm, truth = synth.make_cloud(20000, prim, 0.1, seed=1)
subs = synth.make_subsets(20000, 2, 1)
pc = R.RANSACCloud(xyz, nrm, subs); oc = orc.Cloud(xyz, nrm, subs[0])
cp = R.params_to_c(R.ransacparameters()); op = orc.Params.from_buffer_copy(bytes(cp))
for name, outw, v in synth.jittered_candidates(truth, 4, seed=3):
    cls = {"plane": R.FittedPlane, "sphere": R.FittedSphere, "cylinder": R.FittedCylinder, "cone": R.FittedCone}[name]
    s = cls(v[0:3], v[3:6]) if name == "plane" else (cls(v[0:3], v[3], True) if name == "sphere" else cls(v[0:3], v[3:6], v[6], True))
    try:
        g = R.refit_lsq(s, pc, cp, 12)
        print(name, "GPU", g)
    except Exception as e:
        print(name, "GPU ERR", e)
    e = oc.refit_lsq(orc.Shape.from_buffer_copy(bytes(s.to_c())), op, 12)
    print(name, "ORC", list(e[0].v)[:7], e[1:])
