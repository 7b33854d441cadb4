"""End-to-end rh_ransac on the cfg3 cloud only (the bench's end_to_end leg without the rest), for
`rocprofv3 --kernel-trace --stats -- python tools/prof_e2e.py [itermax] [runs] [root|octree] [cfg3|cfg5]` and
RH_DRIVER_PROF=1."""
import os
_DIAG_ENV = [k for k in os.environ if k.startswith("RH_") and k not in ("RH_LIB_VARIANT", "RH_EXTRA_FLAGS", "RH_TYPES", "RH_SYSTEM_HIP") and not k.startswith("RH_BENCH")]
if _DIAG_ENV:   # RH_* switches exist in the diag build only (the product library reads no environment variable)
    os.environ.setdefault("RH_LIB_VARIANT", "diag")
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ransac_jl_amd as R
from ransac_jl_amd import _lib as L
from ransac_jl_amd import synth

itermax = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
octree = len(sys.argv) > 3 and sys.argv[3] == "octree"
cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg3"
c = synth.config(cfg)
n = c["xyz"].shape[0]
subs = synth.make_subsets(n, c["r"], c["seed"])
pc = R.RANSACCloud(c["xyz"], c["nrm"], subs)
types = [R.FittedPlane, R.FittedSphere, R.FittedCylinder] + ([R.FittedCone] if cfg == "cfg5" else [])
if os.environ.get("RH_TYPES"):   # ablations: e.g. RH_TYPES=p or RH_TYPES=ps
    types = [{"p": R.FittedPlane, "s": R.FittedSphere, "c": R.FittedCylinder, "k": R.FittedCone}[ch] for ch in os.environ["RH_TYPES"]]
p = R.ransacparameters(types, iteration={"minsubsetN": 4096, "itermax": itermax, "τ": 900, "prob_det": 0.9})
cp = R.params_to_c(p, score_mode=L.SCORE_F64, sphere_uses_enabled=True, sampling_streams=1, octree_sampling=octree)
cp.itermax = 4
R.ransac(pc, cp, seed=99)
cp.itermax = itermax
for r in range(runs):
    pc.enable_all()
    t0 = time.perf_counter()
    got, secs, st = R.ransac(pc, cp, seed=1234, return_stats=True)
    dt = time.perf_counter() - t0
    print("run %d: %d shapes in %.4f s (%.0f shapes/s), rh_ransac %.4f s, sample_fit %.4f extract %.4f" %
          (r, len(got), dt, len(got) / dt, st["seconds"], st["seconds_host"], st["seconds_extract"]), flush=True)
