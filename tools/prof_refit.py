"""Micro-harness: refit scan on a 10M-point random cloud (for rocprofv3 runs)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ransac_jl_amd as R
n = int(os.environ.get("N", 10_000_000))
rng = np.random.default_rng(0)
xyz = rng.uniform(0, 100, size=(n, 3)); nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
nrm[: n // 50] = [0, 0, 1.0]; xyz[: n // 50, 2] = 50.0
perm = rng.permutation(n); xyz = np.ascontiguousarray(xyz[perm]); nrm = np.ascontiguousarray(nrm[perm])
pc = R.RANSACCloud(xyz, nrm, [np.arange(1, n // 32 + 1, dtype=np.int64)])
cp = R.params_to_c(R.ransacparameters())
shapes = [R.FittedPlane([50, 50, 50.0], [0, 0, 1.0]), R.FittedSphere([50, 50, 50.0], 20.0, True),
          R.FittedCylinder([0, 0, 1.0], [50, 50, 50.0], 10.0, True), R.FittedCone([50, 50, 0.0], [0, 0, 1.0], 0.8, True)]
for s in shapes:
    for _ in range(5):
        t = time.perf_counter(); ex = R.refit(s, pc, cp); dt = time.perf_counter() - t
    print(R.strt(s), len(ex.inpoints), "host wall ms %.3f" % (dt * 1e3))
