"""One rank of tests/test_comm_gpu.py's multi-rank runs (a process per rank, all on GPU 0): the library's own collective
step, rh_score_batch_allreduce_dev, against plain scoring.  argv: rank world idfile outfile"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ransac_jl_amd as R  # noqa: E402
from ransac_jl_amd import _lib as L  # noqa: E402
from ransac_jl_amd import dist as rdist, synth  # noqa: E402

rank, world, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.cuda.set_device(0)


def exchange(raw):
    """rank 0's 128 bytes to everybody, through a file"""
    if rank == 0:
        with open(idfile + ".tmp", "wb") as f:
            f.write(raw)
        os.replace(idfile + ".tmp", idfile)
        return raw
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise SystemExit("rank %d: no id file" % rank)
        time.sleep(0.01)
    return open(idfile, "rb").read()


prim = ["plane", "plane", "sphere", "cylinder", "cone"]
xyz, nrm, truth = synth.make_cloud(120_000, prim, 0.2, seed=21)
subs = synth.make_subsets(120_000, 4, seed=21)
pc = R.RANSACCloud(xyz, nrm, subs)
cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]))
import bench  # noqa: E402

B = 701
cands = synth.jittered_candidates(truth, B, seed=4)
arr = bench.shapes_to_c(R, L, cands)
batch = rdist.DeviceBatch(pc, arr, B)
lib = R.lib()
ref = torch.zeros(B, dtype=torch.int32, device="cuda")
L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), B, C.byref(cp), C.c_void_p(ref.data_ptr()), None))
L.check(lib.rh_cloud_sync(pc._h))
ref = ref.cpu().numpy()
assert ref.sum() > 10000
comm = rdist.LibComm(pc, rank, world, exchange=exchange)

# every batch: the candidates [lo, hi) of the reference batch, cut into `world` slices; rank r scores slice r into its
# place of a zero-padded total of `tot` counts that starts `pad` entries in.  Slices of every size: more than 32
# candidates (the total is zeroed by the prepare launch), 32 and fewer (plain fill), none at all.
rng = np.random.default_rng(99)        # the same plan on every rank
plans = []
for it in range(18):
    lo = int(rng.integers(0, 300))
    hi = int(rng.integers(lo, B + 1)) if it % 5 else lo + int(rng.integers(0, 40))
    cuts = sorted(int(x) for x in rng.integers(lo, hi + 1, size=world - 1))
    if it == 3:
        cuts = [lo] * (world - 1)       # everything on the last rank, the others empty
    pad = int(rng.integers(0, 50))
    plans.append((lo, hi, [lo] + cuts + [hi], pad, hi - lo + pad + int(rng.integers(0, 50))))
bufs = [torch.full((1000,), 13, dtype=torch.int32, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
checked = 0


def check(i):
    lo, hi, edges, pad, tot = plans[i]
    exp = np.zeros(tot, dtype=np.int32)
    exp[pad:pad + hi - lo] = ref[lo:hi]
    got = bufs[i & 1][:tot].cpu().numpy()
    assert np.array_equal(got, exp), "rank %d, batch %d (%d..%d, edges %s): %d counts differ" % (rank, i, lo, hi, edges, int((got != exp).sum()))


def enqueue(i):
    lo, hi, edges, pad, tot = plans[i]
    a, b = edges[rank], edges[rank + 1]
    comm.score_allreduce(batch.slice_ptr(a), b - a, pad + (a - lo), tot, cp, bufs[i & 1].data_ptr())


# two batches in flight, both verified (stream-ordered fence, then a wait on the cloud's stream only)
enqueue(0); enqueue(1)
comm.fence()
L.check(lib.rh_cloud_sync(pc._h))
check(0); check(1)
checked += 2
# bursts of four with no wait of any kind in between: batch k + 2 takes the buffer batch k's collective is still reading
# and writing -- the library orders it behind that collective (rh_comm's done[] events); had it not, the late sum of
# batch k would land on top of batch k + 2's counts.  The last two of every burst are verified.
for k in range(2, len(plans), 4):
    for i in range(k, k + 4):
        enqueue(i)
    if (k // 4) & 1:
        comm.sync()                        # the host wait on the collectives' stream ...
    else:
        comm.fence()                       # ... or the stream-ordered fence
        L.check(lib.rh_cloud_sync(pc._h))
    check(k + 2); check(k + 3)
    checked += 2
comm.close()
with open(outfile, "w") as f:
    f.write("ok %d %d\n" % (checked, int(ref.sum())))
