"""The multi-GPU step behind the C ABI (rh_comm_create / rh_score_batch_allreduce_dev, include/ransac_hip.h): the
library reaches RCCL itself.  On the one GPU of the test box: a ONE-rank communicator (a real ncclCommInitRank and a
real ncclAllReduce on the communicator's stream) must give exactly what rh_score_batch_dev gives, for a slice with an
offset into a larger zero-padded count buffer as well.  Two ranks cannot share one GPU under RCCL (it refuses duplicate
devices), so the step meets its second and third rank over a test-only stand-in for librccl's five entry points
(tests/native/fake_rccl.cpp: stream-ordered copies + a host callback that sums the ranks' data in POSIX shared memory),
bound through RH_RCCL_LIB: offsets, the prepare launch's zeroing of the whole total, the plain-fill path of small and
empty slices, the two-buffer rotation and the stream-ordered fence, across processes.  The real RCCL path at N > 1 is
the driver's multi-GPU run (bench.py --collective lib): unmeasured here."""
import ctypes as C

import numpy as np
import pytest

import ransac_jl_amd as R
from ransac_jl_amd import _lib as L
from ransac_jl_amd import dist as rdist, synth

pytestmark = pytest.mark.gpu

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    so = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"), "-o", so, "-lrt"])
    return so


@pytest.mark.diag      # (RH_RCCL_LIB, the hook that binds the stand-in, exists in the diag build only)
@pytest.mark.parametrize("world", [2, 3])
def test_library_collective_across_processes_sharing_gpu0(world, fake_rccl, tmp_path):
    """world ranks, one process each, all on GPU 0: every rank scores its slice of 18 batches (pairs and bursts of four in flight) through
    rh_score_batch_allreduce_dev and must end up with EVERY candidate's count, equal to plain rh_score_batch_dev."""
    env = dict(os.environ, RH_RCCL_LIB=fake_rccl, HSA_ENABLE_IPC_MODE_LEGACY="0")
    idfile = str(tmp_path / "id.bin")
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "native", "comm_worker.py"), str(r), str(world), idfile,
                                       str(tmp_path / ("out%d.txt" % r))], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:           # exactly the processes started above
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
        line = open(tmp_path / ("out%d.txt" % r)).read().split()
        assert line[0] == "ok" and int(line[1]) == 10 and int(line[2]) > 10000


def test_one_rank_library_collective_equals_plain_scoring():
    import torch
    prim = ["plane", "plane", "sphere", "cylinder", "cone"]
    xyz, nrm, truth = synth.make_cloud(120_000, prim, 0.2, seed=21)
    subs = synth.make_subsets(120_000, 4, seed=21)
    pc = R.RANSACCloud(xyz, nrm, subs)
    cp = R.params_to_c(R.ransacparameters([R.FittedPlane, R.FittedSphere, R.FittedCylinder, R.FittedCone]))
    import bench
    cands = synth.jittered_candidates(truth, 700, seed=4)
    arr = bench.shapes_to_c(R, L, cands)
    batch = rdist.DeviceBatch(pc, arr, 700)
    lib = R.lib()
    ref = torch.zeros(700, dtype=torch.int32, device="cuda")
    L.check(lib.rh_score_batch_dev(pc._h, batch.slice_ptr(0), 700, C.byref(cp), C.c_void_p(ref.data_ptr()), None))
    L.check(lib.rh_cloud_sync(pc._h))
    ref = ref.cpu().numpy()
    assert ref.sum() > 10000
    comm = rdist.LibComm(pc, 0, 1)
    # the whole batch
    out = torch.full((700,), 77, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # (the fill runs on torch's stream, the library on the cloud's)
    comm.score_allreduce(batch.slice_ptr(0), 700, 0, 700, cp, out.data_ptr())
    comm.sync()
    assert np.array_equal(out.cpu().numpy(), ref)
    # a slice at an offset inside a larger buffer (what rank r of N does): the rest stays zero; twice in a row with two
    # buffers in flight and a stream-ordered fence instead of a host wait
    bufs = [torch.full((1000,), 5, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for i, (lo, hi) in enumerate([(100, 350), (350, 700)]):
        comm.score_allreduce(batch.slice_ptr(lo), hi - lo, 200 + lo, 1000, cp, bufs[i].data_ptr())
    comm.fence()
    L.check(lib.rh_cloud_sync(pc._h))
    for i, (lo, hi) in enumerate([(100, 350), (350, 700)]):
        got = bufs[i].cpu().numpy()
        exp = np.zeros(1000, dtype=np.int32)
        exp[200 + lo:200 + hi] = ref[lo:hi]
        assert np.array_equal(got, exp)
    # many batches with two buffers in flight and no wait in between (a buffer comes back two calls later: the library
    # orders that call behind the collective that read it), slices of every size -- 32 candidates and fewer take the
    # plain fill, larger ones have their total zeroed by the prepare launch -- and a rank without candidates
    ring = [torch.full((900,), 9, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    slices = [(0, 5), (5, 37), (37, 38), (38, 300), (300, 300), (300, 699), (699, 700), (0, 700)]
    for i, (lo, hi) in enumerate(slices):
        comm.score_allreduce(batch.slice_ptr(lo), hi - lo, 100 + lo, 900, cp, ring[i & 1].data_ptr())
        if i >= len(slices) - 2:
            continue
        # check the one that is two calls old before its buffer is handed out again
        if i >= 1:
            comm.fence()
            L.check(lib.rh_cloud_sync(pc._h))
            plo, phi = slices[i - 1]
            exp = np.zeros(900, dtype=np.int32)
            exp[100 + plo:100 + phi] = ref[plo:phi]
            assert np.array_equal(ring[(i - 1) & 1].cpu().numpy(), exp), (plo, phi)
    comm.sync()
    for i in (len(slices) - 2, len(slices) - 1):
        plo, phi = slices[i]
        exp = np.zeros(900, dtype=np.int32)
        exp[100 + plo:100 + phi] = ref[plo:phi]
        assert np.array_equal(ring[i & 1].cpu().numpy(), exp), (plo, phi)
    # bad arguments fail loudly
    with pytest.raises(R.RansacHipError):
        comm.score_allreduce(batch.slice_ptr(0), 700, 400, 1000, cp, bufs[0].data_ptr())    # slice beyond the buffer
    comm.close()
