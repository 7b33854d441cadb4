"""Multi-GPU scoring: one process per GPU (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm).  Candidates are independent (fitting.jl:182-186), so a batch shards
embarrassingly: every rank holds a replica of subset 1 in its own HBM (15 MB at 10M points /
32 subsets), scores a contiguous slice of the candidate batch, and ONE collective per batch --
an int32[B] sum all-reduce of the zero-padded per-candidate counts -- gives every rank every
score.  Integer sums are order-independent, so the result is bit-identical to the 1-GPU run.
The payload is 4*B bytes (16 KB at B = 4096): latency-bound, link bandwidth is irrelevant."""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import check, lib


def shard_bounds(b, rank, world):
    """Contiguous slice [lo, hi) of a b-candidate batch owned by `rank`; sizes differ by <= 1."""
    base, rem = divmod(b, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_counts(counts_full, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts_full, op=dist.ReduceOp.SUM, group=group)
    return counts_full


def score_batch_sharded(b, rank, world, local_score, counts_full, group=None, same_stream=False):
    """counts_full: int32 tensor [b] on the rank's device.  local_score(lo, hi, out_view) must fill
    out_view (= counts_full[lo:hi]) with the counts of candidates lo..hi-1.
    same_stream: the cloud was put on torch's current stream (RANSACCloud.set_stream) and local_score
    does not wait -- fill, scoring and collective are then ordered by the stream alone, with no host
    synchronisation inside a step."""
    lo, hi = shard_bounds(b, rank, world)
    counts_full.zero_()
    if counts_full.is_cuda and not same_stream:
        # the fill runs on torch's stream, the scoring on the library's: order them
        import torch
        torch.cuda.current_stream(counts_full.device).synchronize()
    if hi > lo:
        local_score(lo, hi, counts_full[lo:hi])
    return allreduce_counts(counts_full, group)


class DeviceBatch:
    """A candidate batch resident in HBM next to a cloud, for the *_dev entry points."""

    def __init__(self, pc, shapes_ctypes_array, b):
        self.pc, self.b = pc, b
        self.d_shapes = C.c_void_p()
        nbytes = C.sizeof(L.Shape) * max(1, b)
        check(lib().rh_dev_alloc(pc._h, nbytes, C.byref(self.d_shapes)))
        check(lib().rh_dev_upload(pc._h, self.d_shapes, C.cast(shapes_ctypes_array, C.c_void_p), C.sizeof(L.Shape) * b))

    def slice_ptr(self, lo):
        return C.c_void_p(self.d_shapes.value + lo * C.sizeof(L.Shape))

    def free(self):
        if self.d_shapes:
            lib().rh_dev_free(self.pc._h, self.d_shapes)
            self.d_shapes = None


def gpu_local_score(pc, batch, cparams, wait=True):
    """local_score callback: scores candidates lo..hi-1 of `batch` on pc's device, writing
    straight into the torch tensor view (device pointer); wait=True waits for the cloud's stream
    (needed unless the cloud shares torch's stream, see score_batch_sharded)."""
    def fn(lo, hi, out_view):
        assert out_view.is_cuda and out_view.dtype.itemsize == 4 and out_view.is_contiguous()
        check(lib().rh_score_batch_dev(pc._h, batch.slice_ptr(lo), hi - lo, C.byref(cparams),
                                       C.c_void_p(out_view.data_ptr()), None))
        if wait:
            check(lib().rh_cloud_sync(pc._h))
    return fn
