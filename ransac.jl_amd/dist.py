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


class ShardedScorer:
    """score_batch_sharded as a two-deep software pipeline: batch i's all-reduce (asynchronous, on the
    collective's own stream) runs while batch i + 1 is being scored.  The collective is tiny (4*B bytes)
    but latency-bound over xGMI, so hiding it is what keeps weak scaling flat.  submit() enqueues one
    batch and returns its ticket; result(ticket) waits for that batch's collective and returns the
    int32[B] tensor (valid until `depth` more batches have been submitted); drain() waits for all."""

    def __init__(self, b, rank, world, local_score, device, group=None, same_stream=False, depth=2, points=False):
        """points=True: the point-sharded partitioning -- local_score fills the WHOLE buffer with this
        rank's partial counts and the all-reduce is a true sum (score_batch_point_sharded)."""
        import torch
        self.b, self.rank, self.world, self.local_score = b, rank, world, local_score
        self.group, self.same_stream, self.depth = group, same_stream, depth
        self.bufs = [torch.zeros(max(1, b), dtype=torch.int32, device=device)[:b] for _ in range(depth)]
        self.works = [None] * depth
        self.k = 0
        self.lo, self.hi = (0, b) if points else shard_bounds(b, rank, world)

    def submit(self):
        import torch
        import torch.distributed as dist
        i = self.k % self.depth
        if self.works[i] is not None:       # the buffer's previous collective (two batches ago)
            self.works[i].wait()
            self.works[i] = None
        buf = self.bufs[i]
        buf.zero_()
        if buf.is_cuda and not self.same_stream:
            torch.cuda.current_stream(buf.device).synchronize()
        if self.hi > self.lo:
            self.local_score(self.lo, self.hi, buf[self.lo:self.hi])
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self.works[i] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.k += 1
        return self.k - 1

    def result(self, ticket):
        assert self.k - self.depth <= ticket < self.k, "that batch's buffer has been reused"
        i = ticket % self.depth
        if self.works[i] is not None:
            self.works[i].wait()
            self.works[i] = None
        return self.bufs[i]

    def drain(self):
        for i in range(self.depth):
            if self.works[i] is not None:
                self.works[i].wait()
                self.works[i] = None


# ---- the other partitioning: points sharded, every rank scores every candidate ----------------------
#   Rank g holds a contiguous 1/G slice of subset 1 (scoring) and a contiguous 1/G slice
# of the cloud in original order (refit).  The collective of the score step is then a TRUE reduction:
# int32[B] partial inlier counts summed over ranks (bit-exact in any order); refit lists concatenate
# in rank order and stay ascending.  Candidate sharding (above) is the default because subset 1 is
# small (15 MB at cfg3) and a replica per GPU costs nothing; point sharding is what scales the
# HBM-bound refit scan and the per-GPU state.
def point_shard_subset(vertices, normals, subset1, enabled_bool, rank, world):
    """This rank's slice of subset 1 as a self-contained (xyz, nrm, subset, enabled) quadruple: a cloud
    of just those points, all of them subset 1, with their enabled bits gathered."""
    lo, hi = shard_bounds(subset1.size, rank, world)
    idx0 = subset1[lo:hi] - 1
    xyz = np.ascontiguousarray(vertices[idx0])
    nrm = np.ascontiguousarray(normals[idx0])
    en = None if enabled_bool is None else np.ascontiguousarray(enabled_bool[idx0])
    return xyz, nrm, np.arange(1, hi - lo + 1, dtype=np.int64), en


def score_batch_point_sharded(local_score_all, counts_full, group=None):
    """local_score_all(out) fills out (= counts_full, int32[B]) with this rank's PARTIAL counts of all B
    candidates over its point slice; the sum all-reduce makes them the full counts on every rank."""
    local_score_all(counts_full)
    return allreduce_counts(counts_full, group)


def refit_point_sharded(local_indices_1based, offset, device, group=None):
    """local_indices_1based: ascending inlier indices within this rank's cloud slice (which starts at
    original 0-based index `offset`).  Returns the ascending list over the whole cloud on every rank:
    all-gather of the lengths, then of the zero-padded lists."""
    import torch
    import torch.distributed as dist
    mine = torch.as_tensor(np.asarray(local_indices_1based, dtype=np.int64) + int(offset), device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mine.cpu().numpy()
    world = dist.get_world_size(group)
    n_mine = torch.tensor([mine.numel()], dtype=torch.int64, device=device)
    lens = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(lens, n_mine, group=group)
    lens = [int(x.item()) for x in lens]
    cap = max(max(lens), 1)
    padded = torch.zeros(cap, dtype=torch.int64, device=device)
    padded[: mine.numel()] = mine
    parts = [torch.zeros(cap, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return np.concatenate([p[:n].cpu().numpy() for p, n in zip(parts, lens)])


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


class LibComm:
    """The multi-GPU step inside the C ABI (rh_comm_*, include/ransac_hip.h): librccl is reached by the library itself,
    the host program only carries the 128-byte id from rank 0 to the others (here: through torch.distributed's
    default group if one is up, else `exchange`, a callable bytes -> bytes that broadcasts rank 0's value)."""

    def __init__(self, pc, rank, world, exchange=None):
        self.pc, self.rank, self.world = pc, rank, world
        ident = (C.c_ubyte * 128)()
        if rank == 0:
            check(lib().rh_comm_unique_id(ident))
        raw = bytes(ident)
        if world > 1:
            if exchange is not None:
                raw = exchange(raw)
            else:
                import torch
                import torch.distributed as dist
                t = torch.tensor(list(raw), dtype=torch.uint8)
                if dist.get_backend() == "nccl":
                    t = t.cuda()
                dist.broadcast(t, src=0)
                raw = bytes(t.cpu().tolist())
        ident = (C.c_ubyte * 128).from_buffer_copy(raw)
        self._h = C.c_void_p()
        check(lib().rh_comm_create(pc._h, rank, world, ident, C.byref(self._h)))

    def score_allreduce(self, d_shapes_ptr, b, offset, b_total, cparams, d_counts_total_ptr):
        """enqueue: zero the b_total counts, score this rank's b candidates into [offset, offset + b), all-reduce (sum)"""
        sp = d_shapes_ptr if isinstance(d_shapes_ptr, C.c_void_p) else C.c_void_p(d_shapes_ptr)
        check(lib().rh_score_batch_allreduce_dev(self.pc._h, self._h, sp, b, offset, b_total, C.byref(cparams),
                                                 C.c_void_p(d_counts_total_ptr)))

    def fence(self):
        check(lib().rh_comm_fence(self._h, self.pc._h))

    def sync(self):
        check(lib().rh_comm_sync(self._h))

    def close(self):
        if self._h:
            lib().rh_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
