// comm.hip -- the multi-GPU step of the hot path behind the C ABI: candidates are independent
// (/root/reference/src/fitting.jl:181-190), so every rank (one process per GPU, each with a replica of the cloud)
// scores a slice of the batch and ONE all-reduce (sum, int32) of the zero-padded counts gives every rank every score.
// RCCL is reached directly (librccl, loaded on first use -- a process that already holds one, e.g. PyTorch's, shares
// it), the collective runs on the communicator's own stream behind an event of the cloud's stream, so that batch i's
// all-reduce overlaps batch i + 1's scoring (the caller alternates two count buffers, like dist.ShardedScorer).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>

#include "rh_internal.h"

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r.h ? &r : nullptr;
    tried = true;
    // RH_RCCL_LIB: a library to bind instead (tests/native/fake_rccl.cpp: the same five entry points over host shared
    // memory, so that the step can meet a second rank on a one-GPU box -- RCCL refuses two ranks on one device)
#ifdef RH_DIAG
    const char *forced = rh_opt_env_string("RH_RCCL_LIB");   // (diag build only: the product library binds librccl and nothing else)
    if (forced && forced[0]) {
        r.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!r.h) return nullptr;
    }
#endif
    const char *names[] = { "librccl.so.1", "librccl.so" };
    for (const char *n : names)   // one already in the process (PyTorch's) first
        if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char *n : names)
        if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) return nullptr;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) { r.h = nullptr; return nullptr; }
    return &r;
}

}  // namespace

struct rh_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = -1;
    hipStream_t stream = nullptr;     // the collectives' stream
    hipEvent_t scored = nullptr;      // cloud stream -> collective stream
    hipEvent_t reduced = nullptr;     // collective stream -> whoever waits (rh_comm_fence / rh_comm_sync)
    hipEvent_t done[2] = { nullptr, nullptr };   // the last two collectives, alternately: call k waits for call k - 2 before it
    unsigned long long calls = 0;                //   touches its count buffer (two batches in flight, buffers reused two calls later)
};

#define RH_NCCL(call)                                                                                      \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) {                                                                           \
            rh_set_error("%s failed: %s", #call, R->GetErrorString(r_));                                   \
            return RH_E_NODEVICE;                                                                          \
        }                                                                                                  \
    } while (0)

extern "C" int rh_comm_unique_id(void *id_out)
{
    if (!id_out) { rh_set_error("rh_comm_unique_id: NULL argument"); return RH_E_INVALID; }
    Rccl *R = rccl();
    if (!R) { rh_set_error("rh_comm_unique_id: librccl could not be loaded (%s)", dlerror()); return RH_E_NODEVICE; }
    ncclUniqueId id;
    RH_NCCL(R->GetUniqueId(&id));
    static_assert(sizeof id == RH_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id_out, &id, sizeof id);
    return RH_OK;
}

extern "C" int rh_comm_create(rh_cloud *c, int32_t rank, int32_t world, const void *unique_id, rh_comm **out)
{
    if (!c || !unique_id || !out || world < 1 || rank < 0 || rank >= world) { rh_set_error("rh_comm_create: bad arguments"); return RH_E_INVALID; }
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) { rh_set_error("rh_comm_create: librccl could not be loaded (%s)", dlerror()); return RH_E_NODEVICE; }
    RH_HIP(hipSetDevice(c->device));
    rh_comm *m = new rh_comm;
    m->rank = rank; m->world = world; m->device = c->device;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = R->CommInitRank(&m->comm, world, id, rank);
    if (r != ncclSuccess) { rh_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, R->GetErrorString(r)); delete m; return RH_E_NODEVICE; }
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->scored, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->reduced, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->done[1], hipEventDisableTiming) != hipSuccess) {
        rh_set_error("rh_comm_create: stream / event creation failed");
        (void)R->CommDestroy(m->comm);
        delete m;
        return RH_E_NODEVICE;
    }
    *out = m;
    return RH_OK;
}

extern "C" int rh_comm_destroy(rh_comm *m)
{
    if (!m) return RH_OK;
    Rccl *R = rccl();
    if (m->device >= 0) (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (R && m->comm) (void)R->CommDestroy(m->comm);
    if (m->scored) (void)hipEventDestroy(m->scored);
    if (m->reduced) (void)hipEventDestroy(m->reduced);
    for (hipEvent_t e : m->done) if (e) (void)hipEventDestroy(e);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return RH_OK;
}

// scorecandidates! for this rank's slice of a batch that all ranks share, and the all-reduce that completes it:
// d_counts_total[b_total] is zeroed, the b candidates of d_shapes are scored into [offset, offset + b), and the sum over
// the ranks lands in d_counts_total on every rank.  Everything is enqueued; the counts are valid after rh_comm_fence
// (stream order) or rh_comm_sync (host).
extern "C" int rh_score_batch_allreduce_dev(rh_cloud *c, rh_comm *m, const rh_shape *d_shapes, int32_t b, int32_t offset, int32_t b_total,
                                            const rh_params *p, int32_t *d_counts_total)
{
    if (!c || !m || !p || !d_counts_total || b < 0 || offset < 0 || b_total < 0 || (int64_t)offset + b > b_total || (b > 0 && !d_shapes)) {
        rh_set_error("rh_score_batch_allreduce_dev: bad arguments");
        return RH_E_INVALID;
    }
    if (m->device != c->device) { rh_set_error("rh_score_batch_allreduce_dev: communicator and cloud are on different devices"); return RH_E_INVALID; }
    Rccl *R = rccl();
    if (!R) { rh_set_error("librccl is not loaded"); return RH_E_NODEVICE; }
    RH_HIP(hipSetDevice(c->device));
    RH_TRY(rh_join_batches(c));
    if (b_total == 0) return RH_OK;
    // a caller that keeps two batches in flight alternates two count buffers: this call's buffer was last read by the
    // collective of two calls ago, which the cloud's stream lets finish first (stream order, no host wait)
    // (asked on the host first: it is nearly always long finished, and a wait packet on the stream costs ~9 us of gap)
    if (m->calls >= 2 && hipEventQuery(m->done[m->calls & 1]) != hipSuccess) RH_HIP(hipStreamWaitEvent(c->stream, m->done[m->calls & 1], 0));
    // the whole total is zeroed by the batch's prepare launch (no fill launch of its own); a rank without candidates, or a
    // batch that does not go through that launch (<= 32 candidates travel staged), fills it the plain way
    bool zeroed = false;
    if (b > 32) {
        c->zero_extra = d_counts_total;
        c->zero_extra_n = b_total;
    } else {
        RH_HIP(hipMemsetAsync(d_counts_total, 0, sizeof(int32_t) * (size_t)b_total, c->stream));
        zeroed = true;
    }
    if (b > 0) {
        const int rc = rh_score_batch_dev(c, d_shapes, b, p, d_counts_total + offset, nullptr);
        const bool consumed = c->zero_extra == nullptr;
        c->zero_extra = nullptr; c->zero_extra_n = 0;
        if (rc != RH_OK) return rc;
        if (!zeroed && !consumed) { rh_set_error("rh_score_batch_allreduce_dev: the batch did not pass the prepare launch"); return RH_E_INTERNAL; }
    }
    RH_HIP(hipEventRecord(m->scored, c->stream));
    RH_HIP(hipStreamWaitEvent(m->stream, m->scored, 0));
    RH_NCCL(R->AllReduce(d_counts_total, d_counts_total, (size_t)b_total, ncclInt32, ncclSum, m->comm, m->stream));
    RH_HIP(hipEventRecord(m->reduced, m->stream));
    RH_HIP(hipEventRecord(m->done[m->calls & 1], m->stream));
    m->calls++;
    return RH_OK;
}

// the cloud's stream waits for the collectives enqueued so far (no host synchronisation)
extern "C" int rh_comm_fence(rh_comm *m, rh_cloud *c)
{
    if (!m || !c) { rh_set_error("rh_comm_fence: NULL argument"); return RH_E_INVALID; }
    RH_HIP(hipStreamWaitEvent(c->stream, m->reduced, 0));
    return RH_OK;
}

extern "C" int rh_comm_sync(rh_comm *m)
{
    if (!m) { rh_set_error("rh_comm_sync: NULL argument"); return RH_E_INVALID; }
    RH_HIP(hipStreamSynchronize(m->stream));
    return RH_OK;
}
