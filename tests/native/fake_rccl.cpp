// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl's five entry points that comm.hip binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclGetErrorString), so that
// rh_score_batch_allreduce_dev can meet a SECOND rank on a one-GPU box: RCCL itself refuses two ranks on one device.
// Selected by RH_RCCL_LIB=<path of this library> (comm.hip's loader); never shipped, never used by bench.py.
//
// ncclAllReduce(sum, int32) keeps RCCL's contract as far as the caller can tell: it returns at once and everything is
// ordered on the given stream -- a device-to-host copy of the send buffer, a host callback that publishes the rank's
// data in a POSIX shared-memory segment, waits for every rank of the communicator and sums them, and a host-to-device
// copy of the sum into the receive buffer.  The ranks are separate processes (they may share a GPU).
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime.h>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclDataType_t;   // ncclInt32 = 2
typedef int ncclRedOp_t;      // ncclSum = 0
struct FakeComm;
typedef FakeComm *ncclComm_t;

}

namespace {

constexpr int kMaxWorld = 8;
constexpr size_t kSlotInts = 1 << 20;   // 4 MiB per rank

struct Segment {
    volatile uint64_t magic;
    volatile int32_t world;
    volatile uint64_t arrived[kMaxWorld];   // sequence number of the last reduction rank r has published
    volatile uint64_t left[kMaxWorld];      // ... and of the last one it has finished reading
    int32_t slot[kMaxWorld][kSlotInts];
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

}  // namespace

struct FakeComm {
    Segment *seg = nullptr;
    char name[128] = { 0 };
    int rank = 0, world = 1;
    uint64_t seq = 0;            // reductions enqueued
    int32_t *h_in = nullptr, *h_out = nullptr;   // pinned staging, two buffers each (two reductions may be in flight)
    size_t cap = 0;
    int failed = 0;
};

namespace {

struct Job { FakeComm *m; uint64_t seq; size_t count; int32_t *in, *out; };

void exchange(void *p)
{
    Job *j = (Job *)p;
    FakeComm *m = j->m;
    Segment *s = m->seg;
    const double t0 = now_s();
    // my slot is free once every rank has finished reading the previous reduction
    for (int r = 0; r < m->world; r++)
        while (__atomic_load_n(&s->left[r], __ATOMIC_ACQUIRE) + 1 < j->seq) {
            if (now_s() - t0 > 60.0) { m->failed = 1; delete j; return; }
        }
    memcpy((void *)s->slot[m->rank], j->in, sizeof(int32_t) * j->count);
    __atomic_store_n(&s->arrived[m->rank], j->seq, __ATOMIC_RELEASE);
    for (size_t i = 0; i < j->count; i++) j->out[i] = 0;
    for (int r = 0; r < m->world; r++) {
        while (__atomic_load_n(&s->arrived[r], __ATOMIC_ACQUIRE) < j->seq) {
            if (now_s() - t0 > 60.0) { m->failed = 1; delete j; return; }
        }
        const int32_t *src = (const int32_t *)s->slot[r];
        for (size_t i = 0; i < j->count; i++) j->out[i] += src[i];
    }
    __atomic_store_n(&s->left[m->rank], j->seq, __ATOMIC_RELEASE);
    delete j;
}

}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    case ncclSystemError: return "fake rccl: shared memory";
    default: return "fake rccl: error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    static int counter = 0;
    snprintf(id->internal, sizeof id->internal, "/rh_fake_rccl_%d_%d_%ld", (int)getpid(), counter++, (long)time(nullptr));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId id, int rank)
{
    if (!comm || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
    FakeComm *m = new FakeComm;
    m->rank = rank; m->world = world;
    memcpy(m->name, id.internal, sizeof m->name);
    m->name[sizeof m->name - 1] = 0;
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        shm_unlink(m->name);
        fd = shm_open(m->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)sizeof(Segment)) != 0) { if (fd >= 0) close(fd); delete m; return ncclSystemError; }
    } else {
        for (;;) {
            fd = shm_open(m->name, O_RDWR, 0600);
            if (fd >= 0) {
                off_t len = lseek(fd, 0, SEEK_END);
                if (len >= (off_t)sizeof(Segment)) break;
                close(fd);
                fd = -1;
            }
            if (now_s() - t0 > 60.0) { delete m; return ncclSystemError; }
            usleep(1000);
        }
    }
    void *mem = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) { delete m; return ncclSystemError; }
    m->seg = (Segment *)mem;
    if (rank == 0) {
        m->seg->world = world;
        __atomic_store_n(&m->seg->magic, 0x52484641ULL, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&m->seg->magic, __ATOMIC_ACQUIRE) != 0x52484641ULL) {
            if (now_s() - t0 > 60.0) { munmap(mem, sizeof(Segment)); delete m; return ncclSystemError; }
            usleep(1000);
        }
        if (m->seg->world != world) { munmap(mem, sizeof(Segment)); delete m; return ncclInvalidArgument; }
    }
    // like the real thing, the call returns when every rank has joined
    __atomic_store_n(&m->seg->arrived[rank], 0, __ATOMIC_RELEASE);
    __atomic_store_n(&m->seg->left[rank], 0, __ATOMIC_RELEASE);
    *comm = m;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t m)
{
    if (!m) return ncclSuccess;
    if (m->h_in) (void)hipHostFree(m->h_in);
    if (m->h_out) (void)hipHostFree(m->h_out);
    if (m->seg) munmap((void *)m->seg, sizeof(Segment));
    if (m->rank == 0) shm_unlink(m->name);
    delete m;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t m, hipStream_t stream)
{
    if (!m || !send || !recv || dt != 2 || op != 0 || count > kSlotInts) return ncclInvalidArgument;
    if (m->failed) return ncclInternalError;
    if (count == 0) return ncclSuccess;
    if (m->cap < count) {
        // (grown only while nothing is in flight: the stream is drained first)
        if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
        if (m->h_in) (void)hipHostFree(m->h_in);
        if (m->h_out) (void)hipHostFree(m->h_out);
        m->h_in = m->h_out = nullptr;
        const size_t cap = count < 4096 ? 4096 : count;
        if (hipHostMalloc((void **)&m->h_in, sizeof(int32_t) * cap * 2) != hipSuccess ||
            hipHostMalloc((void **)&m->h_out, sizeof(int32_t) * cap * 2) != hipSuccess) return ncclUnhandledCudaError;
        m->cap = cap;
    }
    m->seq++;
    Job *j = new Job{ m, m->seq, count, m->h_in + (m->seq & 1) * m->cap, m->h_out + (m->seq & 1) * m->cap };
    if (hipMemcpyAsync(j->in, send, sizeof(int32_t) * count, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, exchange, j) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, j->out, sizeof(int32_t) * count, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}
