// driver.hip -- rh_ransac: the reference's ransac() loop (src/iterations.jl:35-162) with the
// data-parallel steps on the GPU.
//
// Host: score statistics, best-candidate bookkeeping, the replay of speculated iterations in order;
// with sampling_streams = 0 (the reference's single sequential stream, or an injected one) also RNG,
// minimal-set sampling (samplepointcloud4!, src/fitting.jl:383-430) and fits.  Device: sampling and
// fitting of whole windows of iterations (sampling_streams = 1, sampler.hip), batched scoring of
// every candidate a window / iteration produced (one launch instead of src/fitting.jl:181-190's
// sequential loop -- legal because nothing inside an iteration reads a score before
// iterations.jl:99 is done), the full-cloud refit scan, enabled-bit maintenance and candidate
// liveness.
//
// Candidate store.  The reference keeps every scored candidate with its inlier index list and
// deletes, at each extraction, every candidate that owns a now-disabled point
// (removeinvalidshapes!, src/fitting.jl:209-221).  Index lists do not scale to thousands of
// candidates per iteration, so liveness is RECOMPUTED: a stored candidate is invalid iff it is
// compatible with some subset-1 point that is disabled now and was part of its inlier list:
//   plane / cylinder / cone (inliers = compatible & enabled at score time): the points disabled
//     by THIS extraction suffice -- earlier extractions already removed their victims;
//   sphere in reference mode (inliers ignore the enabled bit, sphere.jl:121,131): every
//     disabled subset-1 point.
// Both are one more run of the score kernel over the append-only list of disabled subset-1
// points (rh_cloud::dis), new points at the tail.
//
// Octree.  In the reference the level argmax at src/fitting.jl:401 is always 1 because the
// RANSACCloud constructor swaps levelweight/levelscore (src/octree.jl:44-45 vs :84; SURVEY.md
// 0.5), so every minimal set is drawn from the root cell = all enabled points in ascending
// order, and levelscore never influences a result.  The driver therefore samples from the
// enabled set directly and keeps no octree.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "fit_shared.h"
#include "rh_internal.h"

namespace {

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

}  // namespace

// ---- rh_mp: the processes of ONE NODE that run rh_ransac_mp on the same scene (one process per GPU, every one
// with a replica of the cloud).  They exchange the candidate lists of their windows -- a few records per window --
// through a POSIX shared-memory segment: the payload is tiny and the exchange sits on the loop's critical path, so
// what matters is latency (a microsecond through host memory; a collective over the fabric costs tens).  Each rank
// owns two slots (window sequence number parity) and a flag per slot; publishing = write the slot, then store the
// sequence number with release semantics; collecting = wait for every rank's flag to reach the sequence number.
struct rh_mp {
    int rank = 0, world = 1;
    int64_t slot_bytes = 0;
    size_t map_bytes = 0;
    char *base = nullptr;
    uint64_t seq = 0;            // exchanges done so far (the same on every rank)
    char name[128];
};

namespace {

constexpr uint64_t RH_MP_MAGIC = 0x52484d5032303236ULL;   // "RHMP2026"
constexpr int RH_MP_MAX_WORLD = 64;
// magic, world, then from byte 256 on one 64-byte line per (rank, parity) flag -- for the largest group rh_mp_open accepts
// (with a 4096-byte header the flags of ranks >= 30 lay inside rank 0's first slot)
constexpr size_t RH_MP_HDR = (256 + 64 * 2 * (size_t)RH_MP_MAX_WORLD + 4095) / 4096 * 4096;
static_assert(256 + 64 * 2 * (size_t)RH_MP_MAX_WORLD <= RH_MP_HDR, "the flags must not reach into the slots");

inline volatile uint64_t *mp_flag(rh_mp *m, int rank, int parity) { return (volatile uint64_t *)(m->base + 256 + 64 * (size_t)(rank * 2 + parity)); }
inline char *mp_slot(rh_mp *m, int rank, int parity) { return m->base + RH_MP_HDR + (size_t)(rank * 2 + parity) * (size_t)m->slot_bytes; }

// every rank publishes `bytes` of payload; afterwards payload r of every rank r can be read with mp_slot(m, r, parity)
// until the exchange after the next one.  Returns the parity used.
int mp_exchange(rh_mp *m, const void *payload, int64_t bytes, int *parity_out)
{
    if (bytes > m->slot_bytes) {
        rh_set_error("rh_ransac_mp: a window's candidate list (%lld bytes) does not fit the exchange slot (%lld)", (long long)bytes,
                     (long long)m->slot_bytes);
        return RH_E_CAPACITY;
    }
    m->seq++;
    const int par = (int)(m->seq & 1);
    memcpy(mp_slot(m, m->rank, par), payload, (size_t)bytes);
    __atomic_store_n((uint64_t *)mp_flag(m, m->rank, par), m->seq, __ATOMIC_RELEASE);
    const double t0 = now_s();
    for (int r = 0; r < m->world; r++) {
        uint64_t spins = 0;
        while (__atomic_load_n((uint64_t *)mp_flag(m, r, par), __ATOMIC_ACQUIRE) < m->seq) {
            if (++spins > 2000) {
                sched_yield();
                if ((spins & 1023) == 0 && now_s() - t0 > 60.0) {
                    rh_set_error("rh_ransac_mp: rank %d did not reach exchange %llu within 60 s (rank %d waited)", r,
                                 (unsigned long long)m->seq, m->rank);
                    return RH_E_INTERNAL;
                }
            }
        }
    }
    *parity_out = par;
    return RH_OK;
}

// the same for payloads of any size (and different sizes per rank): the payload travels in pieces of the slot size;
// every piece carries the rank's total, so after the first round all ranks agree on the number of rounds.
// recv[r] = rank r's payload.
int mp_exchange_any(rh_mp *m, const void *payload, int64_t bytes, std::vector<std::vector<char>> &recv)
{
    const int64_t cap = m->slot_bytes - 16;
    recv.assign((size_t)m->world, std::vector<char>());
    std::vector<char> piece((size_t)m->slot_bytes);
    int64_t rounds = 1;
    for (int64_t r = 0; r < rounds; r++) {
        const int64_t off = std::min(bytes, r * cap), len = std::min(cap, bytes - off);
        memcpy(piece.data(), &bytes, 8);
        memcpy(piece.data() + 8, &len, 8);
        if (len > 0) memcpy(piece.data() + 16, (const char *)payload + off, (size_t)len);
        int par = 0;
        RH_TRY(mp_exchange(m, piece.data(), 16 + len, &par));
        for (int k = 0; k < m->world; k++) {
            int64_t tot = 0, ln = 0;
            memcpy(&tot, mp_slot(m, k, par), 8);
            memcpy(&ln, mp_slot(m, k, par) + 8, 8);
            if (tot < 0 || ln < 0 || ln > cap) { rh_set_error("rh_ransac_mp: corrupt exchange header from rank %d", k); return RH_E_INTERNAL; }
            if (r == 0) {
                recv[(size_t)k].reserve((size_t)tot);
                rounds = std::max(rounds, (tot + cap - 1) / cap);
            }
            recv[(size_t)k].insert(recv[(size_t)k].end(), mp_slot(m, k, par) + 16, mp_slot(m, k, par) + 16 + ln);
        }
    }
    return RH_OK;
}

}  // namespace

extern "C" int rh_mp_open(const char *name, int32_t rank, int32_t world, int64_t slot_bytes, rh_mp **out)
{
    if (!name || !out || world < 1 || rank < 0 || rank >= world || world > RH_MP_MAX_WORLD || strlen(name) >= 120) {
        rh_set_error("rh_mp_open: bad arguments");
        return RH_E_INVALID;
    }
    if (slot_bytes <= 0) slot_bytes = (int64_t)1 << 20;
    slot_bytes = (slot_bytes + 4095) / 4096 * 4096;
    rh_mp *m = new rh_mp;
    m->rank = rank; m->world = world; m->slot_bytes = slot_bytes;
    m->map_bytes = RH_MP_HDR + (size_t)world * 2 * (size_t)slot_bytes;
    snprintf(m->name, sizeof m->name, "%s", name);
    int fd = -1;
    const double t0 = now_s();
    if (rank == 0) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)m->map_bytes) != 0) {
            rh_set_error("rh_mp_open: cannot create shared memory %s (%lld bytes)", name, (long long)m->map_bytes);
            if (fd >= 0) close(fd);
            delete m;
            return RH_E_NOMEM;
        }
    } else {
        for (;;) {   // wait for rank 0 to create and size the segment
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= m->map_bytes) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (now_s() - t0 > 60.0) { rh_set_error("rh_mp_open: rank 0 did not create %s within 60 s", name); delete m; return RH_E_INTERNAL; }
            usleep(1000);
        }
    }
    void *mem = mmap(nullptr, m->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) { rh_set_error("rh_mp_open: mmap of %s failed", name); delete m; return RH_E_NOMEM; }
    m->base = (char *)mem;
    if (rank == 0) {   // a fresh segment is zero-filled: flags start at 0; publish the header last
        ((volatile int32_t *)(m->base + 8))[0] = world;
        __atomic_store_n((uint64_t *)m->base, RH_MP_MAGIC, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n((uint64_t *)m->base, __ATOMIC_ACQUIRE) != RH_MP_MAGIC) {
            if (now_s() - t0 > 60.0) { rh_set_error("rh_mp_open: %s was never initialised", name); munmap(mem, m->map_bytes); delete m; return RH_E_INTERNAL; }
            usleep(200);
        }
        if (((volatile int32_t *)(m->base + 8))[0] != world) {
            rh_set_error("rh_mp_open: %s was created for %d ranks, this is rank %d of %d", name, ((volatile int32_t *)(m->base + 8))[0], rank, world);
            munmap(mem, m->map_bytes);
            delete m;
            return RH_E_INVALID;
        }
    }
    // everybody has mapped the segment once this first exchange returns: the name can go (no leak if a rank dies later)
    int par = 0;
    const int32_t hello = rank;
    int rc = mp_exchange(m, &hello, sizeof hello, &par);
    if (rc != RH_OK) { munmap(mem, m->map_bytes); delete m; return rc; }
    if (rank == 0) (void)shm_unlink(name);
    *out = m;
    return RH_OK;
}

// the exchange on its own (host memory only, no GPU involved): every rank contributes `bytes` bytes (the same number on
// every rank), out receives world x bytes in rank order
extern "C" int rh_mp_allgather(rh_mp *m, const void *payload, int64_t bytes, void *out)
{
    if (!m || bytes < 0 || (bytes > 0 && (!payload || !out))) { rh_set_error("rh_mp_allgather: bad arguments"); return RH_E_INVALID; }
    std::vector<std::vector<char>> recv;   // (payloads beyond the slot size travel in pieces)
    RH_TRY(mp_exchange_any(m, payload, bytes, recv));
    for (int r = 0; r < m->world; r++) {
        if ((int64_t)recv[(size_t)r].size() != bytes) { rh_set_error("rh_mp_allgather: rank %d sent %lld bytes, expected %lld", r, (long long)recv[(size_t)r].size(), (long long)bytes); return RH_E_INVALID; }
        if (bytes > 0) memcpy((char *)out + (size_t)r * (size_t)bytes, recv[(size_t)r].data(), (size_t)bytes);
    }
    return RH_OK;
}

extern "C" int rh_mp_close(rh_mp *m)
{
    if (!m) return RH_OK;
    if (m->base) munmap(m->base, m->map_bytes);
    delete m;
    return RH_OK;
}

namespace {

// host mirror of pc.isenabled with a rank directory for "k-th enabled point"
struct EnabledMirror {
    std::vector<uint64_t> w;
    std::vector<int64_t> dir;   // enabled count before each 64-word block
    int64_t n = 0, count = 0;
    bool dir_ok = false;
    static constexpr int64_t BLK = 64;

    bool test(int64_t i0) const { return (w[(size_t)(i0 >> 6)] >> (i0 & 63)) & 1ULL; }
    void recount()
    {
        count = 0;
        for (uint64_t x : w) count += __builtin_popcountll(x);
        dir_ok = false;
    }
    void clear(const int64_t *idx1, int64_t k)
    {
        for (int64_t j = 0; j < k; j++) {
            const int64_t i0 = idx1[j] - 1;
            uint64_t &x = w[(size_t)(i0 >> 6)];
            const uint64_t bit = 1ULL << (i0 & 63);
            if (x & bit) { x &= ~bit; count--; }
        }
        dir_ok = false;
    }
    void build()
    {
        const int64_t nb = (int64_t)w.size() / BLK + 1;
        dir.assign((size_t)nb + 1, 0);
        int64_t acc = 0;
        for (int64_t b = 0; b < nb; b++) {
            dir[(size_t)b] = acc;
            const int64_t lo = b * BLK, hi = std::min<int64_t>(lo + BLK, (int64_t)w.size());
            for (int64_t i = lo; i < hi; i++) acc += __builtin_popcountll(w[(size_t)i]);
        }
        dir[(size_t)nb] = acc;
        dir_ok = true;
    }
    // 1-based rank -> 1-based index of the k-th enabled point (ascending)
    int64_t select(int64_t k)
    {
        if (!dir_ok) build();
        const int64_t nb = (int64_t)dir.size() - 1;
        if (k < 1 || k > dir[(size_t)nb]) return 0;
        int64_t lo = 0, hi = nb;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) / 2;
            if (dir[(size_t)mid] < k) lo = mid; else hi = mid;
        }
        int64_t rem = k - dir[(size_t)lo];
        for (int64_t i = lo * BLK; i < (int64_t)w.size(); i++) {
            const int pc = __builtin_popcountll(w[(size_t)i]);
            if (rem <= pc) {
                uint64_t x = w[(size_t)i];
                for (int64_t r = 1; r < rem; r++) x &= x - 1;
                return i * 64 + __builtin_ctzll(x) + 1;
            }
            rem -= pc;
        }
        return 0;
    }
};

constexpr int64_t LIVE_MAX = 4096;   // stores up to this size take the one-wait extraction path

// (the shape itself -- 80 bytes -- lives in Driver::shapes, append-only for the length of a run: the compaction after an
// extraction walks the whole store and moves 24-byte records instead of 96-byte ones)
struct Stored {
    double E;
    int32_t slot;    // index in the device store of its kind
    int32_t sigma;   // its count on subset 1
    int32_t shape;   // index into Driver::shapes
    int32_t kind;
};

// device-resident store of prepared candidates, one growable array per kind
struct DeviceStore {
    rh_prep *prep[4] = { nullptr, nullptr, nullptr, nullptr };
    rh_prep *spare[4] = { nullptr, nullptr, nullptr, nullptr };   // compaction target, same capacity
    int64_t spare_cap[4] = { 0, 0, 0, 0 };
    // device-managed mode (chained octree windows): the host's candidate number of every entry, the spare twin, and the
    // scratch of rhk_store_compact
    int32_t *id[4] = { nullptr, nullptr, nullptr, nullptr };
    int32_t *spare_id[4] = { nullptr, nullptr, nullptr, nullptr };
    double *Eb[4] = { nullptr, nullptr, nullptr, nullptr };        // the entries' scores (the compaction finds the best survivor)
    double *spare_E[4] = { nullptr, nullptr, nullptr, nullptr };
    int32_t *d_work = nullptr;
    int64_t work_cap = 0;
    void *d_cls = nullptr;        // classifier + culling records of the whole store for a liveness pass of the v4 kernel
    float *d_box = nullptr;       //   (made on the fly by rhk_store_cls), cls_cap entries
    int64_t cls_cap = 0;
    int64_t cap[4] = { 0, 0, 0, 0 };
    int32_t n[4] = { 0, 0, 0, 0 };
    int32_t *iota = nullptr;      // 0..iota_cap-1
    int64_t iota_cap = 0;
    int32_t *counts = nullptr;    // liveness / score counts, iota_cap entries
    int32_t *live = nullptr;      // LIVE_MAX liveness flags of the one-wait extraction path, zero between uses
    int32_t *d_idx = nullptr;     // gather lists
    int32_t *d_nk = nullptr;      // one int per launch slot (8)
};

int store_free(rh_cloud *c, DeviceStore &st)
{
    (void)hipStreamSynchronize(c->stream);
    for (int k = 0; k < 4; k++) { (void)hipFree(st.prep[k]); (void)hipFree(st.spare[k]); (void)hipFree(st.id[k]); (void)hipFree(st.spare_id[k]);
                                  (void)hipFree(st.Eb[k]); (void)hipFree(st.spare_E[k]); }
    (void)hipFree(st.d_work); (void)hipFree(st.d_cls); (void)hipFree(st.d_box);
    (void)hipFree(st.iota); (void)hipFree(st.counts); (void)hipFree(st.d_idx); (void)hipFree(st.d_nk);
    (void)hipFree(st.live);
    return RH_OK;
}

int store_reserve(rh_cloud *c, DeviceStore &st, int kind, int64_t need)
{
    if (need <= st.cap[kind]) {
        if (st.id[kind] == nullptr && st.cap[kind] > 0) {   // (a store parked by a run that kept no ids)
            RH_HIP(hipMalloc((void **)&st.id[kind], sizeof(int32_t) * (size_t)st.cap[kind]));
            RH_HIP(hipMalloc((void **)&st.Eb[kind], sizeof(double) * (size_t)st.cap[kind]));
        }
        return RH_OK;
    }
    const int64_t cap = std::max<int64_t>(need, std::max<int64_t>(4096, st.cap[kind] * 2));
    rh_prep *np = nullptr;
    int32_t *ni = nullptr;
    RH_HIP(hipMalloc((void **)&np, sizeof(rh_prep) * (size_t)cap));
    double *ne = nullptr;
    RH_HIP(hipMalloc((void **)&ni, sizeof(int32_t) * (size_t)cap));
    RH_HIP(hipMalloc((void **)&ne, sizeof(double) * (size_t)cap));
    if (st.n[kind] > 0) {
        RH_HIP(hipMemcpyAsync(np, st.prep[kind], sizeof(rh_prep) * (size_t)st.n[kind], hipMemcpyDeviceToDevice, c->stream));
        if (st.id[kind] != nullptr) {
            RH_HIP(hipMemcpyAsync(ni, st.id[kind], sizeof(int32_t) * (size_t)st.n[kind], hipMemcpyDeviceToDevice, c->stream));
            RH_HIP(hipMemcpyAsync(ne, st.Eb[kind], sizeof(double) * (size_t)st.n[kind], hipMemcpyDeviceToDevice, c->stream));
        }
    }
    RH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(st.prep[kind]);
    (void)hipFree(st.id[kind]);
    (void)hipFree(st.Eb[kind]);
    st.prep[kind] = np;
    st.id[kind] = ni;
    st.Eb[kind] = ne;
    st.cap[kind] = cap;
    return RH_OK;
}

int store_reserve_aux(rh_cloud *c, DeviceStore &st, int64_t need)
{
    if (need <= st.iota_cap) return RH_OK;
    const int64_t cap = std::max<int64_t>(need, std::max<int64_t>(4096, st.iota_cap * 2));
    RH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(st.iota); (void)hipFree(st.counts); (void)hipFree(st.d_idx);
    st.iota = st.counts = st.d_idx = nullptr;
    RH_HIP(hipMalloc((void **)&st.iota, sizeof(int32_t) * (size_t)cap));
    RH_HIP(hipMalloc((void **)&st.counts, sizeof(int32_t) * (size_t)cap));
    RH_HIP(hipMalloc((void **)&st.d_idx, sizeof(int32_t) * (size_t)cap));
    RH_TRY(rhk_iota(c, st.iota, (int32_t)cap, 0));
    st.iota_cap = cap;
    return RH_OK;
}

}  // namespace

// ---- result arenas ---------------------------------------------------------------------------
// The index lists of a run (<= 8 bytes x the points enabled at its start) land in ONE pinned host
// block: the D2H copies are asynchronous at PCIe rate and nothing is copied a second time
// (pageable destinations cost ~0.25 ms per extracted shape at 10M points).  Pinning is slow, so
// blocks are recycled through a small process-wide pool: rh_result_free hands the block back.
namespace {
struct ArenaBlock { void *p; size_t cap; bool in_use; };
std::mutex g_arena_mu;
std::vector<ArenaBlock> g_arenas;

void *arena_acquire(size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    int bestfit = -1;
    for (size_t i = 0; i < g_arenas.size(); i++)
        if (!g_arenas[i].in_use && g_arenas[i].cap >= bytes && (bestfit < 0 || g_arenas[i].cap < g_arenas[(size_t)bestfit].cap))
            bestfit = (int)i;
    if (bestfit >= 0) { g_arenas[(size_t)bestfit].in_use = true; return g_arenas[(size_t)bestfit].p; }
    for (size_t i = 0; i < g_arenas.size();) {   // too small to be useful again: give the pages back
        if (!g_arenas[i].in_use) { (void)hipHostFree(g_arenas[i].p); g_arenas.erase(g_arenas.begin() + (long)i); }
        else i++;
    }
    void *p = nullptr;
    const size_t cap = std::max<size_t>(bytes, 1 << 20);
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    g_arenas.push_back({ p, cap, true });
    return p;
}

void arena_release(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    int nfree = 0;
    for (ArenaBlock &b : g_arenas) nfree += !b.in_use;
    for (size_t i = 0; i < g_arenas.size(); i++) {
        if (g_arenas[i].p != p) continue;
        if (nfree >= 2) { (void)hipHostFree(p); g_arenas.erase(g_arenas.begin() + (long)i); }
        else g_arenas[i].in_use = false;
        return;
    }
}
}  // namespace

extern "C" void rh_result_free(rh_result *r)
{
    if (!r) return;
    arena_release(r->arena);
    free(r->shapes);
    memset(r, 0, sizeof *r);
}


namespace {

#define RUN(x) do { int rc_ = (x); if (rc_ != RH_OK) return rc_; } while (0)
#define RUNH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rh_set_error("%s: %s", #x, hipGetErrorString(e_)); return RH_E_NODEVICE; } } while (0)

// a sampled window in flight: device list + status, pinned landing zones, completion event
constexpr int RH_CHAIN_MAX = 64;   // iterations per chained octree window, at most
struct Window {
    rh_cand_entry *d_entries = nullptr, *h_entries = nullptr;   // h_entries: pinned, head of the list
    int32_t entries_cap = 0;
    char *d_status = nullptr, *h_status = nullptr;              // int32 count, int32 gave_up, u64 draws[W]
    int32_t *d_counts = nullptr, *h_counts = nullptr;           // inlier counts per list entry (h: pinned head)
    bool scored = false;                                        // the counts were computed with the window
    // chained octree windows (run_streams_device): pinned state as uploaded, per-iteration headers + events, and pinned
    // twins of the whole list and its counts (every iteration's slice lands at its list positions)
    rh_oct_state *h_ost = nullptr;
    rh_oct_iter_hdr *h_hdr = nullptr;
    rh_cand_entry *h_list = nullptr;
    int32_t *h_list_counts = nullptr, *h_list_rank = nullptr, *h_list_slot = nullptr;
    int32_t h_list_cap = 0;
    hipEvent_t ev_it[RH_CHAIN_MAX] = {};
    hipEvent_t ev = nullptr;
    int64_t k = 0;
    int32_t W = 0;
    bool pending = false;
};

void window_free(Window &w)
{
    (void)hipFree(w.d_entries); (void)hipFree(w.d_status); (void)hipFree(w.d_counts);
    (void)hipHostFree(w.h_status); (void)hipHostFree(w.h_entries); (void)hipHostFree(w.h_counts);
    (void)hipHostFree(w.h_ost); (void)hipHostFree(w.h_hdr); (void)hipHostFree(w.h_list); (void)hipHostFree(w.h_list_counts);
    (void)hipHostFree(w.h_list_rank); (void)hipHostFree(w.h_list_slot);
    for (hipEvent_t e : w.ev_it) if (e) (void)hipEventDestroy(e);
    if (w.ev) (void)hipEventDestroy(w.ev);
    w = Window();
}

// What a run allocates and the next run on the same cloud can use again (two windows, the device store, the
// pinned scratch: a dozen hipMalloc / hipHostMalloc / hipFree pairs, ~3 ms per call): parked on the cloud
// between calls, freed with it.  Only a run that ended cleanly parks its buffers (the windows' status blocks and
// the liveness flags are zero then).
// pinned staging blocks for the prepared records of large batches on their way into the device store (record()): a
// ring of four, each guarded by an event -- a copy from pageable memory is a blocking staged copy inside the runtime
// (~15 us per call, three calls per octree window)
struct PinRing {
    rh_prep *buf[4] = { nullptr, nullptr, nullptr, nullptr };
    int64_t cap[4] = { 0, 0, 0, 0 };
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    bool busy[4] = { false, false, false, false };
    int next = 0;
};

void pin_ring_free(PinRing &r)
{
    for (int i = 0; i < 4; i++) {
        if (r.busy[i] && r.ev[i]) (void)hipEventSynchronize(r.ev[i]);
        if (r.buf[i]) (void)hipHostFree(r.buf[i]);
        if (r.ev[i]) (void)hipEventDestroy(r.ev[i]);
        r.buf[i] = nullptr; r.cap[i] = 0; r.ev[i] = nullptr; r.busy[i] = false;
    }
}

struct DriverCache {
    Window win[2];
    DeviceStore st;
    int32_t *h_scr = nullptr;
    int64_t h_scr_cap = 0;
    PinRing ring;
};

void driver_cache_free(rh_cloud *c, void *p)
{
    DriverCache *dc = (DriverCache *)p;
    if (!dc) return;
    (void)hipHostFree(dc->h_scr);
    pin_ring_free(dc->ring);
    store_free(c, dc->st);
    for (Window &w : dc->win) window_free(w);
    delete dc;
}

struct Driver {
    rh_cloud *c;
    const rh_params *p;
    const double *xyz, *nrm;
    rh_rng *rng;
    int drawN;

    EnabledMirror en;
    DeviceStore st;
    std::vector<Stored> store;              // scoredshapes, reference order
    std::vector<rh_shape> shapes;           // the shapes of every candidate recorded in this run (Stored::shape)
    std::vector<rh_extracted> extracted;
    int64_t cc[4] = { 0, 0, 0, 0 };         // countcandidates (1-based like the reference)
    int64_t best = -1;                      // index into store of the running first maximum
    // Device-managed store (chained octree windows: hundreds of thousands of stored candidates).  `store` is append-only
    // then -- a dead candidate stays as a tombstone (kind -1), its index is the id the device keeps beside its record --
    // and the compaction on the device names the best survivor, so that an extraction costs the host O(dead + blocks)
    // instead of several passes over the whole store.
    bool managed = false;
    int64_t live_count = 0;
    std::vector<uint8_t> alive;                   // per entry of `store` (the dead list arrives in no order: a byte array stays in cache)
    int64_t store_count() const { return managed ? live_count : (int64_t)store.size(); }
    double t_score = 0, t_extract = 0, t_sample = 0;
    double t_last_extraction = 0;           // wall clock at the end of the latest extraction
    double tp[6] = { 0, 0, 0, 0, 0, 0 };     // extraction breakdown (RH_DRIVER_PROF=1 prints it)
    double tw[4] = { 0, 0, 0, 0 };           // windows: enqueue, wait, host list handling, record()
#ifdef RH_OCT_TIMING
    double oa_t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    long long oa_n = 0;
#endif
    int64_t nwin = 0;
    int64_t iterations = 0;
    bool terminated = false;

    // level-weighted octree sampling (octree_sampling = 1)
    bool octree = false;
    bool host_sampling = true;               // false: sampling runs on the device, no host bit mirrors needed
    int od = 1;                              // octree depth
    double oP[32], oS[32];                   // level distribution / summed scores per level
    std::vector<uint64_t> men;               // host mirror: enabled bits in Morton order
    std::vector<int32_t> mprefix;
    std::vector<double> Pwin;

    // scratch
    std::vector<rh_prep> prep_h[4];
    PinRing ring;
    std::vector<int64_t> sd;
    std::vector<double> fp, fn;

    rh_mp *mp = nullptr;                    // rh_ransac_mp: the processes sharing this scene (null: one process)
    std::vector<char> mp_buf;
    std::vector<std::vector<char>> mp_recv;
    std::vector<unsigned long long> mp_draws;

    Window win[2];
    static constexpr int32_t ENTRIES_HEAD = 4096;   // list entries that travel with the window (pack_window_kernel copies min(count, this))

    int32_t *h_scr = nullptr;           // pinned scratch: scalars read back, liveness counts, gather lists
    int64_t h_scr_cap = 0;              // in int32
    int64_t *arena = nullptr;           // pinned block for the extracted index lists (result arenas, above)
    int64_t arena_used = 0, arena_cap = 0;
    bool list_copy_pending = false;     // a list is (or may still be) on its way from idx_out to the arena

    bool clean = false;                 // set when the run ended without an error: its buffers may be parked

    ~Driver()
    {
        if (c) (void)hipStreamSynchronize(c->stream);   // no copy may still be landing in the pinned blocks
        if (c) (void)hipStreamSynchronize(c->copy_stream);
        arena_release(arena);   // null once the result owns it
        if (c && clean && c->drv_cache == nullptr && !getenv("RH_NO_DRIVER_CACHE")) {
            DriverCache *dc = new DriverCache;
            dc->win[0] = win[0]; dc->win[1] = win[1];
            dc->st = st;
            dc->h_scr = h_scr; dc->h_scr_cap = h_scr_cap;
            dc->ring = ring;   // (the stream has been waited for above: no block is busy)
            for (bool &b : dc->ring.busy) b = false;
            c->drv_cache = dc;
            c->drv_cache_free = driver_cache_free;
            return;
        }
        (void)hipHostFree(h_scr);
        pin_ring_free(ring);
        store_free(c, st);
        for (Window &w : win) window_free(w);
    }

    // pinned scratch of at least `ints` int32 (contents are not preserved when it grows)
    int ensure_scratch(int64_t ints)
    {
        if (ints <= h_scr_cap) return RH_OK;
        RUNH(hipStreamSynchronize(c->stream));
        (void)hipHostFree(h_scr);
        h_scr = nullptr;
        h_scr_cap = 0;
        const int64_t cap = std::max<int64_t>(ints, 1 << 16);
        RUNH(hipHostMalloc((void **)&h_scr, sizeof(int32_t) * (size_t)cap));
        h_scr_cap = cap;
        return RH_OK;
    }

    int init()
    {
        drawN = p->drawN;
        sd.resize((size_t)drawN);
        fp.resize(3 * (size_t)drawN);
        fn.resize(3 * (size_t)drawN);
        en.n = c->n;
        en.w.assign((size_t)c->nwords, 0);
        if (c->nwords > 0) RUN(rh_cloud_get_enabled(c, en.w.data(), c->nwords));
        en.recount();
        if (c->drv_cache != nullptr) {   // the buffers the previous run on this cloud parked
            DriverCache *dc = (DriverCache *)c->drv_cache;
            c->drv_cache = nullptr;
            win[0] = dc->win[0]; win[1] = dc->win[1];
            st = dc->st;
            h_scr = dc->h_scr; h_scr_cap = dc->h_scr_cap;
            ring = dc->ring;
            delete dc;
            for (int q = 0; q < 4; q++) st.n[q] = 0;
            for (Window &w : win) { w.pending = false; w.scored = false; }
        }
        RUN(ensure_scratch(1 << 16));
        arena_cap = std::max<int64_t>(en.count, 1);   // a point is extracted at most once
        arena = (int64_t *)arena_acquire(sizeof(int64_t) * (size_t)arena_cap);
        if (!arena) { rh_set_error("rh_ransac: cannot pin %lld bytes for the index lists", (long long)(8 * arena_cap)); return RH_E_NOMEM; }
        arena_used = 0;
        // the disabled list must describe the cloud as it is now (points disabled before the call)
        RUN(rhk_rebuild_sub_enabled(c, true));
        int32_t ndis = 0;
        RUNH(hipMemcpyAsync(&ndis, c->d_ndis, sizeof ndis, hipMemcpyDeviceToHost, c->stream));
        RUNH(hipStreamSynchronize(c->stream));
        c->n_dis = ndis;
        c->select_valid = false;
        if (!st.d_nk) RUNH(hipMalloc((void **)&st.d_nk, sizeof(int32_t) * 8));
        if (!st.live) {
            RUNH(hipMalloc((void **)&st.live, sizeof(int32_t) * (size_t)LIVE_MAX));
            RUNH(hipMemsetAsync(st.live, 0, sizeof(int32_t) * (size_t)LIVE_MAX, c->stream));
        }
        octree = p->octree_sampling != 0;
        if (octree) {
            RUN(rh_octree_ensure(c, xyz, p->octree_max_depth));
            od = c->oct_depth;
            for (int i = 0; i < od; i++) { oP[i] = 1.0 / od; oS[i] = 0.0; }
            if (host_sampling) {
                men.assign((size_t)c->nwords, 0);
                for (int64_t i = 0; i < c->n; i++)
                    if (en.test(i)) { const int32_t mp = c->h_oct_pos[(size_t)i]; men[(size_t)(mp >> 6)] |= 1ULL << (mp & 63); }
                rebuild_mprefix();
            }
        }
        return RH_OK;
    }

    void rebuild_mprefix()
    {
        mprefix.resize((size_t)c->nwords + 1);
        int32_t acc = 0;
        for (int64_t w = 0; w < c->nwords; w++) { mprefix[(size_t)w] = acc; acc += __builtin_popcountll(men[(size_t)w]); }
        mprefix[(size_t)c->nwords] = acc;
    }

    rhfit::OctView host_octview() const
    {
        rhfit::OctView oc;
        oc.code = c->h_oct_code.data(); oc.perm = c->h_oct_perm.data(); oc.pos = c->h_oct_pos.data();
        oc.men = men.data(); oc.prefix = mprefix.data();
        oc.n = c->n; oc.nwords = c->nwords; oc.depth = od;
        return oc;
    }

    // forcefitshapes! (fitting.jl:165-173) for one sampled minimal set
    int fit_set(std::vector<rh_shape> &cands)
    {
        for (int q = 0; q < drawN; q++) {
            memcpy(&fp[3 * (size_t)q], xyz + 3 * (sd[(size_t)q] - 1), 24);
            memcpy(&fn[3 * (size_t)q], nrm + 3 * (sd[(size_t)q] - 1), 24);
        }
        for (int t = 0; t < p->n_shape_types; t++) {
            rh_shape fitted;
            int32_t ok = 0;
            if (c->f32) RUN(rh_fit_f32(p->shape_types[t], fp.data(), fn.data(), drawN, p, &fitted, &ok));   // Float32 cloud: Float32 fits
            else RUN(rh_fit(p->shape_types[t], fp.data(), fn.data(), drawN, p, &fitted, &ok));
            if (ok) cands.push_back(fitted);
        }
        return RH_OK;
    }

    // one iteration's minimal sets on the host: sequential stream (mode 0) or per-set streams (mode 1)
    int sample_iteration_host(int64_t k, std::vector<rh_shape> &cands, std::vector<int32_t> &levels)
    {
        cands.clear();
        levels.clear();
        if (p->sampling_streams) en.build();
        const rhfit::OctView oc = octree ? host_octview() : rhfit::OctView();
        for (int i = 0; i < p->minsubsetN; i++) {
            if (p->sampling_streams) {
                uint64_t x = rhfit::set_stream_init(rng->s[0], (uint64_t)k, (uint64_t)i);
                uint32_t nd = 0;
                bool gave_up = false;
                int level = 1;
                const bool ok = octree ? rhfit::sample_minimal_set_octree<0>(en, oc, oP, c->n, en.count, drawN, &x, sd.data(), &nd,
                                                                          &gave_up, &level)
                                       : rhfit::sample_minimal_set<0>(en, c->n, en.count, drawN, &x, sd.data(), &nd, &gave_up);
                rng->draws += nd;
                if (gave_up) { rh_set_error("rh_ransac: sampling did not find an enabled point"); return RH_E_INTERNAL; }
                if (!ok) continue;
                const size_t before = cands.size();
                RUN(fit_set(cands));
                levels.resize(cands.size(), level);
                (void)before;
                continue;
            } else {
                // samplepointcloud4!: fitting.jl:388-428 on the root cell
                int64_t r1 = rh_rng_range(rng, c->n);
                while (!en.test(r1 - 1)) r1 = rh_rng_range(rng, c->n);
                if (en.count < drawN) continue;
                sd[0] = r1;
                for (int q = 1; q < drawN; q++) {
                    int64_t pick = en.select(rh_rng_range(rng, en.count));
                    if (pick == sd[0]) pick = en.select(rh_rng_range(rng, en.count));   // one redraw: fitting.jl:416-419
                    sd[(size_t)q] = pick;
                }
                bool distinct = true;   // allisdifferent: utilities.jl:285-295
                for (int a = 1; a < drawN && distinct; a++)
                    for (int b = 0; b < a; b++)
                        if (sd[(size_t)a] == sd[(size_t)b]) { distinct = false; break; }
                if (!distinct) continue;
            }
            RUN(fit_set(cands));
            levels.resize(cands.size(), 1);
        }
        return RH_OK;
    }

    // scorecandidates! (fitting.jl:181-190) for a batch: counts in candidate order -- the ABI's own batched call
    // (one launch for all kinds; batches of a few candidates travel as one staged transfer).  Nothing reads a
    // score before the loop ends (iterations.jl:99).
    int score(const rh_shape *cands, int32_t ncand, std::vector<int32_t> &counts)
    {
        counts.assign((size_t)ncand, 0);
        if (ncand == 0) return RH_OK;
        const double t0 = now_s();
        RUN(rh_score_batch(c, cands, ncand, p, counts.data(), nullptr));
        t_score += now_s() - t0;
        return RH_OK;
    }

    // recordscore! (fitting.jl:114-119) in candidate order + prepared records into the device store
    // dev_slots (chained octree windows): the device has appended the candidates' records to the store itself
    // (rhk_oct_advance) -- dev_slots[i] is candidate i's slot in the store of its kind
    int record(const rh_shape *cands, const int32_t *levels, int32_t ncand, const int32_t *counts, const int32_t *dev_slots = nullptr)
    {
        if (ncand == 0) return RH_OK;
        if (dev_slots != nullptr) {
            // (the device keeps a candidate's number on the host as an int32 beside its record)
            if ((int64_t)store.size() + ncand > (int64_t)0x7fff0000) { rh_set_error("rh_ransac: more than 2^31 candidates in one run"); return RH_E_CAPACITY; }
            int32_t nk[4] = { 0, 0, 0, 0 };
            for (int32_t i = 0; i < ncand; i++) {
                double lo, hi, E;
                RUN(rh_estimatescore(c->s, c->n, counts[i], p->score_mode, &lo, &hi, &E));
                Stored rec;
                rec.shape = (int32_t)shapes.size();
                shapes.push_back(cands[i]);
                rec.kind = cands[i].kind;
                rec.E = E;
                rec.slot = dev_slots[i];   // (where the device put it; stale after the first compaction -- the id rules)
                rec.sigma = counts[i];
                store.push_back(rec);
                live_count++;
                alive.push_back(1);
                nk[rec.kind]++;
                oS[levels[i] - 1] += E;   // pc.levelscore[level] += E(sc): fitting.jl:184
                if (best < 0) best = (int64_t)store.size() - 1;
                else if (E > store[(size_t)best].E) best = (int64_t)store.size() - 1;
            }
            for (int q = 0; q < 4; q++) st.n[q] += nk[q];
            return RH_OK;
        }
        // the prepared records are made here (rh_prep_host is the host twin of the device's prep_one) and go
        // straight behind the store of their kind: one small copy per kind present, no launch
        int32_t nk[4] = { 0, 0, 0, 0 };
        for (int q = 0; q < 4; q++) prep_h[q].clear();
        for (int32_t i = 0; i < ncand; i++) {
            const int q = cands[i].kind;
            prep_h[q].emplace_back();
            rh_prep_host(cands[i], &prep_h[q].back());
            nk[q]++;
        }
        // a large batch travels through a pinned block of the ring (truly asynchronous copies), a small one from where it is
        rh_prep *pin = nullptr;
        int slot_r = -1;
        if (ncand >= 64) {
            slot_r = ring.next;
            ring.next = (ring.next + 1) & 3;
            if (ring.busy[slot_r]) { RUNH(hipEventSynchronize(ring.ev[slot_r])); ring.busy[slot_r] = false; }
            if (ring.cap[slot_r] < ncand) {
                if (ring.buf[slot_r]) (void)hipHostFree(ring.buf[slot_r]);
                ring.buf[slot_r] = nullptr;
                ring.cap[slot_r] = 0;
                const int64_t cap = std::max<int64_t>(2 * (int64_t)ncand, 4096);
                RUNH(hipHostMalloc((void **)&ring.buf[slot_r], sizeof(rh_prep) * (size_t)cap));
                ring.cap[slot_r] = cap;
            }
            if (!ring.ev[slot_r]) RUNH(hipEventCreateWithFlags(&ring.ev[slot_r], hipEventDisableTiming));
            pin = ring.buf[slot_r];
        }
        int64_t poff = 0;
        for (int q = 0; q < 4; q++) {
            if (nk[q] == 0) continue;
            RUN(store_reserve(c, st, q, (int64_t)st.n[q] + nk[q]));
            const rh_prep *src = prep_h[q].data();
            if (pin != nullptr) {
                memcpy(pin + poff, prep_h[q].data(), sizeof(rh_prep) * (size_t)nk[q]);
                src = pin + poff;
                poff += nk[q];
            }
            RUNH(hipMemcpyAsync(st.prep[q] + st.n[q], src, sizeof(rh_prep) * (size_t)nk[q], hipMemcpyHostToDevice, c->stream));
        }
        if (pin != nullptr) {
            RUNH(hipEventRecord(ring.ev[slot_r], c->stream));
            ring.busy[slot_r] = true;
        }
        int32_t slot_next[4] = { st.n[0], st.n[1], st.n[2], st.n[3] };
        for (int32_t i = 0; i < ncand; i++) {   // slots follow candidate order within a kind (stable sort)
            double lo, hi, E;
            RUN(rh_estimatescore(c->s, c->n, counts[i], p->score_mode, &lo, &hi, &E));
            Stored rec;
            rec.shape = (int32_t)shapes.size();
            shapes.push_back(cands[i]);
            rec.kind = cands[i].kind;
            rec.E = E;
            rec.slot = slot_next[cands[i].kind]++;
            rec.sigma = counts[i];
            store.push_back(rec);
            if (octree) oS[levels[i] - 1] += E;   // pc.levelscore[level] += E(sc): fitting.jl:184
            // findhighestscore (fitting.jl:140-151) incrementally: first maximum, strict >
            if (best < 0) best = (int64_t)store.size() - 1;
            else if (E > store[(size_t)best].E) best = (int64_t)store.size() - 1;
        }
        for (int q = 0; q < 4; q++) st.n[q] += nk[q];
        return RH_OK;
    }

    // iterations.jl:106-140: extract the best candidate if its detection probability is high enough
    int maybe_extract(int64_t k, bool *did)
    {
        *did = false;
        if (store_count() == 0) return RH_OK;
        const double scr = store[(size_t)best].E;
        const double ppp = rh_prob(scr, cc[p->extract_s], c->n, drawN);
        if (!(ppp > p->prob_det)) return RH_OK;   // iterations.jl:123
        const double t0 = now_s();
        // refit: full-cloud scan + ascending compaction (plane.jl:137-143 ...)
        const rh_shape bestshape = shapes[(size_t)store[(size_t)best].shape];
        const size_t extracted_pos = (size_t)best;   // deleteat!(scoredshapes, best.index): iterations.jl:136
        rh_prep P;
        rh_prep_host(bestshape, &P);
        int64_t base[5] = { 0, 0, 0, 0, 0 };
        for (int q = 0; q < 4; q++) base[q + 1] = base[q] + st.n[q];
        const int64_t sum_n = base[4];
        // A small store (the usual case: root-cell sampling keeps a few hundred candidates) is checked for
        // liveness in the same stream, before the host has seen the list lengths: ONE wait per extraction.
        // (the in-stream pass is brute force over [first, end of the list) x the store: it is for small products --
        // faithful-mode spheres, which are tested against every disabled point, outgrow it as the list fills)
        int64_t live_work = 0;
        for (int q = 0; q < 4; q++) {
            const bool all_disabled = (q == RH_SPHERE && !p->sphere_uses_enabled);
            live_work += (int64_t)st.n[q] * ((all_disabled ? c->n_dis : 0) + store[(size_t)best].sigma);
        }
        const bool fast = !managed && sum_n <= LIVE_MAX && live_work <= ((int64_t)1 << 24) && !getenv("RH_NO_FAST_EXTRACT");
        // (device-managed store: the kinds laid end to end, each padded to a multiple of RH_STORE_PAD)
        int32_t pbase[5] = { 0, 0, 0, 0, 0 };
        for (int q = 0; q < 4; q++) pbase[q + 1] = pbase[q] + (st.n[q] + RH_STORE_PAD - 1) / RH_STORE_PAD * RH_STORE_PAD;
        RUN(store_reserve_aux(c, st, managed ? std::max<int64_t>(sum_n, pbase[4]) : sum_n));
        // (may wait for the stream: before anything lands in the scratch; managed: + one rh_store_best per block of the store)
        RUN(ensure_scratch(32 + 2 * sum_n + (managed ? 4 * (int64_t)(pbase[4] / RH_STORE_PAD) + 8 : 0)));
        int32_t *h_nk = h_scr + 16, *h_counts = h_scr + 32, *h_lists = h_scr + 32 + sum_n;
        const int64_t ndis_old = c->n_dis;
        if (c->f32) RUN(rhk_refit_mask_f32(c, bestshape, p->eps[bestshape.kind], p->cos_alpha[bestshape.kind], true));
        else RUN(rhk_refit_mask(c, P, bestshape.kind, p->eps[bestshape.kind], p->cos_alpha[bestshape.kind], true));
        if (list_copy_pending) {   // the previous list must have left idx_out before it is written again
            RUNH(hipStreamWaitEvent(c->stream, c->ev_copied, 0));
            list_copy_pending = false;
        }
        // ... with invalidate_indexes! (fitting.jl:197-202) folded into the compaction as enabled &= ~mask;
        // then subset bits + disabled list
        RUN(rhk_compact_refit_apply(c));
        RUN(rhk_rebuild_sub_enabled(c, false));
        if (fast) {
            rh_live_args A;
            A.f32 = c->f32 ? 1 : 0;
            int64_t lo = c->s;
            for (int q = 0; q < 4; q++) {
                const bool all_disabled = (q == RH_SPHERE && !p->sphere_uses_enabled);
                A.prep[q] = st.prep[q];
                A.nk[q] = st.n[q];
                A.base[q] = (int32_t)base[q];
                A.first[q] = all_disabled ? 0 : ndis_old;
                A.eps[q] = p->eps[q];
                A.cosa[q] = p->cos_alpha[q];
                if (st.n[q] > 0) lo = std::min(lo, A.first[q]);
            }
            // new entries of the list <= the candidate's own count on subset 1: the bits only went down since it
            // was scored, and the refit scan applies the same per-point test
            const int64_t span = (ndis_old - lo) + store[extracted_pos].sigma;
            if (sum_n > 0) RUN(rhk_liveness_small(c, lo, std::min<int64_t>(c->s - lo, span), A, st.live));
            RUN(rhk_pack_live(c, st.live, (int32_t)sum_n, h_counts, h_scr));
        } else {
            RUN(rhk_fetch2_i32(c, c->d_total, c->d_ndis, h_scr));
        }
        RUNH(hipEventRecord(c->ev_sync, c->stream));
        // the next window needs the select directory of the new bits: its two launches run while the host wakes up
        if (!host_sampling && !octree) RUN(rhk_build_select(c));
        if (octree) RUN(rhk_oct_clear_mask(c, c->refit_mask));   // (its prefix pass reuses d_total: after the read-back)
        RUNH(hipEventSynchronize(c->ev_sync));
        const int32_t total = h_scr[0], ndis_new = h_scr[1];
        rh_extracted ex;
        memset(&ex, 0, sizeof ex);
        ex.shape = bestshape;
        ex.n_inpoints = total;
        if (arena_used + total > arena_cap) { rh_set_error("rh_ransac: index arena overflow"); return RH_E_INTERNAL; }
        ex.inpoints = arena + arena_used;
        arena_used += total;
        extracted.push_back(ex);
        t_last_extraction = now_s();
        if (total > 0) {
            // pinned destination, on the copy stream: the 8 bytes per inlier cross PCIe while the compute stream
            // goes on with the next window; the next extraction waits for ev_copied before it rewrites idx_out
            // (idx_out is complete: the host has just waited for work that was queued behind the compaction)
            const double tc0 = now_s();
            RUNH(hipMemcpyAsync(ex.inpoints, c->idx_out, sizeof(int64_t) * (size_t)total, hipMemcpyDeviceToHost, c->copy_stream));
            tp[5] += now_s() - tc0;
            RUNH(hipEventRecord(c->ev_copied, c->copy_stream));
            list_copy_pending = true;
        }
        if (host_sampling) RUNH(hipStreamSynchronize(c->copy_stream));   // the host mirrors need the list now
        extracted.back().score_E = scr;
        extracted.back().iteration = k;
        double tq = now_s();
        tp[0] += tq - t0;
        if (host_sampling) en.clear(ex.inpoints, total);
        else en.count -= total;   // refit only returns enabled points
        if (octree && host_sampling) {
            for (int32_t q = 0; q < total; q++) {
                const int32_t mp = c->h_oct_pos[(size_t)(ex.inpoints[q] - 1)];
                men[(size_t)(mp >> 6)] &= ~(1ULL << (mp & 63));
            }
            rebuild_mprefix();
        }
        c->n_dis = ndis_new;

        tp[1] += now_s() - tq; tq = now_s();
        if (managed) {
            // removeinvalidshapes! (fitting.jl:209-221) with the store managed on the device: liveness counts per entry,
            // then rhk_store_compact moves the survivors to the spare arrays and hands the dead candidates' numbers
            // over -- one wait, and the host touches only the dead
            const int64_t extracted_id = (int64_t)extracted_pos;
            int32_t *h_out = h_scr + 24, *h_dead = h_counts;
            rh_store_best *h_best = (rh_store_best *)(h_scr + 32 + 2 * ((sum_n + 1) / 2 * 2));   // (8-byte aligned: the scratch is, the offset is even)
            for (int i = 0; i < 5; i++) h_out[i] = 0;
            if (sum_n > 0) {
                const int64_t nblocks = pbase[4] / RH_STORE_PAD;
                // scratch of the compaction: block counts + offsets, then (16-byte aligned) the blocks' best entries; the dead
                // list is staged in st.d_idx (as long as the store)
                const int64_t best_at = ((2 * nblocks + 16 + 3) / 4) * 4, work_ints = best_at + 4 * nblocks;
                if (st.work_cap < work_ints) {
                    RUNH(hipStreamSynchronize(c->stream));
                    (void)hipFree(st.d_work);
                    st.d_work = nullptr; st.work_cap = 0;
                    const int64_t cap = std::max<int64_t>(2 * work_ints, 4096);
                    RUNH(hipMalloc((void **)&st.d_work, sizeof(int32_t) * (size_t)cap));
                    st.work_cap = cap;
                }
                rh_store_best *d_best = (rh_store_best *)(st.d_work + best_at);
                for (int q = 0; q < 4; q++) {
                    if (st.n[q] == 0 || (st.spare_cap[q] >= st.cap[q] && st.spare_id[q] != nullptr)) continue;
                    RUNH(hipStreamSynchronize(c->stream));
                    (void)hipFree(st.spare[q]); (void)hipFree(st.spare_id[q]); (void)hipFree(st.spare_E[q]);
                    st.spare[q] = nullptr; st.spare_id[q] = nullptr; st.spare_E[q] = nullptr; st.spare_cap[q] = 0;
                    RUNH(hipMalloc((void **)&st.spare[q], sizeof(rh_prep) * (size_t)st.cap[q]));
                    RUNH(hipMalloc((void **)&st.spare_id[q], sizeof(int32_t) * (size_t)st.cap[q]));
                    RUNH(hipMalloc((void **)&st.spare_E[q], sizeof(double) * (size_t)st.cap[q]));
                    st.spare_cap[q] = st.cap[q];
                }
                RUNH(hipMemsetAsync(st.counts, 0, sizeof(int32_t) * (size_t)pbase[4], c->stream));
                // h_nk[0..3]: the kinds' lengths; [4..7]: zeros (a kind left out of a pass)
                for (int q = 0; q < 4; q++) { h_nk[q] = st.n[q]; h_nk[4 + q] = 0; }
                const bool v4 = rh_score_v4_enabled(c);   // (else: a small subset, brute force)
                if (v4) {
                    // the culled binary32-classified kernel of the batch path, over the new entries of the disabled list:
                    // its records are made from the stored prepared candidates on the fly
                    if (st.cls_cap < pbase[4]) {
                        RUNH(hipStreamSynchronize(c->stream));
                        (void)hipFree(st.d_cls); (void)hipFree(st.d_box);
                        st.d_cls = nullptr; st.d_box = nullptr; st.cls_cap = 0;
                        const int64_t cap = std::max<int64_t>(2 * (int64_t)pbase[4], 1 << 16);
                        RUNH(hipMalloc(&st.d_cls, 64 * (size_t)cap));
                        RUNH(hipMalloc((void **)&st.d_box, sizeof(float) * 11 * (size_t)cap));
                        st.cls_cap = cap;
                    }
                    RUNH(hipMemcpyAsync(st.d_nk, h_nk, 8 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                    RUN(rhk_store_cls(c, st.prep, st.n, pbase, p->eps, p->cos_alpha, st.d_cls, st.d_box, st.cls_cap));
                    // kinds that look at the same stretch of the list go in one launch (faithful-mode spheres look at all of it)
                    for (int pass = 0; pass < 2; pass++) {
                        const rh_prep *pr[4];
                        const void *cl[4];
                        const float *bx[4];
                        const int32_t *og[4], *nkp[4];
                        int64_t first = -1;
                        int32_t bound = 0;
                        for (int q = 0; q < 4; q++) {
                            const bool all_disabled = (q == RH_SPHERE && !p->sphere_uses_enabled);
                            const bool in = st.n[q] > 0 && (pass == 0 ? !all_disabled : all_disabled);
                            pr[q] = st.prep[q];
                            cl[q] = (const char *)st.d_cls + 64 * (size_t)pbase[q];
                            bx[q] = st.d_box + pbase[q];
                            og[q] = st.iota + pbase[q];
                            nkp[q] = st.d_nk + (in ? q : 4 + q);
                            if (in) { first = all_disabled ? 0 : ndis_old; bound += st.n[q]; }
                        }
                        if (first < 0 || (int64_t)ndis_new - first <= 0) continue;
                        RUN(rhk_score4_dis(c, first, (int64_t)ndis_new - first, pr, cl, bx, st.cls_cap, og, nkp, bound, p->eps, p->cos_alpha, st.counts));
                    }
                } else {
                    RUNH(hipMemcpyAsync(st.d_nk + 4, h_nk, 4 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                    for (int q = 0; q < 4; q++) {
                        if (st.n[q] == 0) continue;
                        const bool all_disabled = (q == RH_SPHERE && !p->sphere_uses_enabled);
                        const int64_t first = all_disabled ? 0 : ndis_old;
                        const int64_t cnt = (int64_t)ndis_new - first;
                        if (cnt <= 0) continue;
                        RUN(rhk_score_kind_dis(c, q, first, cnt, st.prep[q], st.iota + pbase[q], st.d_nk + 4 + q, st.n[q], p->eps[q],
                                               p->cos_alpha[q], st.counts));
                    }
                }
                rh_store_plan SP;
                for (int q = 0; q < 4; q++) {
                    SP.prep[q] = st.prep[q]; SP.spare[q] = st.spare[q]; SP.id[q] = st.id[q]; SP.spare_id[q] = st.spare_id[q];
                    SP.E[q] = st.Eb[q]; SP.spare_E[q] = st.spare_E[q];
                    SP.n[q] = st.n[q];
                }
                for (int q = 0; q < 5; q++) SP.pbase[q] = pbase[q];
                SP.counts = st.counts;
                SP.extracted_id = (int32_t)extracted_id;
                RUN(rhk_store_compact(c, SP, st.d_work, h_out, h_dead, h_best, st.d_idx, d_best));
                RUNH(hipStreamSynchronize(c->stream));
            }
            tp[2] += now_s() - tq; tq = now_s();
            const int32_t ndead = h_out[4];
            if (ndead < 1 || ndead > sum_n) { rh_set_error("rh_ransac: store compaction reported %d dead of %lld", ndead, (long long)sum_n); return RH_E_INTERNAL; }
            bool saw_extracted = false;
            for (int32_t i = 0; i < ndead; i++) {
                const int64_t id = h_dead[i];
                if (id < 0 || id >= (int64_t)store.size() || !alive[(size_t)id]) {
                    rh_set_error("rh_ransac: the device store names candidate %lld, which is not alive", (long long)id);
                    return RH_E_INTERNAL;
                }
                saw_extracted |= id == extracted_id;
                alive[(size_t)id] = 0;
                live_count--;
            }

            if (!saw_extracted) { rh_set_error("rh_ransac: the extracted candidate is missing from the dead list"); return RH_E_INTERNAL; }
            for (int q = 0; q < 4; q++) {
                if (st.n[q] == 0) continue;
                std::swap(st.prep[q], st.spare[q]);
                std::swap(st.id[q], st.spare_id[q]);
                std::swap(st.Eb[q], st.spare_E[q]);
                std::swap(st.cap[q], st.spare_cap[q]);
                st.n[q] = h_out[q];
            }
            tp[3] += now_s() - tq; tq = now_s();
            // findhighestscore over the survivors: the blocks' best entries, first maximum = greatest score, smallest number
            best = -1;
            double bE = 0;
            for (int64_t b = 0; b < (int64_t)(pbase[4] / RH_STORE_PAD); b++) {
                const rh_store_best &m = h_best[b];
                if (m.id < 0) continue;
                if (m.id >= (int64_t)store.size() || !alive[(size_t)m.id]) { rh_set_error("rh_ransac: bad best survivor %lld", m.id); return RH_E_INTERNAL; }
                if (best < 0 || m.E > bE || (m.E == bE && m.id < best)) { best = m.id; bE = m.E; }
            }
            if ((best < 0) != (live_count == 0)) { rh_set_error("rh_ransac: %lld live candidates but no best survivor", (long long)live_count); return RH_E_INTERNAL; }
            if (best >= 0 && store[(size_t)best].E != bE) { rh_set_error("rh_ransac: the device store's score of candidate %lld differs from the host's", (long long)best); return RH_E_INTERNAL; }
            tp[4] += now_s() - tq;
            t_extract += now_s() - t0;
            *did = true;
            return RH_OK;
        }
        // removeinvalidshapes!: fitting.jl:209-221, recomputed on the device (see header)
        std::vector<char> dead_slot[4];
        for (int q = 0; q < 4; q++) dead_slot[q].assign((size_t)st.n[q], 0);   // every slot is referenced by `store`
        dead_slot[store[extracted_pos].kind][(size_t)store[extracted_pos].slot] = 1;
        if (fast) {
            for (int q = 0; q < 4; q++)
                for (int32_t sl = 0; sl < st.n[q]; sl++)
                    if (h_counts[base[q] + sl] != 0) dead_slot[q][(size_t)sl] = 1;
        } else if (sum_n > 0) {
            // every kind's pass goes to its own slice of st.counts (orig = iota + base: counts[base + slot]);
            // one read-back and one wait for all of them
            bool any_live = false;
            RUNH(hipMemsetAsync(st.counts, 0, sizeof(int32_t) * (size_t)sum_n, c->stream));
            for (int q = 0; q < 4; q++) h_nk[q] = st.n[q];
            RUNH(hipMemcpyAsync(st.d_nk + 4, h_nk, 4 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            for (int q = 0; q < 4; q++) {
                if (st.n[q] == 0) continue;
                const bool all_disabled = (q == RH_SPHERE && !p->sphere_uses_enabled);
                const int64_t first = all_disabled ? 0 : ndis_old;
                const int64_t cnt = (int64_t)ndis_new - first;
                if (cnt <= 0) continue;
                RUN(rhk_score_kind_dis(c, q, first, cnt, st.prep[q], st.iota + base[q], st.d_nk + 4 + q, st.n[q], p->eps[q],
                                       p->cos_alpha[q], st.counts));
                any_live = true;
            }
            if (any_live) {
                RUNH(hipMemcpyAsync(h_counts, st.counts, sizeof(int32_t) * (size_t)sum_n, hipMemcpyDeviceToHost, c->stream));
                RUNH(hipStreamSynchronize(c->stream));
                for (int q = 0; q < 4; q++)
                    for (int32_t sl = 0; sl < st.n[q]; sl++)
                        if (h_counts[base[q] + sl] > 0) dead_slot[q][(size_t)sl] = 1;
            }
        }
        tp[2] += now_s() - tq; tq = now_s();
        // drop dead candidates on the host (order preserved), compact the device store
        std::vector<int32_t> remap[4];
        for (int q = 0; q < 4; q++) {
            remap[q].assign((size_t)st.n[q], -1);
            int32_t *lst = h_lists + base[q];      // pinned, one slice per kind: nothing waits between the kinds
            int32_t alive = 0;
            for (int32_t sl = 0; sl < st.n[q]; sl++)
                if (!dead_slot[q][(size_t)sl]) {
                    remap[q][(size_t)sl] = alive;
                    lst[alive++] = sl;
                }
            if (alive != st.n[q]) {
                if (alive > 0) {
                    if (st.spare_cap[q] < st.cap[q]) {
                        RUNH(hipStreamSynchronize(c->stream));
                        (void)hipFree(st.spare[q]);
                        st.spare[q] = nullptr;
                        st.spare_cap[q] = 0;
                        RUNH(hipMalloc((void **)&st.spare[q], sizeof(rh_prep) * (size_t)st.cap[q]));
                        st.spare_cap[q] = st.cap[q];
                    }
                    RUNH(hipMemcpyAsync(st.d_idx + base[q], lst, sizeof(int32_t) * (size_t)alive, hipMemcpyHostToDevice, c->stream));
                    RUN(rhk_gather_prep(c, st.prep[q], st.d_idx + base[q], alive, st.spare[q]));
                    std::swap(st.prep[q], st.spare[q]);
                    std::swap(st.cap[q], st.spare_cap[q]);
                }
                st.n[q] = alive;
            }
        }
        // (the lists stay in the scratch until the next extraction, which starts with a stream wait)
        tp[3] += now_s() - tq; tq = now_s();
        // one pass: survivors move up (order kept), and the running maximum -- first maximum, strict > -- is
        // recomputed over them on the way
        size_t wpos = 0;
        best = -1;
        double best_E = 0;
        for (size_t i = 0; i < store.size(); i++) {
            const int q = store[i].kind;
            const int32_t ns = remap[q][(size_t)store[i].slot];
            if (ns < 0 || i == extracted_pos) continue;
            if (wpos != i) store[wpos] = store[i];
            store[wpos].slot = ns;
            const double E = store[wpos].E;
            if (best < 0 || E > best_E) { best = (int64_t)wpos; best_E = E; }
            wpos++;
        }
        store.resize(wpos);
        tp[4] += now_s() - tq;
        t_extract += now_s() - t0;
        *did = true;
        return RH_OK;
    }

    // everything of iteration k after the candidates exist: iterations.jl:98-156.
    // Returns through *stop whether the loop ends after this iteration.
    int finish_iteration(int64_t k, const rh_shape *cands, const int32_t *levels, int32_t ncand, const int32_t *counts,
                         bool *did_extract, bool *stop, const int32_t *dev_slots = nullptr)
    {
        cc[2] += ncand;
        const double tr0 = now_s();
        RUN(record(cands, levels, ncand, counts, dev_slots));
        tw[3] += now_s() - tr0;
        cc[3] = k * p->minsubsetN;
        cc[1] = store_count();
        RUN(maybe_extract(k, did_extract));
        // updatelevelweight (octree.jl:198-205): in the reference it only ever produces NaN weights (header)
        if (octree) rhfit::update_level_probs(oP, oS, od);
        *stop = rh_prob((double)p->tau, cc[p->terminate_s], c->n, drawN) > p->prob_det;
        iterations = k;
        return RH_OK;
    }

    int run_sequential()
    {
        std::vector<rh_shape> cands;
        std::vector<int32_t> counts, levels;
        for (int64_t k = 1; k <= p->itermax; k++) {
            if (en.count < p->tau) break;   // iterations.jl:75
            const double t0 = now_s();
            RUN(sample_iteration_host(k, cands, levels));
            t_sample += now_s() - t0;
            RUN(score(cands.data(), (int32_t)cands.size(), counts));
            bool did = false, stop = false;
            RUN(finish_iteration(k, cands.data(), levels.data(), (int32_t)cands.size(), counts.data(), &did, &stop));
            if (stop) break;
        }
        return RH_OK;
    }

    // sampling_streams = 1 with every shape type fittable on the device: iterations are sampled,
    // fitted and scored SPECULATIVELY in windows (the enabled bits only change at an extraction, and
    // a set's draws are a pure function of (seed, k, j)); the host replays the window in order and,
    // when an extraction happens at iteration kk, throws the rest of the window away and resumes at
    // kk + 1 -- bit-identical to the sequential loop.
    int run_streams_device()
    {
        const int64_t sets_budget = 1 << 21;   // minimal sets per window: 512 iterations at minsubsetN = 4096
        const int64_t Kmax = std::max<int64_t>(1, std::min<int64_t>(512, sets_budget / std::max(1, p->minsubsetN)));
        const int64_t K = Kmax;  // longest window
        // window length in use: slow start (an extraction within the first iterations would throw a long first
        // window away), doubled by every window that is used to its end, halved by one that is cut short
        int64_t Kcur = octree ? 1 : std::min<int64_t>(Kmax, 2);   // (chained octree windows: Kchain, below)
        // (sized for the longest window whatever this run's parameters: the windows outlive the run on the cloud)
        const size_t status_bytes = (8 + sizeof(unsigned long long) * (size_t)512 + 63) / 64 * 64;
        for (Window &w : win) {
            if (w.d_status != nullptr) continue;   // parked by the previous run
            RUNH(hipMalloc((void **)&w.d_status, status_bytes));
            RUNH(hipMemsetAsync(w.d_status, 0, status_bytes, c->stream));   // kept zero by pack_window_kernel from here on
            RUNH(hipHostMalloc((void **)&w.h_status, status_bytes));
            RUNH(hipHostMalloc((void **)&w.h_entries, sizeof(rh_cand_entry) * (size_t)ENTRIES_HEAD));
            w.entries_cap = 1 << 16;
            RUNH(hipMalloc((void **)&w.d_entries, sizeof(rh_cand_entry) * (size_t)w.entries_cap));
            RUNH(hipMalloc((void **)&w.d_counts, sizeof(int32_t) * (size_t)w.entries_cap));
            RUNH(hipHostMalloc((void **)&w.h_counts, sizeof(int32_t) * (size_t)ENTRIES_HEAD));
            RUNH(hipEventCreateWithFlags(&w.ev, hipEventDisableTiming));
        }
        // With the culled score kernel (it takes its candidate counts from device memory) the window's
        // candidates are scored on the device right after they are fitted, in the same stream: the
        // host gets list + counts in one wait instead of a second round trip per window.
        const bool fused_score = rh_score_v4_enabled(c) && !getenv("RH_NO_FUSED_SCORE");
        int32_t cnt_est = 64;
        // Without the octree a window's draws depend only on (seed, k, j) and the enabled bits, so the
        // NEXT window is put on the stream before the host waits for this one: it is valid unless this
        // one ends in an extraction (then it is dropped and drawn again).  The GPU samples window
        // w + 1 while the host replays window w.
        const bool pipeline = !octree && !getenv("RH_NO_PIPELINE");
        std::vector<rh_cand_entry> entries;
        std::vector<rh_shape> cands;
        std::vector<int32_t> counts, levels, wcounts, order, wslots;
        std::vector<int64_t> slots;
        const int T = p->n_shape_types;
        // Octree windows, one process: CHAINED.  Every iteration's scores change the level distribution the next
        // one samples from (fitting.jl:184, octree.jl:198-205), so nothing can be sampled ahead.  Instead the whole
        // iteration -- sampling, fits, scoring, the level update and the copy of its candidates to the host
        // (rhk_oct_advance) -- is queued W times back to back, with an event behind each, and the host replays iteration
        // i while the device runs i + 1, ...: recordscore!, the extraction test, updatelevelweight, checking the
        // device's level distribution against its own bit for bit.  The device ends the window (stop flag: the remaining
        // launches return at once) at the first iteration whose extraction test passes in its arithmetic; the decision
        // is the host's.
        const bool chain = octree && fused_score && mp == nullptr && !getenv("RH_NO_OCT_CHAIN");
        if (chain) {
            int64_t Kchain = 8;
            if (const char *e = getenv("RH_OCT_CHAIN_W")) Kchain = std::max<int64_t>(1, std::min<int64_t>(atoll(e), RH_CHAIN_MAX));
            if (c->oct_state == nullptr) RUNH(hipMalloc((void **)&c->oct_state, sizeof(rh_oct_state)));
            managed = !getenv("RH_NO_MANAGED_STORE");   // (the store is empty here: run_streams_device is where a run starts)
            auto ensure_pinned = [&](Window &w) -> int {
                if (w.h_ost == nullptr) {
                    RUNH(hipHostMalloc((void **)&w.h_ost, sizeof(rh_oct_state)));
                    RUNH(hipHostMalloc((void **)&w.h_hdr, sizeof(rh_oct_iter_hdr) * RH_CHAIN_MAX));
                    for (hipEvent_t &e : w.ev_it) RUNH(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                }
                if (w.h_list_cap < w.entries_cap) {
                    (void)hipHostFree(w.h_list); (void)hipHostFree(w.h_list_counts); (void)hipHostFree(w.h_list_rank); (void)hipHostFree(w.h_list_slot);
                    w.h_list = nullptr; w.h_list_counts = w.h_list_rank = w.h_list_slot = nullptr; w.h_list_cap = 0;
                    RUNH(hipHostMalloc((void **)&w.h_list, sizeof(rh_cand_entry) * (size_t)w.entries_cap));
                    RUNH(hipHostMalloc((void **)&w.h_list_counts, sizeof(int32_t) * (size_t)w.entries_cap));
                    RUNH(hipHostMalloc((void **)&w.h_list_rank, sizeof(int32_t) * (size_t)w.entries_cap));
                    RUNH(hipHostMalloc((void **)&w.h_list_slot, sizeof(int32_t) * (size_t)w.entries_cap));
                    w.h_list_cap = w.entries_cap;
                }
                return RH_OK;
            };
            const int32_t per_it = (int32_t)std::min<int64_t>((int64_t)p->minsubsetN * T, (int64_t)INT32_MAX / 2);
            // the score launch of an iteration is sized for this share of the previous iteration's candidates (the tail
            // launch covers the rest)
            int64_t bound_pct = 200;
            if (const char *e = getenv("RH_OCT_BOUND_PCT")) bound_pct = std::max<int64_t>(100, std::min<int64_t>(atoll(e), 1000));
            // Two windows in flight.  A window that is not the first after an extraction CONTINUES from the state the device
            // holds (level scores and distribution, best score, counters, store fill): nothing is uploaded, the next
            // window is queued before the host has replayed the current one, and the device never waits for the host
            // between windows (it used to idle ~0.2 ms at every window boundary without an extraction).  Whatever ends
            // a window early -- an extraction, the stop flag, a full list -- empties the pipeline: what is still queued
            // returns at once (stop flag) or is simply not replayed, and the next window starts from the host's state.
            auto enqueue = [&](Window &w, int64_t k0, int32_t W, bool upload, int64_t ahead) -> int {
                // one iteration's candidates must fit the list (a longer list is only a matter of how far a window gets)
                if (w.entries_cap < per_it + per_it / 4) {
                    RUNH(hipStreamSynchronize(c->stream));
                    (void)hipFree(w.d_entries); (void)hipFree(w.d_counts);
                    w.d_entries = nullptr; w.d_counts = nullptr;
                    w.entries_cap = per_it + per_it / 4;
                    RUNH(hipMalloc((void **)&w.d_entries, sizeof(rh_cand_entry) * (size_t)w.entries_cap));
                    RUNH(hipMalloc((void **)&w.d_counts, sizeof(int32_t) * (size_t)w.entries_cap));
                }
                RUN(ensure_pinned(w));
                if (upload) {
                    rh_oct_state &h = *w.h_ost;
                    memset(&h, 0, sizeof h);
                    for (int i = 0; i < od; i++) { h.S[i] = oS[i]; h.P[i] = oP[i]; }
                    h.has_best = store_count() == 0 ? 0 : 1;
                    h.best_E = store_count() == 0 ? 0.0 : store[(size_t)best].E;
                    h.store_count = (long long)store_count();
                    h.appended = (long long)store.size();
                    h.cc2 = cc[2];
                    // the iterations append their candidates' records to the device store: room for the windows that can be
                    // in flight before the next upload (the device checks the capacity itself and ends the window otherwise)
                    for (int q = 0; q < 4; q++) {
                        int64_t slots_of_kind = 0;   // a minimal set yields at most one candidate per entry of shape_types
                        for (int ti = 0; ti < T; ti++) slots_of_kind += p->shape_types[ti] == q ? 1 : 0;
                        if (slots_of_kind > 0) RUN(store_reserve(c, st, q, (int64_t)st.n[q] + ahead * p->minsubsetN * slots_of_kind));
                        h.store_prep[q] = st.prep[q];
                        h.store_id[q] = st.id[q];
                        h.store_E[q] = st.Eb[q];
                        h.store_cap[q] = st.cap[q];
                        h.store_n[q] = st.n[q];
                    }
                    RUNH(hipMemcpyAsync(c->oct_state, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
                } else {
                    RUN(rhk_oct_window_begin(c, c->oct_state));   // the list of this window starts at position 0
                }
                RUNH(hipMemsetAsync(w.d_status, 0, status_bytes, c->stream));
                RUN(rh_ensure_batch(c, w.entries_cap));
                const uint64_t *enw[4];
                const rh_prep *pr[4];
                const int32_t *og[4], *nkp[4];
                const void *clsw[4];
                const float *boxw[4];
                for (int q = 0; q < 4; q++) {
                    enw[q] = (q == RH_SPHERE && !p->sphere_uses_enabled) ? nullptr : c->sub_enabled;
                    pr[q] = c->d_prep + (int64_t)q * c->batch_cap;
                    og[q] = c->d_orig + (int64_t)q * c->batch_cap;
                    nkp[q] = c->d_nk + q;
                    clsw[q] = (const char *)c->d_qpre + (size_t)q * (size_t)c->batch_cap * 64;
                    boxw[q] = c->d_box + (int64_t)q * c->batch_cap;
                }
                c->s4_stop = &c->oct_state->stop;
                c->s4_open_count = true;
                int rc = RH_OK;
                for (int32_t it = 0; it < W && rc == RH_OK; it++) {
                    rc = rhk_sample_fit(c, p, rng->s[0], k0 + it, 1, (int32_t)en.count, c->oct_state->P, w.d_entries, w.entries_cap, w.d_status, 1,
                                        c->d_nk, it, c->oct_state);
                    if (rc == RH_OK) rc = rhk_prep_entries(c, w.d_entries, (const int32_t *)w.d_status, w.entries_cap, per_it, w.d_counts, 1, p->eps,
                                                           p->cos_alpha, c->oct_state);
                    if (rc == RH_OK) rc = rhk_score_all_groups(c, enw, pr, og, nkp, std::min<int32_t>(per_it, std::max<int32_t>((int32_t)((int64_t)cnt_est * bound_pct / 100) + 64, 1024)), p->eps,
                                                               p->cos_alpha, w.d_counts, nullptr, clsw, boxw, 4 * c->batch_cap);
                    if (rc == RH_OK) rc = rhk_oct_advance(c, p, c->oct_state, w.d_entries, w.d_status, w.entries_cap, w.d_counts, it, k0 + it, w.h_list,
                                                          w.h_list_counts, w.h_list_rank, w.h_list_slot, w.h_hdr);
                    if (rc == RH_OK && hipEventRecord(w.ev_it[it], c->stream) != hipSuccess) { rh_set_error("hipEventRecord failed"); rc = RH_E_NODEVICE; }
                }
                c->s4_stop = nullptr;
                c->s4_open_count = false;
                if (rc != RH_OK) return rc;
                nwin++;
                return RH_OK;
            };
            struct Flight { int wi; int64_t k0; int32_t W; };
            Flight fl[2];
            int nfl = 0, next_w = 0;
            const int max_flight = getenv("RH_OCT_ONE_WINDOW") ? 1 : 2;
            // (iterations per window: the launches of Kchain iterations are in the queue at most, whatever the number of
            // windows they are cut into -- a deeper queue makes the launches themselves slow)
            int64_t Wfl = std::max<int64_t>(1, max_flight == 2 ? (Kchain * 3) / 8 : Kchain);   // (8 -> two windows of 3: swept 2 / 3 / 4 / 6 -> 0.0482 / 0.0474 / 0.0484 / 0.0492 s)
            if (const char *e = getenv("RH_OCT_WINDOW_ITERS")) Wfl = std::max<int64_t>(1, std::min<int64_t>(atoll(e), RH_CHAIN_MAX));
            bool need_upload = true;
            int64_t k = 1, k_enq = 1;
            for (;;) {
                if (nfl == 0 && (k > p->itermax || en.count < p->tau)) break;
                const double t0 = now_s();
                while (nfl < max_flight && k_enq <= p->itermax && !(need_upload && nfl > 0)) {
                    // Is iteration k_enq certain to extract?  (prob() grows with the counters and the best score can only
                    // rise: "the stored best already passes with the counters as they are" decides it.)  Then the window is
                    // that one iteration -- everything behind it would be queued for nothing.
                    bool certain = false;
                    if (need_upload && store_count() > 0) {
                        int64_t lb[4] = { 0, store_count(), cc[2], k_enq * p->minsubsetN };
                        certain = rh_prob(store[(size_t)best].E, lb[p->extract_s], c->n, drawN) > p->prob_det;
                    }
                    const int32_t W = (int32_t)std::min<int64_t>(certain ? 1 : Wfl, p->itermax - k_enq + 1);
                    RUN(enqueue(win[next_w], k_enq, W, need_upload, 2 * Kchain));
                    fl[nfl++] = Flight{ next_w, k_enq, W };
                    next_w ^= 1;
                    k_enq += W;
                    need_upload = false;
                    if (certain) break;
                }
                const double tw0 = now_s();
                tw[0] += tw0 - t0;
                t_sample += tw0 - t0;
                if (nfl == 0) break;
                const Flight F = fl[0];
                fl[0] = fl[1];
                nfl--;
                Window &w = win[F.wi];
                const int32_t W = F.W;
                // ---- replay it, iteration by iteration, as the results arrive
                bool stop = false, did = false, regrow = false, refill = false;
                int32_t it = 0;
                for (; it < W; it++) {
                    const double ta = now_s();
                    RUNH(hipEventSynchronize(w.ev_it[it]));
                    const double tb = now_s();
                    tw[1] += tb - ta;
                    const rh_oct_iter_hdr &H = w.h_hdr[it];
#ifdef RH_OCT_TIMING
                    if (!H.skipped) { for (int i = 0; i < 7; i++) oa_t[i] += (double)H.t[i] / 100.0; oa_n++; }
#endif
                    if (H.skipped) break;                       // the device saw an extraction coming that the host did not take: go on from here
                    if (H.gave_up) { rh_set_error("rh_ransac: sampling did not find an enabled point"); return RH_E_INTERNAL; }
                    // the list or the store is full: this iteration is drawn again -- in a longer list (regrow) / behind an upload
                    // that reserves the store anew
                    if (H.overflow) { regrow = (H.overflow & 1) != 0; refill = true; break; }
                    if (en.count < p->tau) { stop = true; break; }
                    const int32_t cnt = H.end - H.start;
                    // candidate order = slot order: the device ranked the entries (no sort here)
                    cands.resize((size_t)cnt);
                    levels.resize((size_t)cnt);
                    counts.resize((size_t)cnt);
                    wslots.resize((size_t)cnt);
                    for (int32_t i = 0; i < cnt; i++) {
                        const int32_t r = w.h_list_rank[H.start + i];
                        if (r < 0 || r >= cnt) { rh_set_error("rh_ransac: bad candidate rank from the device (%d of %d)", r, cnt); return RH_E_INTERNAL; }
                        const rh_cand_entry &e = w.h_list[H.start + i];
                        cands[(size_t)r] = e.shape; levels[(size_t)r] = e.level;
                        counts[(size_t)r] = w.h_list_counts[H.start + i];
                        wslots[(size_t)r] = w.h_list_slot[H.start + i];
                    }
                    cnt_est = cnt;
                    rng->draws += (int64_t)H.draws;
                    const double tc = now_s();
                    tw[2] += tc - tb;
                    t_sample += tc - ta;
                    RUN(finish_iteration(F.k0 + it, cands.data(), levels.data(), cnt, counts.data(), &did, &stop, wslots.data()));
                    if (memcmp(oP, H.P, sizeof(double) * (size_t)od) != 0) {
                        // (the device advanced the level distribution with the operations of update_level_probs on the sums
                        // it built in candidate order: any difference is a defect, never a rounding matter)
                        rh_set_error("rh_ransac: the device's level distribution left the host's at iteration %lld", (long long)(F.k0 + it));
                        return RH_E_INTERNAL;
                    }
                    if (stop || did) { it++; break; }
                }
                k = F.k0 + it;
                if (it < W || did || stop || regrow || refill) {   // the window ended early: whatever is queued behind it is void
                    nfl = 0;
                    k_enq = k;
                    need_upload = true;
                }
                if (regrow) {
                    RUNH(hipStreamSynchronize(c->stream));
                    for (Window &g : win) {
                        (void)hipFree(g.d_entries); (void)hipFree(g.d_counts);
                        g.d_entries = nullptr; g.d_counts = nullptr;
                        g.entries_cap *= 2;
                        RUNH(hipMalloc((void **)&g.d_entries, sizeof(rh_cand_entry) * (size_t)g.entries_cap));
                        RUNH(hipMalloc((void **)&g.d_counts, sizeof(int32_t) * (size_t)g.entries_cap));
                    }
                }
                if (stop) break;
            }
            // the tail of a window that was cut short may still be in the queue; the status blocks go back zeroed
            for (Window &w : win) RUNH(hipMemsetAsync(w.d_status, 0, status_bytes, c->stream));
            RUNH(hipStreamSynchronize(c->stream));
            return RH_OK;
        }
        auto issue = [&](Window &w, int64_t k0, int32_t W) -> int {
            const double *d_P = nullptr;
            if (octree) {
                // the level distribution of every iteration of the window, assuming no candidate is
                // scored inside it (the window is cut at the first iteration that has one)
                Pwin.resize((size_t)W * (size_t)od);
                double Pw[32];
                for (int i = 0; i < od; i++) Pw[i] = oP[i];
                for (int32_t it = 0; it < W; it++) {
                    for (int i = 0; i < od; i++) Pwin[(size_t)it * od + i] = Pw[i];
                    rhfit::update_level_probs(Pw, oS, od);
                }
                if ((int64_t)Pwin.size() > c->oct_P_cap) {
                    RUNH(hipStreamSynchronize(c->stream));
                    (void)hipFree(c->oct_P);
                    c->oct_P = nullptr;
                    c->oct_P_cap = (int64_t)K * 32;
                    RUNH(hipMalloc((void **)&c->oct_P, sizeof(double) * (size_t)c->oct_P_cap));
                }
                RUNH(hipMemcpyAsync(c->oct_P, Pwin.data(), sizeof(double) * Pwin.size(), hipMemcpyHostToDevice, c->stream));
                d_P = c->oct_P;
            }
            RUN(rhk_sample_fit(c, p, rng->s[0], k0, W, (int32_t)en.count, d_P, w.d_entries, w.entries_cap, w.d_status, 1,
                               fused_score ? c->d_nk : nullptr));
            w.scored = false;
            if (fused_score) {
                RUN(rh_ensure_batch(c, w.entries_cap));
                // launch sizes from the previous windows' list lengths; any length is handled (the
                // kernels read the true count), a longer list only gets fewer blocks per candidate
                const int32_t bound = std::min<int32_t>(w.entries_cap, std::max<int32_t>(4 * cnt_est, 1024));
                RUN(rhk_prep_entries(c, w.d_entries, (const int32_t *)w.d_status, w.entries_cap, w.entries_cap, w.d_counts, 1, p->eps, p->cos_alpha));
                const uint64_t *enw[4];
                const rh_prep *pr[4];
                const int32_t *og[4], *nkp[4];
                for (int q = 0; q < 4; q++) {
                    enw[q] = (q == RH_SPHERE && !p->sphere_uses_enabled) ? nullptr : c->sub_enabled;
                    pr[q] = c->d_prep + (int64_t)q * c->batch_cap;
                    og[q] = c->d_orig + (int64_t)q * c->batch_cap;
                    nkp[q] = c->d_nk + q;
                }
                const void *clsw[4];
                const float *boxw[4];
                for (int q = 0; q < 4; q++) {
                    clsw[q] = (const char *)c->d_qpre + (size_t)q * (size_t)c->batch_cap * 64;
                    boxw[q] = c->d_box + (int64_t)q * c->batch_cap;
                }
                c->s4_open_count = true;   // (bound is a guess: the kernel's tail launch covers a longer list)
                const int rcs = rhk_score_all_groups(c, enw, pr, og, nkp, bound, p->eps, p->cos_alpha, w.d_counts, nullptr,
                                                     clsw, boxw, 4 * c->batch_cap);
                c->s4_open_count = false;
                if (rcs != RH_OK) return rcs;
                w.scored = true;
            }
            // status + head of the list (+ counts) land in pinned host memory through one small kernel
            RUN(rhk_pack_window(c, w.d_status, W, w.d_entries, w.scored ? w.d_counts : nullptr, ENTRIES_HEAD, w.h_status,
                                w.h_entries, w.h_counts));
            RUNH(hipEventRecord(w.ev, c->stream));
            w.k = k0; w.W = W; w.pending = true;
            return RH_OK;
        };
        int cur = 0;
        int64_t k = 1;
        while (k <= p->itermax) {
            if (en.count < p->tau) break;
            Window &A = win[cur], &B = win[1 - cur];
            const double t0 = now_s();
            // Is iteration k certain to extract?  prob() grows with the candidate counters and the best score can
            // only rise, so "the stored best already passes with the counters as they are now" decides it before
            // anything of this window is known.  Then everything behind iteration k would be thrown away: the
            // window is one iteration long and nothing is speculated behind it (the refit scan would queue
            // behind that work).
            bool certain = false;
            if (!store.empty()) {
                int64_t lb[4] = { 0, (int64_t)store.size(), cc[2], k * p->minsubsetN };
                certain = rh_prob(store[(size_t)best].E, lb[p->extract_s], c->n, drawN) > p->prob_det;
            }
            if (!(A.pending && A.k == k))
                RUN(issue(A, k, (int32_t)std::min<int64_t>(certain ? 1 : Kcur, p->itermax - k + 1)));
            const int32_t W = A.W;
            B.pending = false;
            if (pipeline && !certain && k + W <= p->itermax)
                RUN(issue(B, k + W, (int32_t)std::min<int64_t>(Kcur, p->itermax - (k + W) + 1)));
            const double tw0 = now_s();
            tw[0] += tw0 - t0;
            RUNH(hipEventSynchronize(A.ev));
            tw[1] += now_s() - tw0;
            nwin++;
            A.pending = false;
            int32_t cnt = ((const int32_t *)A.h_status)[0];
            const int32_t gave_up = ((const int32_t *)A.h_status)[1];
            const unsigned long long *draws = (const unsigned long long *)(A.h_status + 8);
            if (gave_up && mp == nullptr) { rh_set_error("rh_ransac: sampling did not find an enabled point"); return RH_E_INTERNAL; }
            if (cnt > A.entries_cap && mp == nullptr) {   // the list overflowed: grow it and draw the window again
                RUNH(hipStreamSynchronize(c->stream));
                B.pending = false;
                (void)hipFree(A.d_entries);
                A.d_entries = nullptr;
                A.entries_cap = cnt + cnt / 4;
                RUNH(hipMalloc((void **)&A.d_entries, sizeof(rh_cand_entry) * (size_t)A.entries_cap));
                (void)hipFree(A.d_counts);
                A.d_counts = nullptr;
                RUNH(hipMalloc((void **)&A.d_counts, sizeof(int32_t) * (size_t)A.entries_cap));
                t_sample += now_s() - t0;
                continue;
            }
            cnt_est = cnt;
            const bool overflow = cnt > A.entries_cap;   // (only reachable with mp: handled collectively below)
            if (overflow) cnt = 0;
            entries.resize((size_t)cnt);
            wcounts.resize((size_t)cnt);
            if (cnt > 0) {
                const int32_t head = std::min(cnt, ENTRIES_HEAD);
                memcpy(entries.data(), A.h_entries, sizeof(rh_cand_entry) * (size_t)head);
                if (A.scored) memcpy(wcounts.data(), A.h_counts, sizeof(int32_t) * (size_t)head);
                if (cnt > head) {
                    RUNH(hipMemcpyAsync(entries.data() + head, A.d_entries + head, sizeof(rh_cand_entry) * (size_t)(cnt - head),
                                        hipMemcpyDeviceToHost, c->stream));
                    if (A.scored)
                        RUNH(hipMemcpyAsync(wcounts.data() + head, A.d_counts + head, sizeof(int32_t) * (size_t)(cnt - head),
                                            hipMemcpyDeviceToHost, c->stream));
                    RUNH(hipStreamSynchronize(c->stream));
                }
            }
            if (mp != nullptr) {
                // Every process drew its share of the window's minimal sets (set j of an iteration belongs to rank
                // j % world): publish the local list -- entries, their counts, the draws per iteration -- and collect
                // everybody's.  The union, in slot order, is the list one process would have produced; from here on every
                // rank replays the same window and takes the same decisions (extractions included, each on its replica).
                struct Hdr { int32_t cnt, overflow, gave_up, W, scored, pad; };
                const size_t bytes = sizeof(Hdr) + sizeof(unsigned long long) * (size_t)W + (sizeof(rh_cand_entry) + sizeof(int32_t)) * (size_t)cnt;
                mp_buf.resize(bytes);
                Hdr h = { cnt, overflow ? 1 : 0, gave_up, W, A.scored ? 1 : 0, 0 };
                char *q = mp_buf.data();
                memcpy(q, &h, sizeof h); q += sizeof h;
                memcpy(q, draws, sizeof(unsigned long long) * (size_t)W); q += sizeof(unsigned long long) * (size_t)W;
                if (cnt > 0) {
                    memcpy(q, entries.data(), sizeof(rh_cand_entry) * (size_t)cnt); q += sizeof(rh_cand_entry) * (size_t)cnt;
                    memcpy(q, wcounts.data(), sizeof(int32_t) * (size_t)cnt);
                }
                // (a list longer than the exchange slot travels in pieces: mp_exchange_any)
                RUN(mp_exchange_any(mp, mp_buf.data(), (int64_t)bytes, mp_recv));
                bool any_overflow = false, any_gave_up = false;
                int64_t total = 0;
                for (int r = 0; r < mp->world; r++) {
                    Hdr hr;
                    if (mp_recv[(size_t)r].size() < sizeof hr) { rh_set_error("rh_ransac_mp: short exchange from rank %d", r); return RH_E_INTERNAL; }
                    memcpy(&hr, mp_recv[(size_t)r].data(), sizeof hr);
                    if (hr.W != W || hr.scored != h.scored) { rh_set_error("rh_ransac_mp: rank %d is at another window (W %d vs %d)", r, hr.W, W); return RH_E_INTERNAL; }
                    any_overflow |= hr.overflow != 0;
                    any_gave_up |= hr.gave_up != 0;
                    total += hr.cnt;
                }
                if (any_gave_up) { rh_set_error("rh_ransac: sampling did not find an enabled point"); return RH_E_INTERNAL; }
                if (any_overflow) {   // some rank's list overflowed: it grows, and everybody draws the window again
                    RUNH(hipStreamSynchronize(c->stream));
                    B.pending = false;
                    if (overflow) {
                        (void)hipFree(A.d_entries);
                        A.d_entries = nullptr;
                        A.entries_cap = cnt_est + cnt_est / 4;
                        RUNH(hipMalloc((void **)&A.d_entries, sizeof(rh_cand_entry) * (size_t)A.entries_cap));
                        (void)hipFree(A.d_counts);
                        A.d_counts = nullptr;
                        RUNH(hipMalloc((void **)&A.d_counts, sizeof(int32_t) * (size_t)A.entries_cap));
                    }
                    t_sample += now_s() - t0;
                    continue;
                }
                if (total > (int64_t)INT32_MAX / 2) { rh_set_error("rh_ransac_mp: window with %lld candidates", (long long)total); return RH_E_CAPACITY; }
                mp_draws.assign((size_t)W, 0ULL);
                entries.resize((size_t)total);
                wcounts.resize((size_t)total);
                size_t at = 0;
                for (int r = 0; r < mp->world; r++) {
                    const char *src = mp_recv[(size_t)r].data();
                    Hdr hr;
                    memcpy(&hr, src, sizeof hr); src += sizeof hr;
                    if (mp_recv[(size_t)r].size() != sizeof hr + sizeof(unsigned long long) * (size_t)W + (sizeof(rh_cand_entry) + sizeof(int32_t)) * (size_t)hr.cnt) {
                        rh_set_error("rh_ransac_mp: exchange from rank %d has the wrong length", r);
                        return RH_E_INTERNAL;
                    }
                    for (int32_t i = 0; i < W; i++) { unsigned long long d; memcpy(&d, src + 8 * (size_t)i, 8); mp_draws[(size_t)i] += d; }
                    src += sizeof(unsigned long long) * (size_t)W;
                    if (hr.cnt > 0) {
                        memcpy(entries.data() + at, src, sizeof(rh_cand_entry) * (size_t)hr.cnt); src += sizeof(rh_cand_entry) * (size_t)hr.cnt;
                        memcpy(wcounts.data() + at, src, sizeof(int32_t) * (size_t)hr.cnt);
                        at += (size_t)hr.cnt;
                    }
                }
                cnt = (int32_t)total;
                draws = mp_draws.data();
            }
            if (cnt > 0) {
                // candidate order of the reference = slot order; the counts travel with their entries
                order.resize((size_t)cnt);
                for (int32_t i = 0; i < cnt; i++) order[(size_t)i] = i;
                std::sort(order.begin(), order.end(),
                          [&](int32_t a, int32_t b) { return entries[(size_t)a].slot < entries[(size_t)b].slot; });
            }
            t_sample += now_s() - t0;
            const double tw2 = now_s();
            if (octree && cnt > 0) {
                // candidates after the first candidate-bearing iteration were drawn from a stale level
                // distribution: drop them (they are re-drawn in the next window)
                const int64_t per_it = (int64_t)p->minsubsetN * T;
                const int64_t first_it = entries[(size_t)order[0]].slot / per_it;
                int32_t keep = 0;
                while (keep < cnt && entries[(size_t)order[(size_t)keep]].slot / per_it == first_it) keep++;
                cnt = keep;
            }
            cands.resize((size_t)cnt);
            levels.resize((size_t)cnt);
            slots.resize((size_t)cnt);
            counts.resize((size_t)cnt);
            for (int32_t i = 0; i < cnt; i++) {
                const rh_cand_entry &e = entries[(size_t)order[(size_t)i]];
                cands[(size_t)i] = e.shape; levels[(size_t)i] = e.level; slots[(size_t)i] = e.slot;
                if (A.scored) counts[(size_t)i] = wcounts[(size_t)order[(size_t)i]];
            }
            if (!A.scored) RUN(score(cands.data(), cnt, counts));
            tw[2] += now_s() - tw2;
            // replay the window in iteration order
            int32_t pos = 0;
            bool stop = false, did = false;
            int32_t it = 0;
            for (; it < W; it++) {
                const int64_t kk = k + it;
                if (en.count < p->tau) { stop = true; break; }   // iterations.jl:75 (only after an extraction)
                const int64_t slot_end = (int64_t)(it + 1) * p->minsubsetN * T;
                int32_t e = pos;
                while (e < cnt && slots[(size_t)e] < slot_end) e++;
                rng->draws += (int64_t)draws[it];
                RUN(finish_iteration(kk, cands.data() + pos, levels.data() + pos, e - pos, counts.data() + pos, &did, &stop));
                const bool cut = octree && e > pos;   // new scores change the level distribution
                pos = e;
                if (stop || did || cut) { it++; break; }
            }
            k += it;
            if (stop) break;
            // the speculated window stands only if this one ran to its end without touching the enabled bits
            if (B.pending && !did && it == W && B.k == k) cur = 1 - cur;
            else B.pending = false;
            // a window cut short wasted its tail: halve; a window used to the end: double
            if (it < W) Kcur = std::max<int64_t>(1, std::min<int64_t>(Kcur, it) / 2);
            else if (!certain) Kcur = std::min<int64_t>(K, Kcur * 2);
        }
        // nothing of a dropped window may still be in flight when the buffers go away
        RUNH(hipStreamSynchronize(c->stream));
        return RH_OK;
    }
};

#undef RUN
#undef RUNH

}  // namespace

static int ransac_impl(rh_cloud *c, const double *xyz, const double *nrm, const rh_params *p, rh_rng *rng, rh_mp *mp,
                       rh_result *out);

extern "C" int rh_ransac(rh_cloud *c, const double *xyz, const double *nrm, const rh_params *p, rh_rng *rng,
                         rh_result *out)
{
    return ransac_impl(c, xyz, nrm, p, rng, nullptr, out);
}

// ransac() on a Float32 cloud from Julia's Vector{SVector{3,Float32}} memory as is (rh_ransac takes the same values as
// doubles): the host-side fits of sampling_streams = 0 read them
extern "C" int rh_ransac_f32(rh_cloud *c, const float *xyz, const float *nrm, const rh_params *p, rh_rng *rng, rh_result *out)
{
    if (!c) { rh_set_error("rh_ransac_f32: NULL argument"); return RH_E_INVALID; }
    if (!c->f32) { rh_set_error("rh_ransac_f32: the cloud is not a Float32 cloud (rh_cloud_create_f32)"); return RH_E_INVALID; }
    if (c->n > 0 && (!xyz || !nrm)) { rh_set_error("rh_ransac_f32: xyz/nrm are NULL"); return RH_E_INVALID; }
    std::vector<double> x((size_t)(3 * c->n)), n((size_t)(3 * c->n));
    for (int64_t i = 0; i < 3 * c->n; i++) { x[(size_t)i] = (double)xyz[i]; n[(size_t)i] = (double)nrm[i]; }
    return ransac_impl(c, x.data(), n.data(), p, rng, nullptr, out);
}

// ransac() on ONE scene by the `world` processes of `mp` (one per GPU, each with a replica of the cloud in the same
// state): the minimal sets of every iteration are dealt round-robin to the ranks -- sampling, fits and scoring of
// a window shrink by the number of ranks -- and the ranks exchange their windows' candidate lists through `mp`
// (host shared memory: the lists are tiny).  Every rank replays the merged window, takes the same decisions and runs
// every extraction on its own replica, so every rank returns the result rh_ransac returns for the same inputs, bit
// for bit, and leaves its cloud in the same state.  Needs sampling_streams = 1 (per-set random streams).
extern "C" int rh_ransac_mp(rh_cloud *c, const double *xyz, const double *nrm, const rh_params *p, rh_rng *rng, rh_mp *mp,
                            rh_result *out)
{
    if (!mp) { rh_set_error("rh_ransac_mp: mp is NULL"); return RH_E_INVALID; }
    if (!c || !p) { rh_set_error("rh_ransac_mp: NULL argument"); return RH_E_INVALID; }
    if (!p->sampling_streams) { rh_set_error("rh_ransac_mp needs sampling_streams = 1 (one random stream per minimal set)"); return RH_E_INVALID; }
    if (p->minsubsetN < mp->world) {   // (a rank without a single minimal set per iteration would have nothing to launch)
        rh_set_error("rh_ransac_mp: minsubsetN = %d is below the number of ranks (%d)", p->minsubsetN, mp->world);
        return RH_E_INVALID;
    }
    c->mp_rank = mp->rank;
    c->mp_world = mp->world;
    const int rc = ransac_impl(c, xyz, nrm, p, rng, mp->world > 1 ? mp : nullptr, out);
    c->mp_rank = 0;
    c->mp_world = 1;
    return rc;
}

static int ransac_impl(rh_cloud *c, const double *xyz, const double *nrm, const rh_params *p, rh_rng *rng, rh_mp *mp,
                       rh_result *out)
{
    if (!c || !p || !rng || !out) { rh_set_error("rh_ransac: NULL argument"); return RH_E_INVALID; }
    memset(out, 0, sizeof *out);
    RH_TRY(rh_validate_params(p));
    if (c->f32) {
        // Float32 cloud: fits, scoring, liveness and refit in binary32.  Its cone fit would need rank() and \ of Float32
        // matrices the way LAPACK's single precision does them (cone.jl:40-50): no fixture exists to pin a restatement on
        for (int t = 0; t < p->n_shape_types && t < 8; t++)
            if (p->shape_types[t] == RH_CONE) {
                rh_set_error("rh_ransac: FittedCone is not available in shape_types on a Float32 cloud (its fit is not restated in binary32)");
                return RH_E_INVALID;
            }
    }
    if (c->n > 0 && (!xyz || !nrm)) { rh_set_error("rh_ransac: xyz/nrm are NULL"); return RH_E_INVALID; }
    if (p->drawN < 2 || p->drawN > 16) {   // @assert drawN > 1: src/fitting.jl:386
        rh_set_error("rh_ransac: drawN=%d outside 2..16", p->drawN);
        return RH_E_INVALID;
    }
    if (p->n_shape_types < 0 || p->n_shape_types > 8) { rh_set_error("rh_ransac: bad n_shape_types"); return RH_E_INVALID; }
    for (int t = 0; t < p->n_shape_types; t++)
        if (p->shape_types[t] < 0 || p->shape_types[t] > 3) { rh_set_error("rh_ransac: bad shape type"); return RH_E_INVALID; }
    if (p->extract_s < 1 || p->extract_s > 3 || p->terminate_s < 1 || p->terminate_s > 3) {
        rh_set_error("rh_ransac: extract_s / terminate_s must be 1..3");
        return RH_E_INVALID;
    }
    if (p->minsubsetN < 0) { rh_set_error("rh_ransac: minsubsetN < 0"); return RH_E_INVALID; }
    if (p->octree_sampling && !p->sampling_streams) {
        rh_set_error("rh_ransac: octree_sampling needs sampling_streams = 1");
        return RH_E_INVALID;
    }
    RH_HIP(hipSetDevice(c->device));
    const double t_start = now_s();

    bool device_sampler = p->sampling_streams != 0 && p->drawN <= 8 && p->minsubsetN > 0 && c->n > 0;
    if (getenv("RH_HOST_SAMPLER")) device_sampler = false;   // A/B and tests: the same streams drawn on the host
    if (mp != nullptr && !device_sampler) {
        rh_set_error("rh_ransac_mp: the minimal sets are dealt to the ranks by the device sampler (drawN <= 8, minsubsetN > 0, no RH_HOST_SAMPLER)");
        return RH_E_INVALID;
    }
    Driver d;
    d.c = c; d.p = p; d.xyz = xyz; d.nrm = nrm; d.rng = rng;
    d.mp = mp;
    d.host_sampling = !device_sampler;
    RH_TRY(d.init());
    const double t_init = now_s() - t_start;
    RH_TRY(device_sampler ? d.run_streams_device() : d.run_sequential());
    const double t_loop = now_s() - t_start - t_init;
    RH_HIP(hipStreamSynchronize(c->stream));
    RH_HIP(hipStreamSynchronize(c->copy_stream));   // the index lists have landed in the arena

    out->iterations = d.iterations;
    out->candidates_scored = d.cc[2];
    out->scored_left = d.store_count();
    out->n_shapes = (int64_t)d.extracted.size();
    out->shapes = (rh_extracted *)malloc(sizeof(rh_extracted) * std::max<size_t>(d.extracted.size(), 1));
    if (!out->shapes) { rh_set_error("out of host memory"); return RH_E_NOMEM; }
    for (size_t i = 0; i < d.extracted.size(); i++) out->shapes[i] = d.extracted[i];
    d.extracted.clear();
    out->arena = d.arena;   // ownership of the index lists moved to the result
    d.arena = nullptr;
    d.clean = true;
    c->select_valid = false;
    out->seconds = now_s() - t_start;
    out->seconds_to_last_extraction = d.t_last_extraction > 0 ? d.t_last_extraction - t_start : 0.0;
    out->seconds_score = d.t_score;
    out->seconds_extract = d.t_extract;
    out->seconds_host = d.t_sample;
#ifdef RH_OCT_TIMING
    if (d.oa_n > 0) fprintf(stderr, "[rh_ransac] oct_advance phases (us, mean of %lld): copy %.2f scatter %.2f hist %.2f scan %.2f emit %.2f sums+ranks %.2f sync %.2f final %.2f\n", d.oa_n,
                            d.oa_t[0] / d.oa_n, d.oa_t[1] / d.oa_n, d.oa_t[2] / d.oa_n, d.oa_t[3] / d.oa_n, d.oa_t[4] / d.oa_n, d.oa_t[5] / d.oa_n, d.oa_t[6] / d.oa_n, 0.0);
#endif
    if (getenv("RH_DRIVER_PROF")) {
        fprintf(stderr, "[rh_ransac] init %.4f loop %.4f tail %.4f s\n", t_init, t_loop, out->seconds - t_init - t_loop);
        fprintf(stderr, "[rh_ransac] %lld windows: enqueue %.4f wait %.4f lists %.4f record %.4f s; total %.4f\n", (long long)d.nwin,
                d.tw[0], d.tw[1], d.tw[2], d.tw[3], out->seconds);
        fprintf(stderr, "[rh_ransac] extract: refit+invalidate %.4f (list copy call %.4f) erase %.4f liveness %.4f store-compact %.4f host-compact %.4f s\n",
                d.tp[0], d.tp[5], d.tp[1], d.tp[2], d.tp[3], d.tp[4]);
    }
    return RH_OK;
}
