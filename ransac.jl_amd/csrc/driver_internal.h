// driver_internal.h -- what the units of the rh_ransac driver share: the loop state (struct Driver), the device store of
// prepared candidates, the sampled windows, the pinned scratch and the node-local exchange of rh_ransac_mp.
//   driver.hip          entry points, set-up, the sequential loop, host-side sampling, recordscore! (record)
//   driver_extract.hip  the extraction step: refit, invalidate, liveness of the store (maybe_extract)
//   driver_windows.hip  speculated windows of iterations on the device, chained octree windows (run_streams_device)
//   driver_store.hip    device store / window / pinned-block lifetimes, result arenas
//   mp.hip              rh_mp_*: the shared-memory exchange of the processes of one node
#pragma once

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "fit_shared.h"
#include "rh_internal.h"

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

namespace rhdrv {

double now_s();

int mp_exchange_any(rh_mp *m, const void *payload, int64_t bytes, std::vector<std::vector<char>> &recv);   // mp.hip

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

int store_free(rh_cloud *c, DeviceStore &st);                       // driver_store.hip
int store_reserve(rh_cloud *c, DeviceStore &st, int kind, int64_t need);
int store_reserve_aux(rh_cloud *c, DeviceStore &st, int64_t need);
void *arena_acquire(size_t bytes);                                   // result arenas (driver_store.hip)
void arena_release(void *p);

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

void window_free(Window &w);

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

void pin_ring_free(PinRing &r);

struct DriverCache {
    Window win[2];
    DeviceStore st;
    int32_t *h_scr = nullptr;
    int64_t h_scr_cap = 0;
    PinRing ring;
};

void driver_cache_free(rh_cloud *c, void *p);

struct Driver {
    rh_cloud *c;
    const rh_params *p;
    const double *xyz, *nrm;
    const float *xyz32 = nullptr, *nrm32 = nullptr;   // rh_ransac_f32: the caller's Float32 arrays (xyz / nrm are null then); only the host-side fits read points
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

    ~Driver();

    // pinned scratch of at least `ints` int32 (contents are not preserved when it grows)
    int ensure_scratch(int64_t ints);

    int init();

    void rebuild_mprefix();

    rhfit::OctView host_octview() const
    {
        rhfit::OctView oc;
        oc.code = c->h_oct_code.data(); oc.perm = c->h_oct_perm.data(); oc.pos = c->h_oct_pos.data();
        oc.men = men.data(); oc.prefix = mprefix.data();
        oc.n = c->n; oc.nwords = c->nwords; oc.depth = od;
        return oc;
    }

    // forcefitshapes! (fitting.jl:165-173) for one sampled minimal set
    int fit_set(std::vector<rh_shape> &cands);

    // one iteration's minimal sets on the host: sequential stream (mode 0) or per-set streams (mode 1)
    int sample_iteration_host(int64_t k, std::vector<rh_shape> &cands, std::vector<int32_t> &levels);

    // scorecandidates! (fitting.jl:181-190) for a batch: counts in candidate order -- the ABI's own batched call
    // (one launch for all kinds; batches of a few candidates travel as one staged transfer).  Nothing reads a
    // score before the loop ends (iterations.jl:99).
    int score(const rh_shape *cands, int32_t ncand, std::vector<int32_t> &counts);

    // recordscore! (fitting.jl:114-119) in candidate order + prepared records into the device store
    // dev_slots (chained octree windows): the device has appended the candidates' records to the store itself
    // (rhk_oct_advance) -- dev_slots[i] is candidate i's slot in the store of its kind
    int record(const rh_shape *cands, const int32_t *levels, int32_t ncand, const int32_t *counts, const int32_t *dev_slots = nullptr);

    // iterations.jl:106-140: extract the best candidate if its detection probability is high enough
    int maybe_extract(int64_t k, bool *did);
    int prune_managed_store(size_t extracted_pos, int64_t sum_n, const int32_t pbase[5], int32_t *h_nk, int32_t *h_counts,
                            int64_t ndis_old, int32_t ndis_new, double t0, double tq, bool *did);   // driver_extract.hip

    // everything of iteration k after the candidates exist: iterations.jl:98-156.
    // Returns through *stop whether the loop ends after this iteration.
    int finish_iteration(int64_t k, const rh_shape *cands, const int32_t *levels, int32_t ncand, const int32_t *counts,
                         bool *did_extract, bool *stop, const int32_t *dev_slots = nullptr);

    int run_sequential();

    // sampling_streams = 1 with every shape type fittable on the device: iterations are sampled,
    // fitted and scored SPECULATIVELY in windows (the enabled bits only change at an extraction, and
    // a set's draws are a pure function of (seed, k, j)); the host replays the window in order and,
    // when an extraction happens at iteration kk, throws the rest of the window away and resumes at
    // kk + 1 -- bit-identical to the sequential loop.
    int run_streams_device();
    int run_chained_windows(size_t status_bytes);   // octree sampling, one process (driver_windows.hip)
};

}  // namespace rhdrv
