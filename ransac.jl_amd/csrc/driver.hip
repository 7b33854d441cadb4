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
//
// Units: driver_internal.h lists them.
#include "driver_internal.h"

namespace rhdrv {

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

Driver::~Driver()
{
    if (c) (void)hipStreamSynchronize(c->stream);   // no copy may still be landing in the pinned blocks
    if (c) (void)hipStreamSynchronize(c->copy_stream);
    arena_release(arena);   // null once the result owns it
    if (c && clean && c->drv_cache == nullptr && !rh_opt_on(c, RH_OPT_NO_DRIVER_CACHE)) {
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
int Driver::ensure_scratch(int64_t ints)
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

int Driver::init()
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

void Driver::rebuild_mprefix()
{
    mprefix.resize((size_t)c->nwords + 1);
    int32_t acc = 0;
    for (int64_t w = 0; w < c->nwords; w++) { mprefix[(size_t)w] = acc; acc += __builtin_popcountll(men[(size_t)w]); }
    mprefix[(size_t)c->nwords] = acc;
}

// forcefitshapes! (fitting.jl:165-173) for one sampled minimal set
int Driver::fit_set(std::vector<rh_shape> &cands)
{
    for (int q = 0; q < drawN; q++) {
        if (xyz32 != nullptr) {   // rh_ransac_f32: the sampled rows of the caller's Float32 arrays, converted exactly
            for (int a = 0; a < 3; a++) {
                fp[3 * (size_t)q + a] = (double)xyz32[3 * (sd[(size_t)q] - 1) + a];
                fn[3 * (size_t)q + a] = (double)nrm32[3 * (sd[(size_t)q] - 1) + a];
            }
            continue;
        }
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
int Driver::sample_iteration_host(int64_t k, std::vector<rh_shape> &cands, std::vector<int32_t> &levels)
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
int Driver::score(const rh_shape *cands, int32_t ncand, std::vector<int32_t> &counts)
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
int Driver::record(const rh_shape *cands, const int32_t *levels, int32_t ncand, const int32_t *counts, const int32_t *dev_slots)
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

// everything of iteration k after the candidates exist: iterations.jl:98-156.
// Returns through *stop whether the loop ends after this iteration.
int Driver::finish_iteration(int64_t k, const rh_shape *cands, const int32_t *levels, int32_t ncand, const int32_t *counts,
                     bool *did_extract, bool *stop, const int32_t *dev_slots)
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

int Driver::run_sequential()
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

}  // namespace rhdrv

using namespace rhdrv;

static int ransac_impl(rh_cloud *c, const double *xyz, const double *nrm, const rh_params *p, rh_rng *rng, rh_mp *mp,
                       rh_result *out, const float *xyz32 = nullptr, const float *nrm32 = nullptr);

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
    // (no copy of the cloud: only the host-side fits of sampling_streams = 0 read points, and they convert the rows they sample)
    return ransac_impl(c, nullptr, nullptr, p, rng, nullptr, out, xyz, nrm);
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
                       rh_result *out, const float *xyz32, const float *nrm32)
{
    if (!c || !p || !rng || !out) { rh_set_error("rh_ransac: NULL argument"); return RH_E_INVALID; }
    memset(out, 0, sizeof *out);
    RH_TRY(rh_validate_params(p));
    if (c->n > 0 && (!xyz || !nrm) && (!xyz32 || !nrm32)) { rh_set_error("rh_ransac: xyz/nrm are NULL"); return RH_E_INVALID; }
    if (c->f32 && mp != nullptr) {   // (the ranks' exchange has only ever been held against the single-GPU run on Float64 clouds)
        rh_set_error("rh_ransac_mp is not available on a Float32 cloud");
        return RH_E_INVALID;
    }
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
    RH_TRY(rh_join_batches(c));
    const double t_start = now_s();

    bool device_sampler = p->sampling_streams != 0 && p->drawN <= 8 && p->minsubsetN > 0 && c->n > 0;
    if (rh_opt_on(c, RH_OPT_HOST_SAMPLER)) device_sampler = false;   // A/B and tests: the same streams drawn on the host
    if (mp != nullptr && !device_sampler) {
        rh_set_error("rh_ransac_mp: the minimal sets are dealt to the ranks by the device sampler (drawN <= 8, minsubsetN > 0, device sampler not switched off)");
        return RH_E_INVALID;
    }
    Driver d;
    d.c = c; d.p = p; d.xyz = xyz; d.nrm = nrm; d.rng = rng;
    d.xyz32 = xyz == nullptr ? xyz32 : nullptr; d.nrm32 = xyz == nullptr ? nrm32 : nullptr;
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
    if (rh_opt_on(c, RH_OPT_DRIVER_PROF)) {
        fprintf(stderr, "[rh_ransac] init %.4f loop %.4f tail %.4f s\n", t_init, t_loop, out->seconds - t_init - t_loop);
        fprintf(stderr, "[rh_ransac] %lld windows: enqueue %.4f wait %.4f lists %.4f record %.4f s; total %.4f\n", (long long)d.nwin,
                d.tw[0], d.tw[1], d.tw[2], d.tw[3], out->seconds);
        fprintf(stderr, "[rh_ransac] extract: refit+invalidate %.4f (list copy call %.4f) erase %.4f liveness %.4f store-compact %.4f host-compact %.4f s\n",
                d.tp[0], d.tp[5], d.tp[1], d.tp[2], d.tp[3], d.tp[4]);
    }
    return RH_OK;
}

