// driver_windows.hip -- sampling_streams = 1: windows of iterations sampled, fitted and scored on the device ahead of the
// host's replay; octree sampling: chained windows (every iteration queued whole, the level update on the device)
#include "driver_internal.h"

namespace rhdrv {

// Octree windows, one process: CHAINED.  Every iteration's scores change the level distribution the next
// one samples from (fitting.jl:184, octree.jl:198-205), so nothing can be sampled ahead.  Instead the whole
// iteration -- sampling, fits, scoring, the level update and the copy of its candidates to the host
// (rhk_oct_advance) -- is queued W times back to back, with an event behind each, and the host replays iteration
// i while the device runs i + 1, ...: recordscore!, the extraction test, updatelevelweight, checking the
// device's level distribution against its own bit for bit.  The device ends the window (stop flag: the remaining
// launches return at once) at the first iteration whose extraction test passes in its arithmetic; the decision
// is the host's.
int Driver::run_chained_windows(const size_t status_bytes)
{
    int32_t cnt_est = 64;
    std::vector<rh_shape> cands;
    std::vector<int32_t> counts, levels, wcounts, order, wslots;
    const int T = p->n_shape_types;
    int64_t Kchain = 8;
    if (const int64_t e = rh_opt_int(c, RH_OPT_OCT_CHAIN_W, 0)) Kchain = std::max<int64_t>(1, std::min<int64_t>(e, RH_CHAIN_MAX));
    if (c->oct_state == nullptr) RUNH(hipMalloc((void **)&c->oct_state, sizeof(rh_oct_state)));
    managed = !rh_opt_on(c, RH_OPT_NO_MANAGED_STORE);   // (the store is empty here: run_streams_device is where a run starts)
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
    const int64_t bound_pct = 200;
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
    const int max_flight = rh_opt_on(c, RH_OPT_OCT_ONE_WINDOW) ? 1 : 2;
    // (iterations per window: the launches of Kchain iterations are in the queue at most, whatever the number of
    // windows they are cut into -- a deeper queue makes the launches themselves slow)
    int64_t Wfl = std::max<int64_t>(1, max_flight == 2 ? (Kchain * 3) / 8 : Kchain);   // (8 -> two windows of 3: swept 2 / 3 / 4 / 6 -> 0.0482 / 0.0474 / 0.0484 / 0.0492 s)
    if (const int64_t e = rh_opt_int(c, RH_OPT_OCT_WINDOW_ITERS, 0)) Wfl = std::max<int64_t>(1, std::min<int64_t>(e, RH_CHAIN_MAX));
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

// sampling_streams = 1 with every shape type fittable on the device: iterations are sampled,
// fitted and scored SPECULATIVELY in windows (the enabled bits only change at an extraction, and
// a set's draws are a pure function of (seed, k, j)); the host replays the window in order and,
// when an extraction happens at iteration kk, throws the rest of the window away and resumes at
// kk + 1 -- bit-identical to the sequential loop.
int Driver::run_streams_device()
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
    const bool fused_score = rh_score_v4_enabled(c) && !rh_opt_on(c, RH_OPT_NO_FUSED_SCORE);
    int32_t cnt_est = 64;
    // Without the octree a window's draws depend only on (seed, k, j) and the enabled bits, so the
    // NEXT window is put on the stream before the host waits for this one: it is valid unless this
    // one ends in an extraction (then it is dropped and drawn again).  The GPU samples window
    // w + 1 while the host replays window w.
    const bool pipeline = !octree && !rh_opt_on(c, RH_OPT_NO_PIPELINE);
    std::vector<rh_cand_entry> entries;
    std::vector<rh_shape> cands;
    std::vector<int32_t> counts, levels, wcounts, order, wslots;
    std::vector<int64_t> slots;
    const int T = p->n_shape_types;
    // (octree windows of one process are CHAINED: run_chained_windows)
    const bool chain = octree && fused_score && mp == nullptr && !rh_opt_on(c, RH_OPT_NO_OCT_CHAIN);
    if (chain) return run_chained_windows(status_bytes);
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

}  // namespace rhdrv
