// driver_extract.hip -- the extraction step of rh_ransac (iterations.jl:106-147): the best candidate's refit over the
// enabled cloud, invalidate_indexes!, and removeinvalidshapes! as a recomputed liveness pass over the candidate store
#include "driver_internal.h"

namespace rhdrv {

// removeinvalidshapes! (fitting.jl:209-221) with the store managed on the device (chained octree windows): liveness counts
// per entry over the new part of the disabled list, then rhk_store_compact moves the survivors to the spare arrays, names
// the best of them and hands the dead candidates' numbers over -- one wait, and the host touches only the dead.
// pbase: the kinds laid end to end, each padded to RH_STORE_PAD; h_nk / h_counts: pinned scratch (maybe_extract)
int Driver::prune_managed_store(const size_t extracted_pos, const int64_t sum_n, const int32_t pbase[5], int32_t *h_nk, int32_t *h_counts,
                                const int64_t ndis_old, const int32_t ndis_new, const double t0, double tq, bool *did)
{
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

// iterations.jl:106-140: extract the best candidate if its detection probability is high enough
int Driver::maybe_extract(int64_t k, bool *did)
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
    const bool fast = !managed && sum_n <= LIVE_MAX && live_work <= ((int64_t)1 << 24) && !rh_opt_on(c, RH_OPT_NO_FAST_EXTRACT);
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
    if (managed) return prune_managed_store(extracted_pos, sum_n, pbase, h_nk, h_counts, ndis_old, ndis_new, t0, tq, did);
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

}  // namespace rhdrv
