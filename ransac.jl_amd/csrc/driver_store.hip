// driver_store.hip -- lifetimes of what a run of rh_ransac allocates: the device store of prepared candidates, the
// sampled windows, the pinned staging ring, the buffers parked on the cloud between runs, and the result arenas
#include "driver_internal.h"

namespace rhdrv {

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

// ---- result arenas ---------------------------------------------------------------------------
// The index lists of a run (<= 8 bytes x the points enabled at its start) land in ONE pinned host
// block: the D2H copies are asynchronous at PCIe rate and nothing is copied a second time
// (pageable destinations cost ~0.25 ms per extracted shape at 10M points).  Pinning is slow, so
// blocks are recycled through a small process-wide pool: rh_result_free hands the block back.
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

void pin_ring_free(PinRing &r)
{
    for (int i = 0; i < 4; i++) {
        if (r.busy[i] && r.ev[i]) (void)hipEventSynchronize(r.ev[i]);
        if (r.buf[i]) (void)hipHostFree(r.buf[i]);
        if (r.ev[i]) (void)hipEventDestroy(r.ev[i]);
        r.buf[i] = nullptr; r.cap[i] = 0; r.ev[i] = nullptr; r.busy[i] = false;
    }
}

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

}  // namespace rhdrv

using namespace rhdrv;

extern "C" void rh_result_free(rh_result *r)
{
    if (!r) return;
    arena_release(r->arena);
    free(r->shapes);
    memset(r, 0, sizeof *r);
}
