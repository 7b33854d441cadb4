// cloud.hip -- C-ABI entry points around the device-resident cloud (include/ransac_hip.h).
#include <functional>
#include <system_error>
#include <thread>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <utility>
#include <vector>

#include "rh_internal.h"
#include "score4_device.h"

// ---------------------------------------------------------------- errors ----
static thread_local char g_err[512] = "";

void rh_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *rh_last_error(void) { return g_err; }
extern "C" int rh_version(void) { return RH_VERSION; }

extern "C" int rh_device_count(int *n_out)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_out = 0;
        rh_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return RH_E_NODEVICE;
    }
    *n_out = n;
    return RH_OK;
}

int rh_validate_params(const rh_params *p)
{
    if (!p) { rh_set_error("params is NULL"); return RH_E_INVALID; }
    for (int k = 0; k < 4; k++)
        if (!(p->eps[k] == p->eps[k]) || !(p->cos_alpha[k] == p->cos_alpha[k])) {
            rh_set_error("params: eps/cos_alpha of kind %d is NaN", k);
            return RH_E_INVALID;
        }
    return RH_OK;
}

// ------------------------------------------------------------ allocation ----
template <typename T>
static int dev_alloc(T **p, int64_t count)
{
    *p = nullptr;
    const size_t bytes = sizeof(T) * (size_t)(count > 0 ? count : 1);
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipSuccess) {
        rh_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RH_E_NOMEM : RH_E_NODEVICE;
    }
    return RH_OK;
}

int rh_ensure_pin(rh_cloud *c, int64_t bytes)
{
    if (bytes <= c->h_pin_cap) return RH_OK;
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    c->h_pin = nullptr;
    c->h_pin_cap = 0;
    int64_t cap = std::max<int64_t>(bytes, 1 << 20);
    RH_HIP(hipHostMalloc(&c->h_pin, (size_t)cap, hipHostMallocDefault));
    c->h_pin_cap = cap;
    return RH_OK;
}

int rh_ensure_batch(rh_cloud *c, int64_t b)
{
    if (b <= c->batch_cap) return RH_OK;
    RH_HIP(hipStreamSynchronize(c->stream));
    int64_t cap = std::max<int64_t>(b, std::max<int64_t>(1024, c->batch_cap * 2));
    (void)hipFree(c->d_shapes); (void)hipFree(c->d_prep); (void)hipFree(c->d_orig); (void)hipFree(c->d_counts);
    c->d_shapes = nullptr; c->d_prep = nullptr; c->d_orig = nullptr; c->d_counts = nullptr;
    c->batch_cap = 0;
    RH_TRY(dev_alloc(&c->d_shapes, cap));
    RH_TRY(dev_alloc(&c->d_prep, 4 * cap));
    (void)hipFree(c->d_qpre);
    c->d_qpre = nullptr;
    RH_HIP(hipMalloc(&c->d_qpre, (size_t)(4 * cap) * 64));   // rhdev::rh_pre (40 B) or rh4::rh_cls (64 B) per slot
    (void)hipFree(c->d_box);
    c->d_box = nullptr;
    RH_TRY(dev_alloc(&c->d_box, (int64_t)rh4::RH_BOX_FIELDS * 4 * cap));
    RH_TRY(dev_alloc(&c->d_orig, 4 * cap));
    RH_TRY(dev_alloc(&c->d_counts, cap));
    if (c->f32) {
        (void)hipFree(c->d_prep32);
        c->d_prep32 = nullptr;
        RH_HIP(hipMalloc(&c->d_prep32, (size_t)(4 * cap) * 12 * sizeof(float)));
    }
    c->batch_cap = cap;
    return RH_OK;
}

int rh_ensure_masks(rh_cloud *c, int64_t words)
{
    if (words <= c->masks_cap) return RH_OK;
    RH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_masks);
    c->d_masks = nullptr;
    c->masks_cap = 0;
    RH_TRY(dev_alloc(&c->d_masks, words));
    c->masks_cap = words;
    return RH_OK;
}

// v4 score kernel with masks: per candidate row a list of (internal word number, inlier word) entries, 16 bytes each,
// at most one per group (mstride4 of them), and one int32 cursor per row, zero between batches
static int ensure_masks4(rh_cloud *c, int64_t b)
{
    c->mstride4 = (c->ngroups + 7) / 8 * 8;
    const int64_t words = 2 * b * c->mstride4;      // (two 64-bit words per entry)
    if (words > c->masks_int_cap) {
        RH_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_masks_int);
        c->d_masks_int = nullptr;
        c->masks_int_cap = 0;
        RH_TRY(dev_alloc(&c->d_masks_int, words));
        c->masks_int_cap = words;
    }
    const int64_t cur_bytes = 4 * b;
    if (cur_bytes > c->occ_cap) {
        RH_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_occ);
        c->d_occ = nullptr;
        c->occ_cap = 0;
        RH_TRY(dev_alloc(&c->d_occ, cur_bytes));
        RH_HIP(hipMemsetAsync(c->d_occ, 0, (size_t)cur_bytes, c->stream));
        c->occ_cap = cur_bytes;
    }
    return RH_OK;
}

static int ensure_masks_int(rh_cloud *c, int64_t words)
{
    if (words <= c->masks_int_cap) return RH_OK;
    RH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_masks_int);
    c->d_masks_int = nullptr;
    c->masks_int_cap = 0;
    RH_TRY(dev_alloc(&c->d_masks_int, words));
    c->masks_int_cap = words;
    return RH_OK;
}

// 63-bit Morton code of a point inside the subset's bounding box (21 bits per axis)
static inline uint64_t spread21(uint64_t v)
{
    v &= 0x1FFFFFULL;
    v = (v | (v << 32)) & 0x1F00000000FFFFULL;
    v = (v | (v << 16)) & 0x1F0000FF0000FFULL;
    v = (v | (v << 8)) & 0x100F00F00F00F00FULL;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}

static void cloud_free(rh_cloud *c)
{
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    // a caller's stream (rh_cloud_set_stream) may be gone by now: wait for the device instead of the handle
    if (c->stream != c->own_stream) (void)hipDeviceSynchronize();
    else if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    for (rh_batch_slot &s : c->alt) {
        if (!s.stream) continue;
        (void)hipStreamSynchronize(s.stream);
        (void)hipFree(s.d_shapes); (void)hipFree(s.d_prep); (void)hipFree(s.d_orig); (void)hipFree(s.d_counts); (void)hipFree(s.d_nk2);
        (void)hipFree(s.d_qpre); (void)hipFree(s.d_prep32); (void)hipFree(s.d_box); (void)hipFree(s.d_masks_int); (void)hipFree(s.d_occ); (void)hipFree(s.d_stlist); (void)hipFree(s.d_stcount);
        (void)hipEventDestroy(s.done); (void)hipEventDestroy(s.start); (void)hipStreamDestroy(s.stream);
    }
    (void)hipFree(c->full); (void)hipFree(c->rec); (void)hipFree(c->crec); (void)hipFree(c->sel_list); (void)hipFree(c->set_ws); (void)hipFree(c->set_level); (void)hipFree(c->sub); (void)hipFree(c->dis);
    (void)hipFree(c->sub_idx0); (void)hipFree(c->enabled); (void)hipFree(c->sub_enabled);
    (void)hipFree(c->sub_perm); (void)hipFree(c->gb); (void)hipFree(c->d_masks_int);
    (void)hipFree(c->full32); (void)hipFree(c->sub32); (void)hipFree(c->d_prep32); (void)hipFree(c->d_qpre); (void)hipFree(c->d_zero);
    (void)hipFree(c->s4_stats); (void)hipFree(c->d_box); (void)hipFree(c->gb32); (void)hipFree(c->st32); (void)hipFree(c->d_stlist); (void)hipFree(c->d_stcount); (void)hipFree(c->d_occ); (void)hipFree(c->unp_segmask);
    (void)hipFree(c->oct_code); (void)hipFree(c->oct_perm); (void)hipFree(c->oct_pos); (void)hipFree(c->oct_men);
    (void)hipFree(c->oct_prefix); (void)hipFree(c->oct_P); (void)hipFree(c->oct_tab); (void)hipFree(c->oct_code_o);
    (void)hipFree(c->oct_state); (void)hipFree(c->oct_adv_tab); (void)hipFree(c->oct_adv_bits); (void)hipFree(c->oct_adv_E);
    (void)hipFree(c->fullk); (void)hipFree(c->fullk32); (void)hipFree(c->kgb); (void)hipFree(c->klist); (void)hipFree(c->kctr); (void)hipFree(c->kflag);
    (void)hipFree(c->en_block_sums); (void)hipFree(c->dis_gb); (void)hipFree(c->dis_gb32);
    (void)hipFree(c->d_ndis); (void)hipFree(c->refit_mask); (void)hipFree(c->block_sums);
    (void)hipFree(c->word_prefix); (void)hipFree(c->idx_out); (void)hipFree(c->d_total);
    (void)hipFree(c->d_shapes); (void)hipFree(c->d_prep); (void)hipFree(c->d_orig); (void)hipFree(c->d_nk); (void)hipFree(c->d_nk2);
    (void)hipFree(c->d_counts); (void)hipFree(c->d_masks); (void)hipFree(c->d_ranks); (void)hipFree(c->d_stage);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int k = 0; k < 5; k++)
        if (c->evk[k]) (void)hipEventDestroy(c->evk[k]);
    if (c->ev_cull) (void)hipEventDestroy(c->ev_cull);
    c->stream = c->own_stream;   // (the deleter below waits on c->stream; everything has finished by now)
    if (c->drv_cache != nullptr && c->drv_cache_free != nullptr) c->drv_cache_free(c, c->drv_cache);
    c->drv_cache = nullptr;
    if (c->ev_copied) (void)hipEventDestroy(c->ev_copied);
    if (c->ev_sync) (void)hipEventDestroy(c->ev_sync);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

static int set_all_enabled(rh_cloud *c)
{
    if (c->nwords > 0) {
        RH_HIP(hipMemsetAsync(c->enabled, 0xFF, sizeof(uint64_t) * (size_t)c->nwords, c->stream));
        if (c->n % 64) {
            const uint64_t last = (~0ULL) >> (64 - c->n % 64);
            RH_HIP(hipMemcpyAsync(c->enabled + (c->nwords - 1), &last, sizeof last, hipMemcpyHostToDevice, c->stream));
            RH_HIP(hipStreamSynchronize(c->stream));
        }
        if (c->k_built) {   // the same n bits in Morton order
            RH_HIP(hipMemcpyAsync(c->oct_men, c->enabled, sizeof(uint64_t) * (size_t)c->nwords, hipMemcpyDeviceToDevice, c->stream));
            c->k_men_valid = true;
        }
    }
    RH_TRY(rhk_rebuild_sub_enabled(c, true));
    c->select_valid = false;
    c->en_sums_valid = false;
    c->n_dis = 0;
    return RH_OK;
}

// bounding cube of the cloud like the reference's octree (findAABB: src/utilities.jl:125-136; the cube's edge is the
// largest extent) and the largest finite |coordinate|
static void cloud_aabb(const double *xyz, int64_t n, double lo[3], double *size_out, double *mag_out)
{
    double hi[3] = { 0, 0, 0 }, mag = 0;
    for (int k = 0; k < 3; k++) lo[k] = 0;
    for (int k = 0; k < 3 && n > 0; k++) { lo[k] = xyz[k]; hi[k] = xyz[k]; }
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            const double a = xyz[3 * i + k];
            lo[k] = lo[k] > a ? a : lo[k];
            hi[k] = hi[k] < a ? a : hi[k];
            const double m = fabs(a);
            if (m > mag && m - m == 0) mag = m;
        }
    double size = 0;
    for (int k = 0; k < 3; k++) if (hi[k] - lo[k] > size) size = hi[k] - lo[k];
    size = size * (1 + 1e-9);
    if (!(size > 0)) size = 1;
    *size_out = size;
    *mag_out = mag;
}

// the same box from the uploaded copy (rhk_cloud_aabb, kdorder.hip).  The host loop's lo / hi start at the first point and
// move by plain comparisons, so a NaN in the first point stays (every comparison with it is false) and any other NaN is
// skipped; infinities take part.  Same values, bit for bit: the Morton codes (and the oracle's octree) hang on them.
static int cloud_aabb_device(rh_cloud *c, const double *d_xyz, const double *xyz, int64_t n)
{
    double lo[3], hi[3], mag = 0;
    bool has[3];
    RH_TRY(rhk_cloud_aabb(c, d_xyz, n, lo, hi, has, &mag));
    for (int k = 0; k < 3; k++) {
        const double first = n > 0 ? xyz[k] : 0.0;
        if (n > 0 && !(first == first)) { lo[k] = first; hi[k] = first; }   // a NaN first point: nothing ever replaces it
        else if (!has[k]) { lo[k] = 0; hi[k] = 0; }
        c->k_lo[k] = lo[k];
    }
    double size = 0;
    for (int k = 0; k < 3; k++) if (hi[k] - lo[k] > size) size = hi[k] - lo[k];
    size = size * (1 + 1e-9);
    if (!(size > 0)) size = 1;
    c->k_size = size;
    c->k_mag = mag;
    return RH_OK;
}

// linear (Morton) octree of the full cloud: codes in the bounding CUBE, sorted by (code, index) -- the Morton order the
// cloud was given on the device when it was created (korder.hip); here: the host twins and
// depth = first level whose fullest cell holds <= 8 points (src/octree.jl:163-165), capped.
int rh_octree_ensure(rh_cloud *c, const double *xyz, int max_depth)
{
    (void)xyz;
    if (max_depth < 1) max_depth = 1;
    if (max_depth > 21) max_depth = 21;
    if (c->oct_built && c->oct_max_depth == max_depth) return rhk_oct_sync_enabled(c);
    const int64_t n = c->n;
    if (n > 0 && !c->k_built) { rh_set_error("rh_octree_ensure: the cloud has no Morton order"); return RH_E_INTERNAL; }
    c->h_oct_code.resize((size_t)n);
    c->h_oct_perm.resize((size_t)n);
    c->h_oct_pos.resize((size_t)n);
    if (n > 0) {
        RH_HIP(hipMemcpyAsync(c->h_oct_code.data(), c->oct_code, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipMemcpyAsync(c->h_oct_perm.data(), c->oct_perm, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipMemcpyAsync(c->h_oct_pos.data(), c->oct_pos, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipStreamSynchronize(c->stream));
    }
    int depth = max_depth;
    for (int l = 1; l <= max_depth; l++) {
        const int shift = 3 * (21 - (l - 1));
        int64_t run = 0, best = 0;
        uint64_t prev = 0;
        for (int64_t i = 0; i < n; i++) {
            const uint64_t key = shift >= 63 ? 0 : (c->h_oct_code[(size_t)i] >> shift);
            if (i == 0 || key != prev) { run = 0; prev = key; }
            if (++run > best) best = run;
        }
        if (best <= 8) { depth = l; break; }
    }
    c->oct_depth = depth;
    c->oct_max_depth = max_depth;
    RH_TRY(rhk_oct_build_tab(c));
    c->oct_built = true;
    return rhk_oct_sync_enabled(c);
}

// ------------------------------------------------------------------ cloud ----
extern "C" int rh_cloud_create(const double *xyz, const double *nrm, int64_t n, const int64_t *subset1, int64_t s,
                               int device, rh_cloud **out)
{
    if (!out) { rh_set_error("out is NULL"); return RH_E_INVALID; }
    *out = nullptr;
    if (n < 0 || s < 0 || (n > 0 && (!xyz || !nrm)) || (s > 0 && !subset1)) {
        rh_set_error("rh_cloud_create: bad arguments (n=%lld s=%lld)", (long long)n, (long long)s);
        return RH_E_INVALID;
    }
    if (n > (int64_t)0x7FFFF000) { rh_set_error("clouds above 2^31 points are not supported"); return RH_E_INVALID; }
    for (int64_t j = 0; j < s; j++)
        if (subset1[j] < 1 || subset1[j] > n) {
            rh_set_error("subset index %lld at position %lld outside 1..%lld", (long long)subset1[j], (long long)j,
                         (long long)n);
            return RH_E_INVALID;
        }
    if (n > 2000000000LL) {   // internal indices (subset gather, select list, octree permutation) are int32
        rh_set_error("rh_cloud_create: %lld points exceed the 2e9 limit of one cloud", (long long)n);
        return RH_E_INVALID;
    }
    int ndev = 0;
    RH_TRY(rh_device_count(&ndev));
    if (ndev <= 0) { rh_set_error("no HIP device is visible; libransac_hip has no CPU fallback"); return RH_E_NODEVICE; }
    if (device < 0 || device >= ndev) { rh_set_error("device %d out of range (%d visible)", device, ndev); return RH_E_INVALID; }
    RH_HIP(hipSetDevice(device));
    const auto tc0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double ms_kd = 0, ms_before_kd = 0;
    // RH_CREATE_PROF=1: wall time of every stage on stderr (each stamp waits for the stream: the stages stop overlapping)
    const bool prof = rh_opt_on(nullptr, RH_OPT_CREATE_PROF);
    double t_prev = 0;

    rh_cloud *c = new (std::nothrow) rh_cloud();
    if (!c) { rh_set_error("out of host memory"); return RH_E_NOMEM; }
    rh_opt_init_cloud(c);
    c->device = device;
    c->n = n;
    c->s = s;
    c->n_pad = ((n + RH_SC_TILE - 1) / RH_SC_TILE) * RH_SC_TILE;
    c->s_pad = ((s + RH_SC_TILE - 1) / RH_SC_TILE) * RH_SC_TILE;
    if (c->n_pad == 0) c->n_pad = RH_SC_TILE;
    if (c->s_pad == 0) c->s_pad = RH_SC_TILE;
    c->nwords = (n + 63) / 64;
    c->swords = (s + 63) / 64;
    c->nblocks = (c->nwords + RH_WORDS_PER_BLOCK - 1) / RH_WORDS_PER_BLOCK;
    c->dis_stride = c->s_pad + RH_SC_TILE;

    int rc = RH_OK;
    double *t_xyz = nullptr, *t_nrm = nullptr;
    int32_t *h_idx = nullptr;
    auto stamp = [&](const char *what) {
        if (!prof) return;
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        const double t = ms_since(tc0);
        fprintf(stderr, "[rh_cloud_create] %-28s %8.2f ms\n", what, t - t_prev);
        t_prev = t;
    };
    auto fail = [&](int code) {
        (void)hipFree(t_xyz); (void)hipFree(t_nrm);
        delete[] h_idx;
        cloud_free(c);
        return code;
    };
#define CK(x) do { rc = (x); if (rc != RH_OK) return fail(rc); } while (0)
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rh_set_error("%s: %s", #x, hipGetErrorString(e_)); return fail(RH_E_NODEVICE); } } while (0)
    CKH(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CKH(hipEventCreate(&c->ev0));
    CKH(hipEventCreate(&c->ev1));
    for (int k = 0; k < 5; k++) CKH(hipEventCreate(&c->evk[k]));
    CKH(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    CKH(hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming));
    CKH(hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming));
    CK(dev_alloc(&c->full, 6 * c->n_pad));
    CK(dev_alloc(&c->rec, 8 * std::max<int64_t>(n, 1)));
    CK(dev_alloc(&c->sel_list, c->nwords * 64 + 64));
    CK(dev_alloc(&c->sub, 6 * c->s_pad));
    CK(dev_alloc(&c->dis, 6 * c->dis_stride));
    CK(dev_alloc(&c->sub_idx0, s));
    CK(dev_alloc(&c->sub_perm, c->s_pad));   // (whole 64-lane reads of the last group stay inside the allocation: the mask pass, score4.hip)
    CKH(hipMemsetAsync(c->sub_perm, 0, sizeof(int32_t) * (size_t)std::max<int64_t>(c->s_pad, 1), c->stream));
    c->ngroups = (s + 63) / 64;
    c->ng_pad = ((c->ngroups + RH_G2_TG - 1) / RH_G2_TG) * RH_G2_TG + RH_G2_TG;
    CK(dev_alloc(&c->gb, 7 * c->ng_pad));
    CK(dev_alloc(&c->gb32, 8 * c->ng_pad));
    c->nst = (c->ngroups + 15) / 16;
    CK(dev_alloc(&c->st32, 8 * std::max<int64_t>(1, c->nst)));
    CK(dev_alloc(&c->dis_gb, 7 * c->ng_pad));
    CK(dev_alloc(&c->dis_gb32, 8 * c->ng_pad));
    CKH(hipMemsetAsync(c->dis_gb, 0, sizeof(double) * 7 * (size_t)c->ng_pad, c->stream));
    CKH(hipMemsetAsync(c->gb, 0, sizeof(double) * 7 * (size_t)c->ng_pad, c->stream));
    {
        const int64_t e = rh_opt_int(nullptr, RH_OPT_SCORE_PATH, RH_SCORE_PATH_AUTO);   // rh_set_option(NULL, "score_path", ...): force a path (tests, A/B runs)
        c->use_groups = s >= RH_G2_MIN_POINTS;
        if (e == RH_SCORE_PATH_BRUTE) c->use_groups = false;
        if (e == RH_SCORE_PATH_GROUPS) c->use_groups = s > 0;
    }
    CK(dev_alloc(&c->enabled, c->nwords));
    CK(dev_alloc(&c->sub_enabled, c->swords));
    CK(dev_alloc(&c->d_ndis, 1));
    CK(dev_alloc(&c->refit_mask, c->nwords));
    CK(dev_alloc(&c->block_sums, std::max<int64_t>(c->nblocks, (c->swords + RH_WORDS_PER_BLOCK - 1) / RH_WORDS_PER_BLOCK) + 2));
    CK(dev_alloc(&c->en_block_sums, c->nblocks + 2));
    CK(dev_alloc(&c->word_prefix, c->nwords + 1));
    CK(dev_alloc(&c->idx_out, n));
    CK(dev_alloc(&c->d_total, 1));
    CK(dev_alloc(&c->d_nk, 4));
    CK(dev_alloc(&c->d_nk2, 8));
    CKH(hipMemsetAsync(c->full, 0, sizeof(double) * 6 * (size_t)c->n_pad, c->stream));
    CKH(hipMemsetAsync(c->sub, 0, sizeof(double) * 6 * (size_t)c->s_pad, c->stream));
    CKH(hipMemsetAsync(c->dis, 0, sizeof(double) * 6 * (size_t)c->dis_stride, c->stream));
    CKH(hipMemsetAsync(c->d_total, 0, sizeof(int32_t), c->stream));
    stamp("streams, allocations, memsets");

    if (n > 0) {
        CK(dev_alloc(&t_xyz, 3 * n));
        CK(dev_alloc(&t_nrm, 3 * n));
        CKH(hipMemcpyAsync(t_xyz, xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        CKH(hipMemcpyAsync(t_nrm, nrm, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        stamp("upload (xyz, nrm)");
        CK(rhk_transpose_aos(c, t_xyz, t_nrm, n, nullptr, n, c->full, c->n_pad));
        CK(rhk_pack_records(c, t_xyz, t_nrm, n, c->rec));
        stamp("AoS -> SoA, records");
        if (rh_opt_on(nullptr, RH_OPT_AABB_HOST)) cloud_aabb(xyz, n, c->k_lo, &c->k_size, &c->k_mag);   // (A/B: 25 ms at 10M points, 120 ms at 50M)
        else CK(cloud_aabb_device(c, t_xyz, xyz, n));
        stamp("bounding box");
        CK(rhk_korder_build(c, t_xyz, t_nrm, c->k_lo, c->k_size, c->k_mag));
        stamp("Morton order of the cloud");
        if (s > 0) {
            // internal order of subset 1: 64 consecutive points are spatially compact, which is what the
            // culled score kernel's per-group boxes need (k-d leaves, below)
            h_idx = new (std::nothrow) int32_t[2 * (size_t)s];
            if (!h_idx) { rh_set_error("out of host memory"); return fail(RH_E_NOMEM); }
            ms_before_kd = ms_since(tc0);
            const auto tkd0 = std::chrono::steady_clock::now();
            // Where the order is made: on the device (kdorder.hip: a radix sort per level of the same balanced k-d tree; the
            // subset's coordinate / normal magnitudes come out of the same pass) unless RH_KD_HOST=1 or RH_SUB_ORDER=morton
            // ask for the host forms below (A/B; the counts do not depend on the order).
            const bool ord_morton = rh_opt_on(nullptr, RH_OPT_SUB_ORDER);
            const bool kd_device = !rh_opt_on(nullptr, RH_OPT_KD_HOST) && !ord_morton;
            if (kd_device) {
                for (int64_t j = 0; j < s; j++) h_idx[j] = (int32_t)(subset1[j] - 1);
                int32_t *d_idx_in = nullptr;
                CK(dev_alloc(&d_idx_in, s));
                hipError_t ec = hipMemcpyAsync(d_idx_in, h_idx, sizeof(int32_t) * (size_t)s, hipMemcpyHostToDevice, c->stream);
                rc = ec == hipSuccess ? rhk_kd_order(c, t_xyz, t_nrm, d_idx_in) : RH_E_NODEVICE;
                (void)hipFree(d_idx_in);
                if (ec != hipSuccess) rh_set_error("hipMemcpyAsync(subset indices): %s", hipGetErrorString(ec));
                if (rc != RH_OK) return fail(rc);
                ms_kd = ms_since(tkd0);
                CK(rhk_transpose_aos(c, t_xyz, t_nrm, n, c->sub_idx0, s, c->sub, c->s_pad));
                CK(rhk_group_bounds(c));
            } else {
            double lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 }, mag = 0;
            bool first = true;
            for (int64_t j = 0; j < s; j++) {
                const double *pp = xyz + 3 * (subset1[j] - 1);
                for (int k = 0; k < 3; k++) {
                    const double v = pp[k];
                    if (!(v == v) || v - v != 0) continue;   // NaN / inf do not shape the box
                    if (first || v < lo[k]) lo[k] = v;
                    if (first || v > hi[k]) hi[k] = v;
                    if (fabs(v) > mag) mag = fabs(v);
                }
                first = false;
            }
            c->coord_mag = mag;
            double nmag = 0;
            for (int64_t j = 0; j < s; j++) {
                const double *nn = nrm + 3 * (subset1[j] - 1);
                for (int k = 0; k < 3; k++) {
                    const double v = fabs(nn[k]);
                    if (v - v == 0 && v > nmag) nmag = v;   // (NaN / inf: score4_device.h, guards)
                }
            }
            c->nrm_mag = nmag;
            // Internal order = leaves of a balanced k-d tree, 64 points each (median split along the widest
            // axis of the node's box, the left child always a multiple of 64 so that only the LAST group
            // is partial).  Against Morton order the groups' boxes are ~30 % smaller in radius and
            // 16-26 % fewer (candidate, group) pairs survive the box tests (cfg3).  RH_SUB_ORDER=morton
            // keeps the Morton order (A/B).
            std::vector<int32_t> order((size_t)s);
            for (int64_t j = 0; j < s; j++) order[(size_t)j] = (int32_t)j;
            if (ord_morton) {
                std::vector<std::pair<uint64_t, int32_t>> keys((size_t)s);
                for (int64_t j = 0; j < s; j++) {
                    const double *pp = xyz + 3 * (subset1[j] - 1);
                    uint64_t code = 0;
                    for (int k = 0; k < 3; k++) {
                        const double ext = hi[k] - lo[k];
                        double t = ext > 0 ? (pp[k] - lo[k]) / ext : 0.0;
                        if (!(t >= 0)) t = 0;
                        if (t > 1) t = 1;
                        code |= spread21((uint64_t)(t * 2097151.0)) << k;
                    }
                    keys[(size_t)j] = std::make_pair(code, (int32_t)j);
                }
                std::sort(keys.begin(), keys.end());
                for (int64_t i = 0; i < s; i++) order[(size_t)i] = keys[(size_t)i].second;
            } else {
                auto key = [&](int32_t j, int ax) {   // NaN sorts last; the order only has to be total
                    const double v = xyz[3 * (subset1[j] - 1) + ax];
                    return v == v ? v : HUGE_VAL;
                };
                // one node: widest axis of the finite points' box, median split at a multiple of 64 (std::nth_element with
                // a TOTAL order, so the result does not depend on who runs it); returns the split position, 0 for a leaf
                auto split = [&](int64_t lo_i, int64_t hi_i) -> int64_t {
                    const int64_t cnt = hi_i - lo_i;
                    if (cnt <= 64) return 0;
                    double blo[3] = { HUGE_VAL, HUGE_VAL, HUGE_VAL }, bhi[3] = { -HUGE_VAL, -HUGE_VAL, -HUGE_VAL };
                    for (int64_t i = lo_i; i < hi_i; i++) {
                        const double *pp = xyz + 3 * (subset1[order[(size_t)i]] - 1);
                        for (int k = 0; k < 3; k++) {
                            const double v = pp[k];
                            if (!(v == v) || v - v != 0) continue;
                            if (v < blo[k]) blo[k] = v;
                            if (v > bhi[k]) bhi[k] = v;
                        }
                    }
                    int ax = 0;
                    double best = -1;
                    for (int k = 0; k < 3; k++) {
                        const double e = bhi[k] - blo[k];
                        if (e > best) { best = e; ax = k; }
                    }
                    const int64_t nl = ((cnt / 64 + 1) / 2) * 64;
                    std::nth_element(order.begin() + lo_i, order.begin() + lo_i + nl, order.begin() + hi_i,
                                     [&](int32_t a, int32_t b) {
                                         const double ka = key(a, ax), kb = key(b, ax);
                                         return ka < kb || (ka == kb && a < b);
                                     });
                    return lo_i + nl;
                };
                struct Node { int64_t lo, hi; };
                auto subtree = [&](int64_t lo_i, int64_t hi_i) {   // a whole subtree, depth first, on the calling thread
                    std::vector<Node> stack;
                    stack.push_back({ lo_i, hi_i });
                    while (!stack.empty()) {
                        const Node nd = stack.back();
                        stack.pop_back();
                        const int64_t mid = split(nd.lo, nd.hi);
                        if (mid == 0) continue;
                        stack.push_back({ mid, nd.hi });
                        stack.push_back({ nd.lo, mid });
                    }
                };
                // The subtrees are disjoint ranges of `order`: the top levels hand their right halves to new threads (up
                // to 2^KD_PAR_LEVELS of them, no more than the host has cores), so the O(s) passes of the upper levels run
                // side by side -- 112 ms single-threaded at s = 312 500, a fifth of that on 16 cores.
                constexpr int KD_PAR_LEVELS = 4;
                const unsigned hw = std::thread::hardware_concurrency();
                const int par_levels = (hw >= 16 ? KD_PAR_LEVELS : (hw >= 8 ? 3 : (hw >= 4 ? 2 : (hw >= 2 ? 1 : 0))));
                std::function<void(int64_t, int64_t, int)> build = [&](int64_t lo_i, int64_t hi_i, int depth) {
                    if (depth >= par_levels || hi_i - lo_i < 4096) { subtree(lo_i, hi_i); return; }
                    const int64_t mid = split(lo_i, hi_i);
                    if (mid == 0) return;
                    std::thread right;
                    try {
                        right = std::thread([&, mid, hi_i, depth]() { build(mid, hi_i, depth + 1); });
                    } catch (const std::system_error &) {   // no thread to be had: this one does both halves
                        subtree(mid, hi_i);
                    }
                    build(lo_i, mid, depth + 1);
                    if (right.joinable()) right.join();
                };
                build(0, s, 0);
            }
            int32_t *h_perm = h_idx + s;
            for (int64_t i = 0; i < s; i++) {
                h_perm[i] = order[(size_t)i];
                h_idx[i] = (int32_t)(subset1[h_perm[i]] - 1);
            }
            ms_kd = ms_since(tkd0);
            CKH(hipMemcpyAsync(c->sub_idx0, h_idx, sizeof(int32_t) * (size_t)s, hipMemcpyHostToDevice, c->stream));
            CKH(hipMemcpyAsync(c->sub_perm, h_perm, sizeof(int32_t) * (size_t)s, hipMemcpyHostToDevice, c->stream));
            CK(rhk_transpose_aos(c, t_xyz, t_nrm, n, c->sub_idx0, s, c->sub, c->s_pad));
            CK(rhk_group_bounds(c));
            }   // host forms of the order
        }
    }
    stamp("subset order, gather, boxes");
    CK(set_all_enabled(c));
    CKH(hipStreamSynchronize(c->stream));
    stamp("enabled bits");
#undef CK
#undef CKH
    (void)hipFree(t_xyz);
    (void)hipFree(t_nrm);
    delete[] h_idx;
    c->create_ms[0] = ms_since(tc0);
    c->create_ms[1] = ms_kd;
    c->create_ms[2] = ms_before_kd;
    c->create_ms[3] = c->create_ms[0] - ms_kd - ms_before_kd;
    *out = c;
    return RH_OK;
}

// wall time of the rh_cloud_create that made this cloud, in ms: total, the k-d leaf order of subset 1 (device: kdorder.hip;
// RH_KD_HOST=1: host threads) with the subset gather and the group boxes, everything before it (allocations, uploads,
// AoS -> SoA, bounding box, the Morton order of the cloud on the device), everything after it (enabled bits)
extern "C" int rh_cloud_create_ms(const rh_cloud *c, double *out4)
{
    if (!c || !out4) { rh_set_error("rh_cloud_create_ms: NULL argument"); return RH_E_INVALID; }
    for (int i = 0; i < 4; i++) out4[i] = c->create_ms[i];
    return RH_OK;
}

extern "C" int rh_cloud_destroy(rh_cloud *c)
{
    cloud_free(c);
    return RH_OK;
}

// RANSACCloud(vertices, normals, subsets; force_eltype = Float32) (src/octree.jl:102-109): xyz / nrm are Julia's
// Vector{SVector{3,Float32}} memory as is.  The cloud is built like a Float64 cloud from the exactly converted values
// (k-d leaf order, boxes, enabled bits, index plumbing) and gets float copies of its two point sets on top; scoring and
// refit then compute in binary32 (f32.hip).
extern "C" int rh_cloud_create_f32(const float *xyz, const float *nrm, int64_t n, const int64_t *subset1, int64_t s, int device,
                                   rh_cloud **out)
{
    if (!out) { rh_set_error("out is NULL"); return RH_E_INVALID; }
    *out = nullptr;
    if (n < 0 || (n > 0 && (!xyz || !nrm))) { rh_set_error("rh_cloud_create_f32: bad arguments"); return RH_E_INVALID; }
    std::vector<double> x64((size_t)(3 * n)), n64((size_t)(3 * n));
    for (int64_t i = 0; i < 3 * n; i++) { x64[(size_t)i] = (double)xyz[i]; n64[(size_t)i] = (double)nrm[i]; }
    rh_cloud *c = nullptr;
    RH_TRY(rh_cloud_create(x64.data(), n64.data(), n, subset1, s, device, &c));
    c->f32 = true;
    c->f32_groups = c->use_groups;   // culled kernel with a binary32 exact test where a Float64 cloud would use the culled kernel
    c->use_groups = false;           // (the Float64 dispatch below never runs for this cloud)
    int rc = dev_alloc(&c->full32, 6 * std::max<int64_t>(c->n_pad, 1));
    if (rc == RH_OK) rc = dev_alloc(&c->sub32, 6 * std::max<int64_t>(c->s_pad, 1));
    if (rc == RH_OK) rc = rhk_f32_build(c);
    if (rc == RH_OK) rc = rhk_korder_build_f32(c);
    if (rc == RH_OK && hipStreamSynchronize(c->stream) != hipSuccess) { rh_set_error("rh_cloud_create_f32: device error"); rc = RH_E_NODEVICE; }
    if (rc != RH_OK) { cloud_free(c); return rc; }
    *out = c;
    return RH_OK;
}

extern "C" int rh_cloud_info(const rh_cloud *c, int64_t *n, int64_t *s, int *device)
{
    if (!c) { rh_set_error("cloud is NULL"); return RH_E_INVALID; }
    if (n) *n = c->n;
    if (s) *s = c->s;
    if (device) *device = c->device;
    return RH_OK;
}

static int enter_nojoin(rh_cloud *c)
{
    if (!c) { rh_set_error("cloud is NULL"); return RH_E_INVALID; }
    RH_HIP(hipSetDevice(c->device));
    return RH_OK;
}

// every entry point but the pipelined rh_score_batch_dev: the cloud's stream first waits for what the other batch slots
// still have in flight ("batches_in_flight" > 1), so everything else sees one stream's order
int rh_join_batches(rh_cloud *c)
{
    for (int i = 0; i < RH_MAX_IN_FLIGHT - 1; i++) {
        if (c->alt_dirty[i]) RH_HIP(hipStreamWaitEvent(c->stream, c->alt[i].done, 0));
        c->alt_dirty[i] = false;
        c->alt_started[i] = false;
    }
    for (int i = 0; i < RH_MAX_IN_FLIGHT; i++) { c->slot_counts[i] = nullptr; c->slot_masks[i] = nullptr; }
    c->pipe_k = 0;
    return RH_OK;
}

static int enter(rh_cloud *c)
{
    RH_TRY(enter_nojoin(c));
    return rh_join_batches(c);
}

int rh_cloud_join(rh_cloud *c) { return enter(c); }

extern "C" int rh_cloud_set_enabled(rh_cloud *c, const uint64_t *chunks, int64_t nchunks)
{
    RH_TRY(enter(c));
    if (nchunks != c->nwords || (nchunks > 0 && !chunks)) {
        rh_set_error("rh_cloud_set_enabled: nchunks=%lld, expected %lld", (long long)nchunks, (long long)c->nwords);
        return RH_E_INVALID;
    }
    if (c->nwords > 0) {
        c->k_men_valid = false;
        RH_HIP(hipMemcpyAsync(c->enabled, chunks, sizeof(uint64_t) * (size_t)nchunks, hipMemcpyHostToDevice, c->stream));
        if (c->n % 64) {   // BitVector keeps the unused tail bits zero; enforce it
            const uint64_t last = chunks[nchunks - 1] & ((~0ULL) >> (64 - c->n % 64));
            RH_HIP(hipMemcpyAsync(c->enabled + (nchunks - 1), &last, sizeof last, hipMemcpyHostToDevice, c->stream));
        }
        RH_HIP(hipStreamSynchronize(c->stream));
    }
    RH_TRY(rhk_rebuild_sub_enabled(c, true));
    RH_HIP(hipStreamSynchronize(c->stream));
    c->select_valid = false;
    c->en_sums_valid = false;
    return RH_OK;
}

extern "C" int rh_cloud_get_enabled(rh_cloud *c, uint64_t *chunks, int64_t nchunks)
{
    RH_TRY(enter(c));
    if (nchunks != c->nwords || (nchunks > 0 && !chunks)) {
        rh_set_error("rh_cloud_get_enabled: nchunks=%lld, expected %lld", (long long)nchunks, (long long)c->nwords);
        return RH_E_INVALID;
    }
    if (c->nwords > 0) {
        RH_HIP(hipMemcpyAsync(chunks, c->enabled, sizeof(uint64_t) * (size_t)nchunks, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipStreamSynchronize(c->stream));
    }
    return RH_OK;
}

extern "C" int rh_cloud_enable_all(rh_cloud *c)
{
    RH_TRY(enter(c));
    RH_TRY(set_all_enabled(c));
    RH_HIP(hipStreamSynchronize(c->stream));
    return RH_OK;
}

extern "C" int rh_cloud_count_enabled(rh_cloud *c, int64_t *out)
{
    RH_TRY(enter(c));
    if (!out) { rh_set_error("out is NULL"); return RH_E_INVALID; }
    return rhk_count_enabled(c, out);
}

// ---------------------------------------------------------------- scoring ----
static inline const uint64_t *enabled_for_kind(const rh_cloud *c, int kind, const rh_params *p)
{
    // sphere.jl:121,131: the sphere scorer builds `ens` and never applies it
    if (kind == RH_SPHERE && !p->sphere_uses_enabled) return nullptr;
    return c->sub_enabled;
}

// all four kind bins (bin k at prep/orig + off[k], its size in d_nk[k]) against subset 1.
// Culled path (subsets of RH_G2_MIN_POINTS points and more): ONE launch of score4.hip's kernel over all kinds, reading the
// bins' classifier / culling records d_cls / d_box (64 B per slot at the same offsets / fields bstride apart); masks leave
// it as entry lists (c->masks4).  ms_kind (the bench's per-kind leg): one launch per kind instead, the other kinds' bin
// sizes read as zero, an event before each.  Small subsets: the brute-force kernel, one launch per kind, masks in internal
// order.  Float32 clouds: the same two paths with the exact tests in binary32.
static int score_bins_subset(rh_cloud *c, const rh_params *p, const rh_prep *d_prep, const int32_t *d_orig,
                             const int64_t off[4], const int32_t *d_nk, const int32_t nk_bound[4], int32_t total_bound,
                             int32_t *d_counts, uint64_t *d_masks_int, float *ms_kind, const void *d_cls = nullptr,
                             const float *d_box = nullptr, int64_t bstride = 0)
{
    const uint64_t *en[4];
    for (int k = 0; k < 4; k++) en[k] = enabled_for_kind(c, k, p);
    if (rh_score_v4_enabled(c)) {
        if (d_cls == nullptr || d_box == nullptr) { rh_set_error("internal: culled scoring without the bins' classifier records"); return RH_E_INTERNAL; }
        const rh_prep *pr[4];
        const int32_t *og[4], *nk[4];
        const void *cl[4];
        const float *bx[4];
        for (int k = 0; k < 4; k++) {
            pr[k] = d_prep + off[k];
            og[k] = d_orig + off[k];
            nk[k] = d_nk + k;
            cl[k] = (const char *)d_cls + (size_t)off[k] * 64;
            bx[k] = d_box + off[k];
        }
        if (!ms_kind)
            return rhk_score_all_groups(c, en, pr, og, nk, total_bound, p->eps, p->cos_alpha, d_counts, d_masks_int, cl, bx, bstride);
        if (c->d_zero == nullptr) {
            RH_HIP(hipMalloc((void **)&c->d_zero, 64));
            RH_HIP(hipMemsetAsync(c->d_zero, 0, 64, c->stream));
        }
        for (int k = 0; k < 4; k++) {
            RH_HIP(hipEventRecord(c->evk[k], c->stream));
            if (nk_bound[k] == 0) continue;
            const int32_t *nk1[4];
            for (int q = 0; q < 4; q++) nk1[q] = q == k ? nk[q] : c->d_zero;
            RH_TRY(rhk_score_all_groups(c, en, pr, og, nk1, nk_bound[k], p->eps, p->cos_alpha, d_counts, d_masks_int, cl, bx, bstride));
        }
        return RH_OK;
    }
    if (c->f32) {   // Float32 cloud, small subset: float records from the batch's shapes, brute-force float kernel (f32.hip)
        if (c->f32_shapes == nullptr) { rh_set_error("internal: Float32 scoring without the batch's shapes"); return RH_E_INTERNAL; }
        for (int k = 0; k < 4; k++)
            if (ms_kind) RH_HIP(hipEventRecord(c->evk[k], c->stream));
        return rhk_score_all_f32(c, c->f32_shapes, c->f32_via_orig, en, d_orig, off, d_nk, nk_bound, p->eps, p->cos_alpha,
                                 d_counts, d_masks_int);
    }
    for (int k = 0; k < 4; k++) {
        if (ms_kind) RH_HIP(hipEventRecord(c->evk[k], c->stream));
        if (nk_bound[k] == 0) continue;
        RH_TRY(rhk_score_kind(c, k, c->sub, c->s_pad, c->s, en[k], d_prep + off[k], d_orig + off[k], d_nk + k, nk_bound[k],
                              p->eps[k], p->cos_alpha[k], d_counts, d_masks_int, c->swords));
    }
    return RH_OK;
}

extern "C" int rh_score_batch(rh_cloud *c, const rh_shape *shapes, int32_t b, const rh_params *p, int32_t *counts_out,
                              uint64_t *masks_out)
{
    RH_TRY(enter(c));
    RH_TRY(rh_validate_params(p));
    if (b < 0 || (b > 0 && (!shapes || !counts_out))) { rh_set_error("rh_score_batch: bad arguments"); return RH_E_INVALID; }
    if (b == 0) return RH_OK;
    int32_t nk[4] = { 0, 0, 0, 0 };
    for (int32_t i = 0; i < b; i++) {
        if (shapes[i].kind < 0 || shapes[i].kind > 3) {
            rh_set_error("candidate %d has unknown kind %d", i, shapes[i].kind);
            return RH_E_INVALID;
        }
        nk[shapes[i].kind]++;
    }
    RH_TRY(rh_ensure_batch(c, b));
    // Pinned staging: [records | original positions | bin sizes | counts] -- every transfer of the call is
    // asynchronous and the call waits once.  Batches of the size scorecandidates! sees per iteration (a few
    // candidates; here: up to 32) go as ONE small transfer into a device block of the same layout, carrying the
    // prepared records themselves (rh_prep_host is the host twin of the device's prep_one) and the zeroed counts:
    // upload, score launch, read-back.  (Measured: 30 -> 23 us at b = 1, 38 -> 31 us at b = 15; from ~6 KB on the
    // single larger transfer is slower than three small ones, so larger batches upload shapes, positions and
    // bin sizes separately and a kernel prepares the records and zeroes the counts.)
    const bool staged = b <= 32 && !c->f32;   // (the staged form carries prepared Float64 records only)
    const size_t rec_bytes = sizeof(rh_prep) >= sizeof(rh_shape) ? sizeof(rh_prep) : sizeof(rh_shape);
    const size_t o_orig = (size_t)b * rec_bytes, o_nk = (o_orig + (size_t)b * sizeof(int32_t) + 63) / 64 * 64;
    // (staged, culled path: the classifier records -- host twin of the prep kernels' cls_make -- ride along)
    const bool staged_cls = staged && rh_score_v4_enabled(c);
    const size_t o_counts = o_nk + 64, o_cls = (o_counts + (size_t)b * sizeof(int32_t) + 63) / 64 * 64;
    const size_t o_box = o_cls + (size_t)b * 64;   // culling records: RH_BOX_FIELDS arrays of b floats
    const size_t stage_bytes = staged_cls ? o_box + (size_t)rh4::RH_BOX_FIELDS * b * sizeof(float) : o_counts + (size_t)b * sizeof(int32_t);
    RH_TRY(rh_ensure_pin(c, (int64_t)stage_bytes));
    if (staged && c->d_stage == nullptr) RH_HIP(hipMalloc((void **)&c->d_stage, 32 * (rec_bytes + 8 + 64 + 4 * rh4::RH_BOX_FIELDS) + 512));
    char *hp = (char *)c->h_pin, *dp = (char *)c->d_stage;
    rh_shape *h_sorted = (rh_shape *)hp;
    rh_prep *h_prep = (rh_prep *)hp;
    int32_t *h_orig = (int32_t *)(hp + o_orig);
    int32_t *h_nk = (int32_t *)(hp + o_nk);
    int32_t *h_counts = (int32_t *)(hp + o_counts);
    int32_t off[4], fill[4];
    off[0] = 0;
    for (int k = 1; k < 4; k++) off[k] = off[k - 1] + nk[k - 1];
    for (int k = 0; k < 4; k++) { fill[k] = off[k]; h_nk[k] = nk[k]; }
    // counting sort by kind, walking the batch in the spread order (kernels.hip: neighbours in the batch, often
    // hypotheses of the same primitive, go to different 64-candidate chunks)
    const int64_t spread = rh_opt_on(c, RH_OPT_NO_SPREAD) ? 1 : rh_spread_multiplier(b);
    for (int32_t t = 0; t < b; t++) {
        const int32_t i = (int32_t)(((int64_t)t * spread) % b);
        const int k = shapes[i].kind;
        if (staged) {
            rh_prep_host(shapes[i], &h_prep[fill[k]]);
            if (staged_cls)
                rh4::cls_make(h_prep[fill[k]], k, p->eps[k], p->cos_alpha[k], c->coord_mag, c->nrm_mag, ((rh4::rh_cls *)(hp + o_cls))[fill[k]],
                              (float *)(hp + o_box) + fill[k], b);
        } else h_sorted[fill[k]] = shapes[i];
        h_orig[fill[k]] = i;
        fill[k]++;
    }
    const rh_prep *d_prep_use = c->d_prep;
    const void *d_cls_use = nullptr;
    const float *d_box_use = nullptr;
    int64_t bstride_use = 0;
    const int32_t *d_orig_use = c->d_orig, *d_nk_use = c->d_nk;
    int32_t *d_counts_use = c->d_counts;
    if (staged) {
        memset(h_counts, 0, sizeof(int32_t) * (size_t)b);
        RH_HIP(hipMemcpyAsync(dp, hp, stage_bytes, hipMemcpyHostToDevice, c->stream));
        d_prep_use = (const rh_prep *)dp;
        d_orig_use = (const int32_t *)(dp + o_orig);
        d_nk_use = (const int32_t *)(dp + o_nk);
        d_counts_use = (int32_t *)(dp + o_counts);
        if (staged_cls) { d_cls_use = dp + o_cls; d_box_use = (const float *)(dp + o_box); bstride_use = b; }
    } else {
        RH_HIP(hipMemcpyAsync(c->d_shapes, h_sorted, sizeof(rh_shape) * (size_t)b, hipMemcpyHostToDevice, c->stream));
        RH_HIP(hipMemcpyAsync(c->d_orig, h_orig, sizeof(int32_t) * (size_t)b, hipMemcpyHostToDevice, c->stream));
        RH_HIP(hipMemcpyAsync(c->d_nk, h_nk, 4 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        RH_TRY(rhk_prep_sorted(c, c->d_shapes, b, c->d_prep, c->d_counts, p->eps, p->cos_alpha));   // zeroes d_counts as well
        if (c->qpre_v4) { d_cls_use = c->d_qpre; d_box_use = c->d_box; bstride_use = 4 * c->batch_cap; }
    }
    uint64_t *d_masks = nullptr, *d_masks_int = nullptr;
    c->masks4 = false;
    if (masks_out && c->swords > 0) {
        RH_TRY(rh_ensure_masks(c, (int64_t)b * c->swords));
        d_masks = c->d_masks;
        if (d_cls_use != nullptr) {   // the culled kernel leaves entry lists: nothing to zero
            RH_TRY(ensure_masks4(c, b));
            c->masks4 = true;
        } else {                      // the brute-force kernel ORs into dense rows in internal order
            RH_TRY(ensure_masks_int(c, (int64_t)b * c->swords));
            RH_HIP(hipMemsetAsync(c->d_masks_int, 0, sizeof(uint64_t) * (size_t)b * (size_t)c->swords, c->stream));
        }
        d_masks_int = c->d_masks_int;
    }
    const int64_t off64[4] = { off[0], off[1], off[2], off[3] };
    c->f32_shapes = c->d_shapes;   // sorted like the bins
    c->f32_via_orig = 0;
    RH_TRY(score_bins_subset(c, p, d_prep_use, d_orig_use, off64, d_nk_use, nk, b, d_counts_use, d_masks_int, nullptr, d_cls_use, d_box_use, bstride_use));
    if (d_masks_int && c->masks4) RH_TRY(rhk_unpermute_masks4(c, d_masks_int, c->d_occ, c->mstride4, b, d_masks));
    else if (d_masks_int) RH_TRY(rhk_unpermute_masks(c, d_masks_int, b, d_masks));
    c->masks4 = false;
    RH_HIP(hipMemcpyAsync(h_counts, d_counts_use, sizeof(int32_t) * (size_t)b, hipMemcpyDeviceToHost, c->stream));
    if (d_masks)
        RH_HIP(hipMemcpyAsync(masks_out, d_masks, sizeof(uint64_t) * (size_t)b * (size_t)c->swords,
                              hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    memcpy(counts_out, h_counts, sizeof(int32_t) * (size_t)b);
    return RH_OK;
}

// A stream that really runs beside the given ones.  The runtime deals its streams to a few hardware queues (the least used
// one at creation), and two streams on one queue run one after the other -- which queue a new stream lands on depends on every
// stream the process has made before (other clouds, torch, copy streams).  So: candidates are created until one of them gets
// a kernel through while all the given streams are held busy by a spinning wave (~0.3 ms, once per slot and cloud).
namespace {
__global__ void spin_kernel(unsigned long long ticks)   // one wave, ~ticks of the 100 MHz clock
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void noop_kernel() {}
}  // namespace

static int pick_concurrent_stream(const hipStream_t *busy, int nbusy, hipStream_t *out)
{
    hipStream_t tried[8];
    int ntried = 0, chosen = -1;
    hipEvent_t eb[RH_MAX_IN_FLIGHT], ec;
    for (int i = 0; i < nbusy; i++) RH_HIP(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming));
    RH_HIP(hipEventCreateWithFlags(&ec, hipEventDisableTiming));
    int rc = RH_OK;
    for (; ntried < 8 && chosen < 0 && rc == RH_OK; ntried++) {
        if (hipStreamCreateWithFlags(&tried[ntried], hipStreamNonBlocking) != hipSuccess) { rc = RH_E_NODEVICE; rh_set_error("hipStreamCreateWithFlags failed"); break; }
        for (int i = 0; i < nbusy; i++) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, busy[i], 30000ULL);   // 0.3 ms
            (void)hipEventRecord(eb[i], busy[i]);
        }
        hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, tried[ntried]);
        (void)hipEventRecord(ec, tried[ntried]);
        (void)hipEventSynchronize(ec);
        bool beside = true;   // the candidate's kernel is through: are all the busy streams still spinning?
        for (int i = 0; i < nbusy; i++) beside = beside && hipEventQuery(eb[i]) == hipErrorNotReady;
        for (int i = 0; i < nbusy; i++) (void)hipEventSynchronize(eb[i]);
        if (beside) chosen = ntried;
    }
    if (rc == RH_OK && chosen < 0) chosen = ntried - 1;   // (none found: the last one -- correct, only not concurrent)
    for (int i = 0; i < ntried; i++)
        if (i != chosen) (void)hipStreamDestroy(tried[i]);
    for (int i = 0; i < nbusy; i++) (void)hipEventDestroy(eb[i]);
    (void)hipEventDestroy(ec);
    (void)hipGetLastError();
    if (rc != RH_OK) return rc;
    *out = tried[chosen];
    return RH_OK;
}

static void swap_batch_slot(rh_cloud *c, rh_batch_slot &s)
{
    std::swap(c->stream, s.stream);
    std::swap(c->batch_cap, s.batch_cap);
    std::swap(c->d_shapes, s.d_shapes); std::swap(c->d_prep, s.d_prep); std::swap(c->d_orig, s.d_orig); std::swap(c->d_counts, s.d_counts);
    std::swap(c->d_nk2, s.d_nk2); std::swap(c->d_qpre, s.d_qpre); std::swap(c->d_prep32, s.d_prep32); std::swap(c->d_box, s.d_box);
    std::swap(c->nk2_flip, s.nk2_flip); std::swap(c->nk2_ready, s.nk2_ready); std::swap(c->qpre_v4, s.qpre_v4);
    std::swap(c->d_masks_int, s.d_masks_int); std::swap(c->masks_int_cap, s.masks_int_cap); std::swap(c->d_occ, s.d_occ);
    std::swap(c->occ_cap, s.occ_cap); std::swap(c->mstride4, s.mstride4);
    std::swap(c->d_stlist, s.d_stlist); std::swap(c->d_stcount, s.d_stcount); std::swap(c->stlist_cap, s.stlist_cap); std::swap(c->stlist_nst, s.stlist_nst);
}

static int score_batch_dev_impl(rh_cloud *c, const rh_shape *d_shapes, int32_t b, const rh_params *p,
                                int32_t *d_counts, uint64_t *d_masks, float *ms_kind, bool joined = true)
{
    RH_TRY(joined ? enter(c) : enter_nojoin(c));
    RH_TRY(rh_validate_params(p));
    if (b < 0 || (b > 0 && (!d_shapes || !d_counts))) { rh_set_error("rh_score_batch_dev: bad arguments"); return RH_E_INVALID; }
    const bool product_only = ms_kind && ms_kind[0] < 0.f;   // (profiling passes: no per-kind launches beside the product's)
    if (ms_kind) for (int k = 0; k < 5; k++) ms_kind[k] = 0.f;
    if (b == 0) return RH_OK;
    RH_TRY(rh_ensure_batch(c, b));
    // bin sizes: two halves of d_nk2 used alternately; the prep kernel zeroes the counts and the other half,
    // so a step is two launches and no memset
    if (!c->nk2_ready) {
        RH_HIP(hipMemsetAsync(c->d_nk2, 0, 8 * sizeof(int32_t), c->stream));
        c->nk2_ready = true;
    }
    int32_t *nk_cur = c->d_nk2 + 4 * c->nk2_flip, *nk_next = c->d_nk2 + 4 * (1 - c->nk2_flip);
    c->nk2_flip = 1 - c->nk2_flip;
    RH_TRY(rhk_prep_binned(c, d_shapes, b, c->d_prep, c->d_orig, nk_cur, c->batch_cap, d_counts, nk_next, 1, p->eps, p->cos_alpha));
    uint64_t *d_masks_int = nullptr;
    c->masks4 = false;
    if (d_masks && c->swords > 0) {
        if (c->qpre_v4) {   // the culled kernel leaves entry lists: nothing to zero
            RH_TRY(ensure_masks4(c, b));
            c->masks4 = true;
        } else {
            RH_TRY(ensure_masks_int(c, (int64_t)b * c->swords));
            RH_HIP(hipMemsetAsync(c->d_masks_int, 0, sizeof(uint64_t) * (size_t)b * (size_t)c->swords, c->stream));
        }
        d_masks_int = c->d_masks_int;
    }
    const int64_t off[4] = { 0, c->batch_cap, 2 * (int64_t)c->batch_cap, 3 * (int64_t)c->batch_cap };
    const int32_t bound[4] = { b, b, b, b };
    const void *d_cls = c->qpre_v4 ? c->d_qpre : nullptr;   // (made by rhk_prep_binned above)
    c->f32_shapes = d_shapes;      // the caller's order: the float records go through d_orig
    c->f32_via_orig = 1;
    if (ms_kind) {   // the product launch (all kinds in one kernel) first, then the per-kind launches
        ms_kind[4] = 0.f;
        if (c->ev_cull == nullptr) RH_HIP(hipEventCreate(&c->ev_cull));
        RH_HIP(hipEventRecord(c->evk[0], c->stream));
        c->time_cull = true;
        c->last_s4[1] = 0;
        const int rc_s = score_bins_subset(c, p, c->d_prep, c->d_orig, off, nk_cur, bound, b, d_counts, d_masks_int, nullptr, d_cls, c->d_box, 4 * c->batch_cap);
        c->time_cull = false;
        RH_TRY(rc_s);
        RH_HIP(hipEventRecord(c->evk[1], c->stream));
        RH_HIP(hipEventSynchronize(c->evk[1]));
        c->last_cull_ms = 0.f;
        if (c->last_s4[1] != 0) {   // super-tile lists: the list launch and the score launch apart (ms_kind[4] = the score launch alone)
            RH_HIP(hipEventElapsedTime(&c->last_cull_ms, c->evk[0], c->ev_cull));
            RH_HIP(hipEventElapsedTime(&ms_kind[4], c->ev_cull, c->evk[1]));
        } else {
            RH_HIP(hipEventElapsedTime(&ms_kind[4], c->evk[0], c->evk[1]));
        }
        if (!product_only) {
            RH_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)b, c->stream));
            if (d_masks_int && c->masks4) RH_HIP(hipMemsetAsync(c->d_occ, 0, sizeof(int32_t) * (size_t)b, c->stream));   // (the lists' cursors)
            else if (d_masks_int) RH_HIP(hipMemsetAsync(d_masks_int, 0, sizeof(uint64_t) * (size_t)b * (size_t)c->swords, c->stream));
        }
    }
    if (!product_only)
        RH_TRY(score_bins_subset(c, p, c->d_prep, c->d_orig, off, nk_cur, bound, b, d_counts, d_masks_int, ms_kind, d_cls, c->d_box, 4 * c->batch_cap));
    if (d_masks_int && c->masks4) RH_TRY(rhk_unpermute_masks4(c, d_masks_int, c->d_occ, c->mstride4, b, d_masks));
    else if (d_masks_int) RH_TRY(rhk_unpermute_masks(c, d_masks_int, b, d_masks));
    c->masks4 = false;
    if (ms_kind && !product_only) {
        RH_HIP(hipEventRecord(c->evk[4], c->stream));
        RH_HIP(hipEventSynchronize(c->evk[4]));
        for (int k = 0; k < 4; k++) RH_HIP(hipEventElapsedTime(&ms_kind[k], c->evk[k], c->evk[k + 1]));
    }
    return RH_OK;
}

extern "C" int rh_score_batch_dev(rh_cloud *c, const rh_shape *d_shapes, int32_t b, const rh_params *p,
                                  int32_t *d_counts, uint64_t *d_masks)
{
    // "batches_in_flight" = F > 1 (rh_set_option): batches take turns on the cloud's stream and F - 1 more streams with
    // workspaces of their own, so batch k + 1's prepare and score launches fill the chip while batch k's launch drains (the
    // heaviest tile's block sets a launch's length: cfg2 0.042 -> 0.027 ms a batch, cfg3 0.085 -> 0.071 with F = 2).  The
    // caller gives call k of a run buffer k mod F of F count (and mask) buffers (the stream that wrote a buffer last writes it
    // next; with masks the un-permutation of one batch, HBM-bound, runs under the next batch's score launch, issue-bound); any other call on the cloud (and rh_cloud_sync /
    // rh_timer_stop) first makes the cloud's stream wait for the others.
    const int in_flight = c != nullptr && b > 0 && c->stream == c->own_stream ? rh_opt_int(c, RH_OPT_BATCHES_IN_FLIGHT, 1) : 1;
    if (in_flight > 1) {
        RH_TRY(enter_nojoin(c));
        int slot = (int)(c->pipe_k % (uint32_t)in_flight);
        // a caller whose buffers are not F apart (the same count buffer call after call, say): never two batches in flight on
        // one buffer -- the streams are joined first and the batch runs alone (correct, only not overlapped)
        for (int t = 0; t < RH_MAX_IN_FLIGHT; t++)
            if (t != slot && ((c->slot_counts[t] != nullptr && c->slot_counts[t] == d_counts) || (d_masks != nullptr && c->slot_masks[t] == d_masks))) {
                RH_TRY(rh_join_batches(c));
                slot = 0;
                break;
            }
        c->pipe_k++;
        c->slot_counts[slot] = d_counts;
        c->slot_masks[slot] = d_masks;
        if (slot == 0) return score_batch_dev_impl(c, d_shapes, b, p, d_counts, d_masks, nullptr, false);
        rh_batch_slot &s = c->alt[slot - 1];
        if (s.stream == nullptr) {
            hipStream_t busy[RH_MAX_IN_FLIGHT];
            int nbusy = 0;
            busy[nbusy++] = c->stream;
            for (int q = 0; q < RH_MAX_IN_FLIGHT - 1; q++) if (q != slot - 1 && c->alt[q].stream != nullptr) busy[nbusy++] = c->alt[q].stream;
            RH_TRY(pick_concurrent_stream(busy, nbusy, &s.stream));
            RH_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
            RH_HIP(hipEventCreateWithFlags(&s.start, hipEventDisableTiming));
            RH_HIP(hipMalloc((void **)&s.d_nk2, 8 * sizeof(int32_t)));
        }
        if (!c->alt_started[slot - 1]) {   // once per pipelined stretch: the stream starts behind what the cloud's stream holds so far
            RH_HIP(hipEventRecord(s.start, c->stream));
            RH_HIP(hipStreamWaitEvent(s.stream, s.start, 0));
            c->alt_started[slot - 1] = true;
        }
        swap_batch_slot(c, s);
        const int rc = score_batch_dev_impl(c, d_shapes, b, p, d_counts, d_masks, nullptr, false);
        swap_batch_slot(c, s);
        // (also behind a failed call: whatever it did enqueue on the slot's stream is waited for at the next join)
        RH_HIP(hipEventRecord(s.done, s.stream));
        c->alt_dirty[slot - 1] = true;
        return rc;
    }
    return score_batch_dev_impl(c, d_shapes, b, p, d_counts, d_masks, nullptr);
}

extern "C" int rh_last_list_launch_ms(rh_cloud *c, float *ms_out)
{
    if (!c || !ms_out) { rh_set_error("rh_last_list_launch_ms: bad arguments"); return RH_E_INVALID; }
    *ms_out = c->last_cull_ms;
    return RH_OK;
}

extern "C" int rh_score_launch_info(rh_cloud *c, int32_t *out4)
{
    if (!c || !out4) { rh_set_error("rh_score_launch_info: bad arguments"); return RH_E_INVALID; }
    for (int i = 0; i < 4; i++) out4[i] = c->last_s4[i];
    return RH_OK;
}

extern "C" int rh_score_batch_dev_timed(rh_cloud *c, const rh_shape *d_shapes, int32_t b, const rh_params *p,
                                        int32_t *d_counts, uint64_t *d_masks, float *ms_kind_out)
{
    if (!ms_kind_out) { rh_set_error("rh_score_batch_dev_timed: ms_kind_out is NULL"); return RH_E_INVALID; }
    return score_batch_dev_impl(c, d_shapes, b, p, d_counts, d_masks, ms_kind_out);
}

// ------------------------------------------------------------------ refit ----
extern "C" int rh_refit(rh_cloud *c, const rh_shape *shape, const rh_params *p, int64_t *idx_out, int64_t cap,
                        int64_t *n_out)
{
    RH_TRY(enter(c));
    RH_TRY(rh_validate_params(p));
    if (!shape || !n_out || cap < 0 || (cap > 0 && !idx_out)) { rh_set_error("rh_refit: bad arguments"); return RH_E_INVALID; }
    if (shape->kind < 0 || shape->kind > 3) { rh_set_error("unknown shape kind %d", shape->kind); return RH_E_INVALID; }
    *n_out = 0;
    rh_prep P;
    rh_prep_host(*shape, &P);
    c->select_valid = false;   // block_sums / d_total are shared with the select directory
    RH_HIP(hipEventRecord(c->evk[0], c->stream));
    if (c->f32) RH_TRY(rhk_refit_mask_f32(c, *shape, p->eps[shape->kind], p->cos_alpha[shape->kind]));
    else RH_TRY(rhk_refit_mask(c, P, shape->kind, p->eps[shape->kind], p->cos_alpha[shape->kind]));
    RH_HIP(hipEventRecord(c->evk[1], c->stream));
    RH_TRY(rhk_compact_mask(c, c->refit_mask, c->nwords, c->idx_out, c->n, c->d_total));
    RH_HIP(hipEventRecord(c->evk[2], c->stream));
    int32_t total = 0;
    RH_HIP(hipMemcpyAsync(&total, c->d_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    *n_out = total;
    if (total > cap) {
        rh_set_error("rh_refit: %d inliers, capacity %lld", total, (long long)cap);
        return RH_E_CAPACITY;
    }
    if (total > 0) {
        RH_HIP(hipMemcpyAsync(idx_out, c->idx_out, sizeof(int64_t) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipStreamSynchronize(c->stream));
    }
    return RH_OK;
}

extern "C" int rh_last_refit_ms(rh_cloud *c, float *ms_scan, float *ms_compact)
{
    RH_TRY(enter(c));
    RH_HIP(hipEventSynchronize(c->evk[2]));
    if (ms_scan) RH_HIP(hipEventElapsedTime(ms_scan, c->evk[0], c->evk[1]));
    if (ms_compact) RH_HIP(hipEventElapsedTime(ms_compact, c->evk[1], c->evk[2]));
    return RH_OK;
}

extern "C" int rh_invalidate(rh_cloud *c, const int64_t *idx, int64_t n)
{
    RH_TRY(enter(c));
    if (n < 0 || (n > 0 && !idx)) { rh_set_error("rh_invalidate: bad arguments"); return RH_E_INVALID; }
    if (n == 0) return RH_OK;
    for (int64_t k = 0; k < n; k++)   // before anything else: an empty cloud has no valid index at all
        if (idx[k] < 1 || idx[k] > c->n) {
            rh_set_error("rh_invalidate: index %lld outside 1..%lld", (long long)idx[k], (long long)c->n);
            return RH_E_INVALID;   // the reference would throw a BoundsError
        }
    if (n > c->n) {   // idx_out is the staging buffer; longer lists (duplicates) go in pieces (c->n >= 1 here)
        for (int64_t o = 0; o < n; o += c->n) RH_TRY(rh_invalidate(c, idx + o, std::min<int64_t>(c->n, n - o)));
        return RH_OK;
    }
    RH_HIP(hipMemcpyAsync(c->idx_out, idx, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    RH_TRY(rhk_invalidate_idx(c, c->idx_out, n));
    RH_TRY(rhk_rebuild_sub_enabled(c, false));
    RH_HIP(hipStreamSynchronize(c->stream));
    c->select_valid = false;
    return RH_OK;
}

extern "C" int rh_select_enabled(rh_cloud *c, const int64_t *ranks, int32_t k, int64_t *idx_out)
{
    RH_TRY(enter(c));
    if (k < 0 || (k > 0 && (!ranks || !idx_out))) { rh_set_error("rh_select_enabled: bad arguments"); return RH_E_INVALID; }
    if (k == 0) return RH_OK;
    if (c->nwords == 0) { for (int i = 0; i < k; i++) idx_out[i] = 0; return RH_OK; }
    if (k > c->ranks_cap) {
        RH_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_ranks);
        c->d_ranks = nullptr;
        c->ranks_cap = 0;
        RH_TRY(dev_alloc(&c->d_ranks, 2 * (int64_t)k));
        c->ranks_cap = k;
    }
    if (!c->select_valid) RH_TRY(rhk_build_select(c));
    if (k <= 1024) {
        // the per-sample call of the reference's loop (a couple of ranks): the kernel reads the ranks from the pinned
        // block and writes the indices back into it -- one launch and one wait, no transfer operation
        RH_TRY(rh_ensure_pin(c, 2 * (int64_t)sizeof(int64_t) * k));
        int64_t *h_ranks = (int64_t *)c->h_pin, *h_out = h_ranks + k;
        memcpy(h_ranks, ranks, sizeof(int64_t) * (size_t)k);
        RH_TRY(rhk_select(c, h_ranks, k, h_out));
        RH_HIP(hipStreamSynchronize(c->stream));
        memcpy(idx_out, h_out, sizeof(int64_t) * (size_t)k);
        return RH_OK;
    }
    RH_HIP(hipMemcpyAsync(c->d_ranks, ranks, sizeof(int64_t) * (size_t)k, hipMemcpyHostToDevice, c->stream));
    RH_TRY(rhk_select(c, c->d_ranks, k, c->d_ranks + c->ranks_cap));
    RH_HIP(hipMemcpyAsync(idx_out, c->d_ranks + c->ranks_cap, sizeof(int64_t) * (size_t)k, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    return RH_OK;
}

// samplepointcloud4! (fitting.jl:383-430) k times in a row on the caller's random stream: see sample_sets_seq_kernel.
extern "C" int rh_sample_sets(rh_cloud *c, int32_t drawN, rh_rng *rng, int32_t k, int64_t *idx_out, int32_t *ok_out, int32_t *level_out)
{
    RH_TRY(enter(c));
    if (!rng || k < 0 || drawN < 2 || drawN > 16 || (k > 0 && (!idx_out || !ok_out))) { rh_set_error("rh_sample_sets: bad arguments (drawN 2..16)"); return RH_E_INVALID; }
    if (k == 0) return RH_OK;
    if (c->n == 0) { rh_set_error("rh_sample_sets: the cloud is empty"); return RH_E_INVALID; }
    if (!c->select_valid) RH_TRY(rhk_build_select(c));
    int32_t count = 0;
    RH_HIP(hipMemcpyAsync(&count, c->d_total, sizeof count, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    if (count <= 0) { rh_set_error("rh_sample_sets: no enabled point (the reference would draw for ever, fitting.jl:393)"); return RH_E_INVALID; }
    const int W = drawN + 2;
    constexpr int64_t L_MAX = 1 << 20;
    // draws a call takes: 1 / (enabled share) for the first point + drawN - 1 (+ the rare redraw)
    const double per_call = (double)c->n / (double)count + (double)drawN;
    double grow = 1.5;
    int32_t done = 0;
    while (done < k) {
        int64_t L64 = (int64_t)(grow * per_call * (k - done)) + 8 * drawN + 64;
        if (L64 > L_MAX) L64 = L_MAX;
        const int32_t L = (int32_t)L64;
        // the draws of a COPY of the generator (an injected stream first, like rh_rng_range); the caller's own generator is
        // then advanced by exactly what the calls that completed consumed
        rh_rng tmp = *rng;
        const size_t bytes_raw = sizeof(uint64_t) * (size_t)L, bytes_rec = sizeof(int64_t) * (size_t)L * (size_t)W;
        RH_TRY(rh_ensure_pin(c, (int64_t)(bytes_raw + bytes_rec)));
        uint64_t *h_raw = (uint64_t *)c->h_pin;
        int64_t *h_rec = (int64_t *)(h_raw + L);
        for (int32_t i = 0; i < L; i++) h_raw[i] = rh_rng_next_raw(&tmp);
        const int64_t need = (int64_t)L * (1 + W);
        if (need > 2 * c->ranks_cap) {   // (d_ranks holds 2 x ranks_cap words)
            RH_HIP(hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_ranks);
            c->d_ranks = nullptr;
            c->ranks_cap = 0;
            RH_TRY(dev_alloc(&c->d_ranks, 2 * ((need + 1) / 2)));
            c->ranks_cap = (need + 1) / 2;
        }
        uint64_t *d_raw = (uint64_t *)c->d_ranks;
        int64_t *d_rec = c->d_ranks + L;
        RH_HIP(hipMemcpyAsync(d_raw, h_raw, bytes_raw, hipMemcpyHostToDevice, c->stream));
        RH_TRY(rhk_sample_sets_seq(c, d_raw, L, drawN, d_rec));
        RH_HIP(hipMemcpyAsync(h_rec, d_rec, bytes_rec, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipStreamSynchronize(c->stream));
        int64_t p = 0;
        const int32_t done_before = done;
        while (done < k && p < L) {
            const int64_t *r = h_rec + p * W;
            if (r[drawN] <= 0) break;   // the draws ran out inside this call: it is played again from here in the next round
            for (int q = 0; q < drawN; q++) idx_out[(int64_t)done * drawN + q] = r[q];
            ok_out[done] = (int32_t)r[drawN + 1];
            if (level_out) level_out[done] = r[drawN + 1] ? 1 : 0;   // argmax(levelweight[1:depth]) = 1 always: SURVEY.md 0.5
            p += r[drawN];
            done++;
        }
        for (int64_t i = 0; i < p; i++) (void)rh_rng_next_raw(rng);
        if (done == done_before) {   // not one call finished inside the window: a longer one, or give up
            if (L64 >= L_MAX) { rh_set_error("rh_sample_sets: a call did not finish within %d draws (%d of %lld points enabled)", L, count, (long long)c->n); return RH_E_INTERNAL; }
            grow *= 4.0;
        }
    }
    return RH_OK;
}

// ------------------------------------------------------------ measurement ----
extern "C" int rh_timer_start(rh_cloud *c)
{
    RH_TRY(enter(c));
    RH_HIP(hipEventRecord(c->ev0, c->stream));
    return RH_OK;
}

extern "C" int rh_timer_stop(rh_cloud *c, float *ms_out)
{
    RH_TRY(enter(c));
    RH_HIP(hipEventRecord(c->ev1, c->stream));
    RH_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    RH_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms_out) *ms_out = ms;
    return RH_OK;
}

extern "C" int rh_cloud_set_stream(rh_cloud *c, void *hip_stream, int use_external)
{
    RH_TRY(enter(c));
    RH_HIP(hipStreamSynchronize(c->stream));   // nothing of this cloud may still be in flight on the old stream
    c->stream = use_external ? (hipStream_t)hip_stream : c->own_stream;
    return RH_OK;
}

extern "C" int rh_cloud_sync(rh_cloud *c)
{
    RH_TRY(enter(c));
    RH_HIP(hipStreamSynchronize(c->stream));
    return RH_OK;
}

extern "C" int rh_dev_alloc(rh_cloud *c, int64_t bytes, void **d_out)
{
    RH_TRY(enter(c));
    if (!d_out || bytes < 0) { rh_set_error("rh_dev_alloc: bad arguments"); return RH_E_INVALID; }
    char *p = nullptr;
    RH_TRY(dev_alloc(&p, bytes));
    *d_out = p;
    return RH_OK;
}

extern "C" int rh_dev_free(rh_cloud *c, void *d)
{
    RH_TRY(enter(c));
    RH_HIP(hipStreamSynchronize(c->stream));
    RH_HIP(hipFree(d));
    return RH_OK;
}

extern "C" int rh_dev_upload(rh_cloud *c, void *d_dst, const void *h_src, int64_t bytes)
{
    RH_TRY(enter(c));
    if (bytes <= 0) return RH_OK;
    RH_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    return RH_OK;
}

extern "C" int rh_dev_download(rh_cloud *c, void *h_dst, const void *d_src, int64_t bytes)
{
    RH_TRY(enter(c));
    if (bytes <= 0) return RH_OK;
    RH_HIP(hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    return RH_OK;
}
