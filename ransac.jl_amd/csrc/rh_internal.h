// rh_internal.h -- internal declarations of libransac_hip.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "ransac_hip.h"

// ---- error plumbing -------------------------------------------------------
void rh_set_error(const char *fmt, ...);

#define RH_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            rh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                         __LINE__);                                                         \
            return RH_E_NODEVICE;                                                           \
        }                                                                                   \
    } while (0)

#define RH_TRY(call)                 \
    do {                             \
        int rc_ = (call);            \
        if (rc_ != RH_OK) return rc_; \
    } while (0)

// ---- options (options.cpp): what rh_set_option stores; the diag build also reads the RH_* environment ---------
enum rh_opt_id {
    RH_OPT_SCORE_PATH = 0,      // product keys (rh_set_option, include/ransac_hip.h)
    RH_OPT_S4_ROWS,
    RH_OPT_UNP_WORDS,
    RH_OPT_ST_CULL,
    RH_OPT_REFIT_PATH,
    RH_OPT_BATCHES_IN_FLIGHT,
    RH_OPT_N_PRODUCT,
    // A/B switches and diagnostics: alive in the diag build only (in the product build they read as unset)
    RH_OPT_CREATE_PROF = RH_OPT_N_PRODUCT, RH_OPT_AABB_HOST, RH_OPT_SUB_ORDER, RH_OPT_KD_HOST, RH_OPT_NO_SPREAD, RH_OPT_OCT_CHAIN_W,
    RH_OPT_NO_MANAGED_STORE, RH_OPT_OCT_ONE_WINDOW, RH_OPT_OCT_WINDOW_ITERS, RH_OPT_NO_FUSED_SCORE, RH_OPT_NO_PIPELINE,
    RH_OPT_NO_OCT_CHAIN, RH_OPT_REFIT_BLOCKS, RH_OPT_NO_FUSED_SAMPLER, RH_OPT_NO_CREC, RH_OPT_LONG_WINDOW_SETS, RH_OPT_NO_OCT_TAB,
    RH_OPT_NO_DRIVER_CACHE, RH_OPT_HOST_SAMPLER, RH_OPT_DRIVER_PROF, RH_OPT_G2_DBG, RH_OPT_KREFIT_DBG, RH_OPT_NO_FAST_EXTRACT,
    RH_OPT_COUNT
};
struct rh_cloud;
int64_t rh_opt(const rh_cloud *c, int id);               // cloud -> process -> (diag) environment; RH_OPTION_UNSET = nobody said
void rh_opt_init_cloud(rh_cloud *c);
#ifdef RH_DIAG
const char *rh_opt_env_string(const char *name);         // string-valued diagnostics (RH_RCCL_LIB): diag build only
static inline bool rh_opt_on(const rh_cloud *c, int id) { const int64_t v = rh_opt(c, id); return v != RH_OPTION_UNSET && v != 0; }
static inline int64_t rh_opt_int(const rh_cloud *c, int id, int64_t dflt) { const int64_t v = rh_opt(c, id); return v == RH_OPTION_UNSET ? dflt : v; }
#else
// product build: the diag ids are compile-time "unset" -- the A/B branches fold away
static inline bool rh_opt_on(const rh_cloud *c, int id) { if (id >= RH_OPT_N_PRODUCT) return false; const int64_t v = rh_opt(c, id); return v != RH_OPTION_UNSET && v != 0; }
static inline int64_t rh_opt_int(const rh_cloud *c, int id, int64_t dflt) { if (id >= RH_OPT_N_PRODUCT) return dflt; const int64_t v = rh_opt(c, id); return v == RH_OPTION_UNSET ? dflt : v; }
#endif

// ---- device-side candidate record ------------------------------------------
// Per-candidate constants hoisted out of the per-point loop.  Every hoisted value is
// a pure function of the candidate computed with the same IEEE operations the
// reference evaluates per call, so hoisting does not change a single bit.
//   plane    f[0..2]=point  f[3..5]=normal f[6..8]=normalize(normal)   (plane.jl:85)
//   sphere   f[0..2]=center f[3]=R  f[4]=sgn
//   cylinder f[0..2]=axis   f[3..5]=center f[6]=R f[7]=sgn
//   cone     f[0..2]=apex   f[3..5]=axis   f[6]=cos(-w/2) f[7]=sin(-w/2) f[8]=sgn
// sgn = +1.0 (outwards) / -1.0 (inwards); multiplying by it is exact.
struct rh_prep {
    double f[12];
};

// score kernel geometry (kernels.hip)
constexpr int RH_SC_THREADS = 256;
constexpr int RH_SC_PPT = 4;                             // points per lane
constexpr int RH_SC_WAVE_PTS = 64 * RH_SC_PPT;           // contiguous points per wave per tile
constexpr int RH_SC_TILE = RH_SC_THREADS * RH_SC_PPT;    // points per block per tile
constexpr int RH_SC_CT = 64;                             // candidates per block
constexpr int RH_WORDS_PER_BLOCK = 1024;                 // scan granularity (64-bit words)
constexpr int RH_G2_TG = 4;                              // 64-point groups per LDS tile of the culled score kernel
constexpr int RH_G2_TILE = RH_G2_TG * 64;
constexpr int64_t RH_G2_MIN_POINTS = 8192;               // below this the brute-force kernel is used
constexpr int64_t RH_KREFIT_MIN = 1 << 21;               // clouds from this size on take the culled refit scan (korder.hip): 1M points 17 us culled against 13 us streaming, 10M 24 against 80

// ---- the cloud --------------------------------------------------------------
struct rh_oct_state;
struct rh_s4_points {   // what the v4 score kernel runs over (score4.hip): points as six planes `stride` apart, 64-point groups + their boxes
    const double *pts;
    int64_t stride, s, ngroups;
    const float *gb32;
};
// rh_score_batch_dev with two batches in flight (rh_set_option "batches_in_flight" = 2): batches take turns on the cloud's own
// workspaces and stream and on these, so its prepare + score launches fill the chip while the previous batch's launch drains
#define RH_MAX_IN_FLIGHT 4
struct rh_batch_slot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr, start = nullptr;
    int64_t batch_cap = 0;
    rh_shape *d_shapes = nullptr;
    rh_prep *d_prep = nullptr;
    int32_t *d_orig = nullptr, *d_counts = nullptr, *d_nk2 = nullptr;
    void *d_qpre = nullptr, *d_prep32 = nullptr;
    float *d_box = nullptr;
    uint16_t *d_stlist = nullptr;      // (the super-tile lists of the slot's batch)
    int32_t *d_stcount = nullptr;
    int64_t stlist_cap = 0, stlist_nst = 0;
    uint64_t *d_masks_int = nullptr;   // (batches with mask output: the entry lists and their cursors)
    uint8_t *d_occ = nullptr;
    int64_t masks_int_cap = 0, occ_cap = 0, mstride4 = 0;
    int nk2_flip = 0;
    bool nk2_ready = false, qpre_v4 = false;
};
struct rh_cloud {
    int64_t opt[RH_OPT_COUNT];         // rh_set_option on this cloud (RH_OPTION_UNSET: the process-wide value holds); rh_opt_init_cloud
    int device = -1;
    hipStream_t stream = nullptr;      // the stream every launch / copy of this cloud goes to
    hipStream_t own_stream = nullptr;  // created with the cloud; `stream` is this or the caller's (rh_cloud_set_stream)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evk[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };   // per-kind launch brackets
    hipStream_t copy_stream = nullptr; // rh_ransac: read-back of the extracted index lists, beside the compute stream
    hipEvent_t ev_copied = nullptr;    // idx_out has been read back
    hipEvent_t ev_sync = nullptr;      // rh_ransac: the host's one wait per extraction
    int64_t n = 0, s = 0;
    int64_t n_pad = 0, s_pad = 0;      // padded to RH_SC_TILE
    int64_t nwords = 0, swords = 0;    // ceil(n/64), ceil(s/64)
    int64_t nblocks = 0;               // ceil(nwords / RH_WORDS_PER_BLOCK)

    // HBM layout: six SoA planes of n_pad (resp. s_pad) doubles: x y z nx ny nz
    double *full = nullptr;            // full cloud, original order
    double *set_ws = nullptr;          // sampler -> fitter hand-over: gathered minimal sets, [drawN * 6][sets]
    int32_t *set_level = nullptr;      // per set: octree level it was drawn from, 0 = no set
    int64_t set_ws_sets = 0, set_ws_doubles = 0;
    double *crec = nullptr;            // 64-byte records of the ENABLED points in rank order (long sampling windows)
    int64_t crec_cap = 0;              // in records
    bool crec_valid = false;           // cleared whenever the select list is rebuilt
    int very_long_windows = 0;         // windows of >= 2^19 sets sampled since the select directory was last built
    int32_t *sel_list = nullptr;       // sel_list[r] = 0-based index of the (r+1)-th enabled point (valid with sel_valid)
    bool sel_valid = false;            // built on demand by rhk_build_sel_list, dropped with the select directory
    double *rec = nullptr;             // the same points as 64-byte records (x y z nx ny nz 0 0): one line per random gather
    double *sub = nullptr;             // subset 1, subset order
    double *dis = nullptr;             // disabled subset-1 points (append-only), capacity s_pad + tile
    int64_t dis_stride = 0;
    int32_t *sub_idx0 = nullptr;       // [s] 0-based original index of subset position j
    uint64_t *enabled = nullptr;       // [nwords]  pc.isenabled chunks
    uint64_t *sub_enabled = nullptr;   // [swords]  enabled bits gathered into subset order
    int32_t *sub_perm = nullptr;       // [s] internal (k-d leaf order) position -> subset position j
    double *gb = nullptr;              // 7 planes x ng_pad: box centre cx cy cz, half extents hx hy hz, radius hr
    int64_t ngroups = 0, ng_pad = 0;   // 64-point groups of the subset (internal order)
    double create_ms[4] = { 0, 0, 0, 0 };   // rh_cloud_create: total, host k-d leaf order of subset 1, everything before it, everything after it
    double coord_mag = 0;              // max |coordinate| over the subset (rounding slack of the bounds)
    double nrm_mag = 0;                // max |normal component| over the subset (margins of the binary32 classifier)
    bool use_groups = false;           // culled scoring path available (s large enough)
    float *gb32 = nullptr;             // the same boxes in binary32, 8 floats per group (v4 score kernel: scalar loads)
    hipEvent_t ev_cull = nullptr;      // timed launches (rh_score_batch_dev_timed): behind the list launch, in front of the score launch
    bool time_cull = false;
    float last_cull_ms = 0.f;
    int32_t last_s4[4] = { 0, 0, 0, 0 };   // the last sized launch of the culled score kernel: R, lists taken (0 / 1), rows, tiles (rh_score_launch_info)
    float *st32 = nullptr;             // boxes of the super-tiles (16 consecutive groups = 4 tiles of the k-d leaf order), 8 floats each
    int64_t nst = 0;
    // per-super-tile candidate lists of the batch being scored (score4.hip, st_cull_kernel): [nst][4 kinds][stlist_cap] slots, [nst][4] counts
    uint16_t *d_stlist = nullptr;
    int32_t *d_stcount = nullptr;
    int64_t stlist_cap = 0, stlist_nst = 0;
    double *dis_gb = nullptr;          // boxes of the dis segment in use (7 x ng_pad)
    int32_t *d_ndis = nullptr;         // device counter: entries in dis
    int64_t n_dis = 0;                 // host mirror

    // The cloud in Morton order (korder.hip), built on the device with the cloud: the order of the linear octree of
    // octree_sampling = 1 and of the culled refit scan.
    bool k_built = false;
    bool k_men_valid = false;          // oct_men mirrors `enabled` (else it is regathered on the next use)
    bool k_applied = false;            // the refit scan in flight has already cleared its inliers in oct_men
    bool k_sums_ready = false;         // ... and left the per-block popcounts of refit_mask in block_sums
    double k_lo[3] = { 0, 0, 0 }, k_size = 1;   // bounding cube of the cloud (findAABB, utilities.jl:125-136)
    double k_mag = 0;                  // max |coordinate| over the cloud (rounding slack of the box tests)
    double *fullk = nullptr;           // 6 planes x n_pad, Morton order
    float *fullk32 = nullptr;          // the same as float (Float32 clouds)
    double *kgb = nullptr;             // 7 planes x kg_pad: box of every 64 consecutive points of fullk
    int64_t kg_pad = 0;
    int32_t *klist = nullptr;          // [nwords] groups that survive the box test of the scan in flight
    int32_t *kctr = nullptr;           // [2] list length (zero between scans)
    uint8_t *kflag = nullptr;          // [n_pad] one byte per point, original order: inlier of the scan in flight (zero between scans)
    bool oct_built = false;            // depth + host twins of the octree (rh_octree_ensure)
    int oct_depth = 0, oct_max_depth = 0;
    uint64_t *oct_code = nullptr;      // [n] sorted Morton codes (device)
    int32_t *oct_perm = nullptr;       // [n] Morton position -> original index0
    int32_t *oct_pos = nullptr;        // [n] original index0 -> Morton position
    uint64_t *oct_men = nullptr;       // [nwords] enabled bits in Morton order
    int32_t *oct_prefix = nullptr;     // [nwords + 1]
    rh_oct_state *oct_state = nullptr; // chained octree windows: the window's state,
    const int32_t *s4_stop = nullptr;  //   its stop flag as the score kernel sees it (null outside such windows),
    const rh_s4_points *s4_points = nullptr;   // the launches being queued run over this set instead of subset 1 (rhk_score4_dis)
    float *dis_gb32 = nullptr;         // binary32 twins of dis_gb
    int32_t *zero_extra = nullptr;     // the next rhk_prep_binned launch also zeroes zero_extra_n ints from here (then forgets it)
    int32_t zero_extra_n = 0;
    void *s4_stats = nullptr;          // diag build: event counters of the score launches (rh_dbg_s4_stats), 128 x u64 on the device
    bool s4_open_count = false;        // the candidate count of the score launches being queued is a guess (windows of the candidate loop)
    int32_t *oct_adv_tab = nullptr;    //   the (level, slot) table of rhk_oct_advance, its bitmap (kept zero) and the sorted scores
    unsigned long long *oct_adv_bits = nullptr;
    int64_t oct_adv_cells = 0;
    double *oct_adv_E = nullptr;
    uint64_t *oct_code_o = nullptr;    // [n] the Morton codes in original point order (built with oct_tab)
    int32_t *oct_tab = nullptr;        // first Morton position of every level-oct_tab_level cell (+ n at the end)
    int oct_tab_level = 0;
    double *oct_P = nullptr;           // level distributions of a speculation window
    int64_t oct_P_cap = 0;
    std::vector<uint64_t> h_oct_code;  // host twins (host-side sampling, enabled mirror)
    std::vector<int32_t> h_oct_perm, h_oct_pos;

    // refit / select workspaces
    uint64_t *refit_mask = nullptr;    // [nwords]
    int32_t *block_sums = nullptr;     // [nblocks + 1]
    int32_t *en_block_sums = nullptr;  // [nblocks + 1] popcounts of the enabled words per block (select directory)
    bool en_sums_valid = false;        // left behind by rhk_compact_refit_apply for the next rhk_build_select
    int32_t *word_prefix = nullptr;    // [nwords + 1] exclusive prefix of popcount(enabled) (select)
    bool select_valid = false;
    int64_t *idx_out = nullptr;        // [n] compacted indices
    int32_t *d_total = nullptr;        // scalar

    // batch workspaces (grown on demand)
    int64_t batch_cap = 0;
    rh_shape *d_shapes = nullptr;      // [batch_cap]
    rh_prep *d_prep = nullptr;         // [4 * batch_cap], kind-major
    int32_t *d_orig = nullptr;         // [4 * batch_cap]
    int32_t *d_nk = nullptr;           // [4]
    int32_t *d_nk2 = nullptr;          // [8] rh_score_batch_dev: two halves used alternately, each batch zeroes the other
    int nk2_flip = 0;
    bool nk2_ready = false;
    int32_t *d_counts = nullptr;       // [batch_cap]
    uint64_t *d_masks = nullptr;       // grown on demand
    int64_t masks_cap = 0;
    uint64_t *d_masks_int = nullptr;   // masks in internal order (before un-permuting)
    int64_t masks_int_cap = 0;
    uint8_t *d_occ = nullptr;          // v4 score kernel with masks: one int32 cursor per candidate row of the entry lists in d_masks_int (zero between batches)
    uint64_t *unp_segmask = nullptr;   // mask un-permutation: per output segment and internal word, the bits that land in the segment
    int64_t unp_seg_words = 0;         //   ... made for this segment width
    int64_t occ_cap = 0;
    bool masks4 = false;               // the batch being scored leaves its masks in the v4 form (rows mstride4 apart)
    int64_t mstride4 = 0;
    int64_t *d_ranks = nullptr;        // select in/out
    int64_t ranks_cap = 0;

    // Float32 clouds (rh_cloud_create_f32; f32.hip): float copies of the two point sets and float candidate records
    bool f32 = false;
    bool f32_groups = false;           // scored by the culled kernel (exact test in binary32); else by the brute-force float kernel
    float *full32 = nullptr;           // 6 planes x n_pad, original order (the refit scan streams these: 24 B per point)
    float *sub32 = nullptr;            // 6 planes x s_pad, subset 1 in k-d leaf order
    void *d_prep32 = nullptr;          // [4 * batch_cap] float records (rh_prepf), grown with the batch workspaces
    void *d_qpre = nullptr;            // [4 * batch_cap] classifier records (rh4::rh_cls, 64 B) of the bins in d_prep; culling records: d_box
    float *d_box = nullptr;            // [RH_BOX_FIELDS][4 * batch_cap] culling records of the same bins (v4 score kernel), structure of arrays
    bool qpre_v4 = false;              // ... made by the last prep kernel (for the thresholds it was given)
    int32_t *d_zero = nullptr;         // 64 zero bytes: the bin size a kind left out of a launch reads (per-kind timing leg)
    const rh_shape *f32_shapes = nullptr;   // the batch being scored: its shapes on the device ...
    int f32_via_orig = 0;                   // ... indexed through d_orig (caller's order) or directly (sorted like the bins)

    // rh_ransac_mp: this process's share of every iteration's minimal sets (set j belongs to rank j % world)
    int32_t mp_rank = 0, mp_world = 1;

    // rh_ransac's reusable buffers, parked here between calls (driver.hip owns the layout and the deleter)
    void *drv_cache = nullptr;
    void (*drv_cache_free)(rh_cloud *, void *) = nullptr;

    // rh_score_batch with host buffers: device twin of the pinned staging block of a small batch (one upload)
    void *d_stage = nullptr;

    // two batches in flight (rh_score_batch_dev, "batches_in_flight" = 2)
    rh_batch_slot alt[RH_MAX_IN_FLIGHT - 1];
    uint32_t pipe_k = 0;               // batches since the pipeline (re)started
    const void *slot_counts[RH_MAX_IN_FLIGHT] = {}, *slot_masks[RH_MAX_IN_FLIGHT] = {};   // the buffers the slots' batches in flight write (null: none)
    bool alt_dirty[RH_MAX_IN_FLIGHT - 1] = {};     // work on alt[i].stream the cloud's own stream has not waited for yet
    bool alt_started[RH_MAX_IN_FLIGHT - 1] = {};   // alt[i].stream has been ordered behind the cloud's stream since the last join

    // pinned staging
    void *h_pin = nullptr;
    int64_t h_pin_cap = 0;
};

// liveness pass over a small store (rhk_liveness_small): per kind the prepared candidates, their number, where
// their flags start, and the first entry of the disabled list they have to be tested against
struct rh_live_args {
    const rh_prep *prep[4];
    int32_t nk[4], base[4];
    int64_t first[4];
    double eps[4], cosa[4];
    int32_t f32 = 0;     // Float32 cloud: the binary32 tests
};

// ---- kernel launchers (kernels.hip) -----------------------------------------
int rhk_transpose_aos(rh_cloud *c, const double *d_aos_xyz, const double *d_aos_nrm, int64_t n,
                      const int32_t *d_gather_or_null, int64_t count, double *dst, int64_t dst_stride);
int rhk_fetch2_i32(rh_cloud *c, const int32_t *d_src0, const int32_t *d_src1, int32_t *h_pinned_dst);   // h[0..1] = *d0, *d1 in stream order (pinned h)
int rhk_pack_records(rh_cloud *c, const double *d_xyz, const double *d_nrm, int64_t n, double *d_rec);
int rhk_prep_sorted(rh_cloud *c, const rh_shape *d_shapes_sorted, int32_t b, rh_prep *d_prep,
                    int32_t *d_counts_to_zero = nullptr, const double *eps = nullptr, const double *cosa = nullptr);
struct rh_cand_entry;
struct rh_oct_state;
int rhk_prep_entries(rh_cloud *c, const rh_cand_entry *d_entries, const int32_t *d_count, int32_t cap_entries,
                     int32_t launch_bound, int32_t *d_counts, int nk_is_zero, const double *eps = nullptr,
                     const double *cosa = nullptr, const rh_oct_state *ost = nullptr);   // eps + cosa: also leave the classifier / culling records in d_qpre / d_box
int rhk_prep_binned(rh_cloud *c, const rh_shape *d_shapes, int32_t b, rh_prep *d_prep, int32_t *d_orig,
                    int32_t *d_nk, int64_t cap, int32_t *d_counts_to_zero, int32_t *d_nk_other, int nk_is_zero,
                    const double *eps = nullptr, const double *cosa = nullptr);   // eps + cosa: also leave the classifier / culling records (bins in c->d_prep only)
bool rh_score_v4_enabled(const rh_cloud *c);
int rhk_score4_all(rh_cloud *c, const uint64_t *const en[4], const rh_prep *const prep[4], const void *const cls[4],
                   const float *const box[4], int64_t bstride, const int32_t *const orig[4], const int32_t *const nk[4],
                   int32_t nk_total_bound, const double eps[4], const double cosa[4], int32_t *d_counts,
                   uint64_t *d_masks_int = nullptr, uint8_t *d_occ = nullptr, int64_t mstride = 0);   // masks: entry lists + row cursors
int rhk_unpermute_masks4(rh_cloud *c, const uint64_t *d_in, uint8_t *d_occ, int64_t mstride, int32_t b, uint64_t *d_out);   // score4.hip
// score nk candidates of one kind; nk_host < 0: count is only known on the device (d_nk),
// launch for an upper bound of nk_bound candidates
int rhk_score_kind(rh_cloud *c, int kind, const double *pts, int64_t stride, int64_t s,
                   const uint64_t *enabled_words_or_null, const rh_prep *d_prep, const int32_t *d_orig,
                   const int32_t *d_nk, int32_t nk_bound, double eps, double cosa, int32_t *d_counts,
                   uint64_t *d_masks_or_null, int64_t mask_stride);
// culled path over c->sub (k-d leaf order + per-group boxes): all kinds in one launch of score4.hip's kernel; masks
// (optional) leave it as entry lists (rhk_unpermute_masks4)
int rhk_score_all_groups(rh_cloud *c, const uint64_t *const en[4], const rh_prep *const prep[4],
                         const int32_t *const orig[4], const int32_t *const nk[4], int32_t nk_total_bound,
                         const double eps[4], const double cosa[4], int32_t *d_counts, uint64_t *d_masks_int,
                         const void *const cls[4],                 // cls / box: classifier and culling records (score4_device.h) of the
                         const float *const box[4], int64_t bstride);   // same bins and thresholds
int rhk_gb32_build(rh_cloud *c);
int rhk_store_cls(rh_cloud *c, const rh_prep *const prep[4], const int32_t n[4], const int32_t pbase[5], const double eps[4],
                  const double cosa[4], void *d_cls, float *d_box, int64_t bstride);
int rhk_score4_dis(rh_cloud *c, int64_t first, int64_t cnt, const rh_prep *const prep[4], const void *const cls[4], const float *const box[4],
                   int64_t bstride, const int32_t *const orig[4], const int32_t *const nk[4], int32_t nk_total_bound, const double eps[4],
                   const double cosa[4], int32_t *d_counts);
// (score4.hip: gb32 from gb)
int rhk_prep_f32(rh_cloud *c, const rh_shape *d_shapes, int via_orig, const int32_t *d_orig, const int64_t off[4],
                 const int32_t *d_nk, int32_t nmax);               // fills c->d_prep32 (f32.hip)
// Float32 clouds (f32.hip)
int rhk_f32_build(rh_cloud *c);
int rhk_score_all_f32(rh_cloud *c, const rh_shape *d_shapes, int via_orig, const uint64_t *const en[4], const int32_t *d_orig,
                      const int64_t off[4], const int32_t *d_nk, const int32_t nk_bound[4], const double eps[4],
                      const double cosa[4], int32_t *d_counts, uint64_t *d_masks_int);
int rhk_refit_mask_f32(rh_cloud *c, const rh_shape &shape, double eps, double cosa, bool apply = false);
int rhk_cloud_aabb(rh_cloud *c, const double *d_xyz, int64_t n, double lo[3], double hi[3], bool has[3], double *mag);   // kdorder.hip
int rhk_kd_order(rh_cloud *c, const double *d_xyz, const double *d_nrm, const int32_t *d_idx0);   // kdorder.hip: subset 1's k-d leaf order on the device
int rhk_score_kind_dis(rh_cloud *c, int kind, int64_t first, int64_t cnt, const rh_prep *d_prep, const int32_t *d_orig,
                       const int32_t *d_nk, int32_t nk_bound, double eps, double cosa, int32_t *d_counts);
int rhk_group_bounds(rh_cloud *c);
int rhk_unpermute_masks(rh_cloud *c, const uint64_t *d_in, int32_t b, uint64_t *d_out);
// refit_mask = the shape's inliers among the enabled points (original order); `apply`: the caller follows up with
// rhk_compact_refit_apply, so a culled scan may clear the Morton-order enabled bits on the way
int rhk_refit_mask(rh_cloud *c, const rh_prep &P, int kind, double eps, double cosa, bool apply = false);
// korder.hip
int rhk_korder_build(rh_cloud *c, const double *d_xyz, const double *d_nrm, const double lo[3], double size, double mag);
int rhk_korder_build_f32(rh_cloud *c);
int rhk_korder_sync_enabled(rh_cloud *c);
bool rhk_refit_is_culled(const rh_cloud *c);
int rhk_refitk_mask(rh_cloud *c, const rh_prep &P, int kind, double eps, double cosa, bool apply);
int rhk_refitk_mask_f32(rh_cloud *c, const void *prepf, const rh_prep &P, int kind, double eps, double cosa, bool apply);
int rhk_group_bounds_of(rh_cloud *c, const double *pts, int64_t stride, int64_t count, int64_t ngroups, double *gb, int64_t gstride);
int rhk_oct_build_tab(rh_cloud *c);                                     // the sampler's cell directory (needs oct_depth)
int rhk_oct_gather_enabled(rh_cloud *c);                                // oct_men = enabled in Morton order
int rhk_compact_mask(rh_cloud *c, const uint64_t *mask, int64_t nwords, int64_t *idx_out, int64_t cap,
                     int32_t *d_total);
int rhk_invalidate_idx(rh_cloud *c, const int64_t *d_idx, int64_t n);
int rhk_rebuild_sub_enabled(rh_cloud *c, bool reset_list);
int rhk_compact_refit_apply(rh_cloud *c);
int rhk_build_sel_list(rh_cloud *c);
int rhk_liveness_small(rh_cloud *c, int64_t lo, int64_t span, const rh_live_args &A, int32_t *d_flags);   // flags must be zero on entry
int rhk_pack_live(rh_cloud *c, int32_t *d_flags, int32_t n_flags, int32_t *h_flags, int32_t *h_scalars);
int rhk_build_select(rh_cloud *c);
int rhk_select(rh_cloud *c, const int64_t *d_ranks, int32_t k, int64_t *d_out);
int rhk_count_enabled(rh_cloud *c, int64_t *out);
uint64_t rh_rng_next_raw(rh_rng *r);   // fit.cpp: the next raw 64-bit draw (an injected stream first)
int rhk_sample_sets_seq(rh_cloud *c, const uint64_t *d_raw, int32_t L, int32_t drawN, int64_t *d_rec);   // rh_sample_sets
int rhk_iota(rh_cloud *c, int32_t *d, int32_t n, int32_t base);
int rhk_gather_prep(rh_cloud *c, const rh_prep *src, const int32_t *d_idx, int32_t n, rh_prep *dst);
// removeinvalidshapes! (fitting.jl:209-221) on a device-managed store (driver.hip, chained octree windows): the store's
// entries carry the host's candidate number (id); an entry dies when its liveness count is non-zero or it is the
// extracted candidate.  The survivors move to the spare arrays in order, the ids of the dead ones go to the host.
// Index space: the kinds laid end to end, each padded to a multiple of RH_STORE_PAD (pbase[q]; pbase[4] = the end).
constexpr int RH_STORE_PAD = 256;
struct rh_store_plan {
    const rh_prep *prep[4];
    rh_prep *spare[4];
    const int32_t *id[4];
    int32_t *spare_id[4];
    const double *E[4];          // the entries' scores: the first maximum over the survivors is found on the way
    double *spare_E[4];
    int32_t n[4], pbase[5];
    const int32_t *counts;       // liveness counts at pbase[q] + slot
    int32_t extracted_id;
};
// d_work: 2 * (pbase[4] / RH_STORE_PAD) + 16 ints of scratch; h_out (pinned): [0..3] the kinds' new lengths, [4] the number
// of dead entries; h_dead (pinned): their ids, in no particular order; h_best (pinned, one per block of RH_STORE_PAD
// entries): the block's best survivor -- greatest score, smallest id among equals; id < 0: none
struct rh_store_best { double E; long long id; };
// d_dead (>= the store's entries), d_best (one per block): device staging of the two lists (flushed to the pinned ones by a last small kernel)
int rhk_store_compact(rh_cloud *c, const rh_store_plan &P, int32_t *d_work, int32_t *h_out, int32_t *h_dead, rh_store_best *h_best,
                      int32_t *d_dead, rh_store_best *d_best);

int rhk_compact_generic(hipStream_t stream, const uint64_t *mask, int64_t nwords, int32_t *ws_block_sums,
                        int64_t *idx_out, int64_t cap, int32_t *d_total);

// device-side sampling + fitting (sampler.hip)
struct rh_cand_entry {
    int64_t slot;     // ((iteration - k0) * minsubsetN + set) * n_shape_types + type index
    int32_t level;    // octree level the set was drawn from (1 = root)
    int32_t pad;
    rh_shape shape;
};
// State of a CHAINED octree window on the device (driver.hip, run_streams_device): the iterations of the window are
// queued back to back -- sample, fit, prepare, score, rhk_oct_advance -- and the level scores / level distribution
// the next iteration samples from are advanced on the device, so the host is not in the loop between iterations.
struct rh_oct_state {
    double S[32], P[32];          // pc.levelscore, pc.levelweight: as the NEXT iteration of the window finds them
    double best_E;                // the best stored score (findhighestscore), has_best != 0
    long long store_count, cc2;   // stored candidates / candidates scored so far
    rh_prep *store_prep[4];       // the device store of prepared candidates (driver.hip): arrays per kind, their capacities and
    double *store_E[4];           //   (with the candidates' scores)
    int32_t *store_id[4];         //   fill -- every iteration appends its candidates' records and their numbers on the host
    long long store_cap[4];       //   (appended + the candidate's rank in candidate order)
    long long appended;
    int32_t store_n[4];
    int32_t has_best;
    int32_t stop;                 // an iteration's extraction test passed (approximately: the host decides): the rest of the window is skipped
    int32_t start;                // list position where the entries of the next iteration begin
    int32_t it_done;              // iterations of the window that ran
};
// d_P: null (root-cell sampling) or n_iters x oct_depth level distributions.  Chained octree windows launch one
// iteration at a time: n_iters = 1, it0 = its index in the window (slots, draw counters), ost = the window's state
// (d_P = ost->P; a set stop flag skips the launch).
int rhk_sample_fit(rh_cloud *c, const rh_params *prm, uint64_t seed, int64_t k0, int32_t n_iters, int32_t n_enabled,
                   const double *d_P, rh_cand_entry *d_out, int32_t cap, void *d_status, int status_is_zero,
                   int32_t *d_nk_zero, int32_t it0 = 0, const rh_oct_state *ost = nullptr);
// What the host gets per iteration of a chained window (pinned memory, written by rhk_oct_advance's kernel)
struct rh_oct_iter_hdr {
    int32_t skipped;              // the window had ended before this iteration: nothing else is valid
    int32_t overflow;             // bit 0: the candidate list is full, bit 1: the device store is: the iteration is incomplete (the window ends here)
    int32_t gave_up;              // sampling found no enabled point
    int32_t start, end;           // the iteration's entries in the list (and in the pinned copies of the list and the counts)
    int32_t stop_after;           // the device expects an extraction after this iteration (it skips the rest of the window)
    int32_t pad[2];
    unsigned long long draws;     // random numbers the iteration consumed
    double P[32];                 // the level distribution after the iteration
#ifdef RH_OCT_TIMING
    unsigned long long t[8];      // phase times of the kernel (100 MHz ticks)
#endif
};
// end of an iteration of a chained octree window: levelscore[level] += E(candidate) in candidate order (fitting.jl:184),
// updatelevelweight (octree.jl:198-205), the running best score and the (approximate) extraction test; the iteration's
// entries and counts are copied to h_entries / h_counts (same positions as in the list), h_rank gets every entry's rank
// in candidate order within the iteration, the prepared records go behind the device store (ost->store_*) with h_slot
// saying where, and h_hdr[it] is filled.
// it = index of the iteration in the window, k = its number.
int rhk_oct_window_begin(rh_cloud *c, rh_oct_state *ost);   // a window that continues from the device's state: list position 0
int rhk_oct_advance(rh_cloud *c, const rh_params *prm, rh_oct_state *ost, const rh_cand_entry *d_entries, const void *d_status,
                    int32_t cap, const int32_t *d_counts, int32_t it, int64_t k, rh_cand_entry *h_entries, int32_t *h_counts,
                    int32_t *h_rank, int32_t *h_slot, rh_oct_iter_hdr *h_hdr);
// status block, head of the list and of its counts -> pinned host memory; zeroes the status block
int rhk_pack_window(rh_cloud *c, void *d_status, int32_t n_iters, const rh_cand_entry *d_entries, const int32_t *d_counts,
                    int32_t head_cap, void *h_status, void *h_entries, int32_t *h_counts);
int rh_octree_ensure(rh_cloud *c, const double *xyz, int max_depth);   // cloud.hip
int rhk_oct_sync_enabled(rh_cloud *c);                                  // men = permuted enabled; prefix
int rhk_oct_clear_mask(rh_cloud *c, const uint64_t *mask);             // clear the bits of an original-order mask
int rhk_word_prefix(rh_cloud *c, const uint64_t *words, int64_t nwords, int32_t *prefix_out);

void rh_prep_host(const rh_shape &s, rh_prep *out);
int32_t rh_spread_multiplier(int32_t b);   // t -> (t * m) mod b: a permutation of the batch that separates neighbours

// ---- host helpers (cloud.hip) -----------------------------------------------
int rh_ensure_batch(rh_cloud *c, int64_t b);
int rh_cloud_join(rh_cloud *c);     // the same with the cloud's device made current first (options.cpp)
int rh_join_batches(rh_cloud *c);   // the cloud's stream waits for the second batch slot ("batches_in_flight")
int rh_ensure_masks(rh_cloud *c, int64_t words);
int rh_ensure_pin(rh_cloud *c, int64_t bytes);
int rh_validate_params(const rh_params *p);
