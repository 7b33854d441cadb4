// sampler.hip -- device-side minimal-set sampling + fitting (sampling_streams = 1).
// One thread per (iteration, minimal set): draws the set from its own splitmix64 stream
// (fit_shared.h) and gathers the points (sample_sets_kernel); a second kernel runs the plane /
// sphere / cylinder / cone fits in the order of iteration.shape_types (forcefitshapes!,
// src/fitting.jl:165-173) and appends every fitted candidate, tagged with its slot =
// ((iteration, set), shape type), to a compact list.  The host sorts the list by slot, which
// restores the reference's candidate order exactly.
#include "fit_shared.h"
#include "rh_internal.h"

namespace {

// The minimal sets of an iteration this process draws: j = lo + jl * step, jl < m_local.  One process: (0, 1,
// minsubsetN).  rh_ransac_mp deals the sets of every iteration round-robin to the processes that share a scene
// (driver.hip): a set's draws are a pure function of (seed, iteration, j), so who draws it changes nothing.
struct SetShard {
    int32_t lo, step, m_local;
};

struct DevEnabled {
    const uint64_t *w;
    const int32_t *prefix;   // exclusive popcount prefix per word
    int64_t nwords;
    int32_t total;
    __device__ bool test(int64_t i0) const { return (w[i0 >> 6] >> (i0 & 63)) & 1ULL; }
    // r-th enabled point (1-based), 0 when out of range.  Long windows read the flat select list (one
    // load; the cloud builds it on demand); short ones search the word prefixes.
    const int32_t *sel;
    __device__ int64_t select(int64_t r) const
    {
        if (r < 1 || r > total) return 0;
        if (sel != nullptr) return (int64_t)sel[r - 1] + 1;
        // the last word whose exclusive prefix is <= r - 1 holds the point.  8-ary search: the seven pivots of a
        // round are independent loads, so 156 250 words take 6 dependent rounds instead of 17
        const int64_t t = r - 1;
        int64_t lo = 0, hi = nwords - 1;
        while (hi - lo >= 8) {
            const int64_t step = (hi - lo + 8) >> 3;
            int k = 0;   // prefix is non-decreasing: the pivots <= t are the first k
#pragma unroll
            for (int i = 1; i < 8; i++) {
                const int64_t pv = lo + i * step;
                k += (pv <= hi && (int64_t)prefix[pv <= hi ? pv : hi] <= t) ? 1 : 0;
            }
            lo += k * step;
            hi = hi < lo + step - 1 ? hi : lo + step - 1;
        }
        {
            int k = 0;
#pragma unroll
            for (int i = 1; i < 8; i++) {
                const int64_t pv = lo + i;
                k += (pv <= hi && (int64_t)prefix[pv <= hi ? pv : hi] <= t) ? 1 : 0;
            }
            lo += k;
        }
        uint64_t m = w[lo];
        for (int64_t k = r - 1 - prefix[lo]; k > 0; k--) m &= m - 1;
        return (lo << 6) + __builtin_ctzll(m) + 1;
    }
};

constexpr int RH_MAX_DRAWN = 8;   // device path; larger minimal sets use the host sampler

// samplepointcloud4! on the root cell in RANK space (see sample_sets_kernel): the first point is drawn by
// rejection on rand(1:n) like the reference and converted to its rank with the per-word popcount prefix; the
// others are ranks already.  Draw count and stream position equal rhfit::sample_minimal_set's.
template <int DN>
__device__ bool sample_ranks(const DevEnabled &en, int64_t n, int64_t n_enabled, int drawN_rt, uint64_t *x, int64_t *sd,
                             uint32_t *ndraws, bool *gave_up, int64_t *first_index = nullptr)
{
    const int drawN = DN > 0 ? DN : drawN_rt;
    if (n_enabled <= 0) return false;
    // rand(1:n) until enabled (fitting.jl:391-394).  The stream advances by a constant per draw, so draw k is a
    // pure function of (x0, k): four draws are tested per round (four independent loads instead of a chain of
    // dependent ones); the accepted index and the number of draws consumed are those of the one-at-a-time loop.
    const uint64_t G = 0x9E3779B97F4A7C15ULL, x0 = *x;
    uint32_t nd = 0;
    int64_t r1 = 0;
    for (;;) {
        int64_t r[4];
        bool e[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            r[i] = 1 + (int64_t)rhfit::mulhi64(rhfit::mix64(x0 + (uint64_t)(nd + 1 + i) * G), (uint64_t)n);
            e[i] = en.test(r[i] - 1);
        }
        int hit = -1;
#pragma unroll
        for (int i = 3; i >= 0; i--) hit = e[i] ? i : hit;
        if (hit >= 0) { nd += (uint32_t)hit + 1; r1 = r[hit]; break; }
        nd += 4;
        if (nd > (1u << 24)) { *gave_up = true; *x = x0 + (uint64_t)nd * G; *ndraws += nd; return false; }
    }
    *x = x0 + (uint64_t)nd * G;
    *ndraws += nd;
    if (n_enabled < drawN) return false;
    const int64_t i0 = r1 - 1;
    if (first_index != nullptr) *first_index = r1;
    sd[0] = (int64_t)en.prefix[i0 >> 6] + __popcll(en.w[i0 >> 6] & ((1ULL << (i0 & 63)) - 1ULL)) + 1;
#pragma unroll
    for (int q = 1; q < drawN; q++) {
        int64_t pick = rhfit::set_stream_range(x, n_enabled);
        ++*ndraws;
        if (pick == sd[0]) {   // one redraw: fitting.jl:416-419
            pick = rhfit::set_stream_range(x, n_enabled);
            ++*ndraws;
        }
        sd[q] = pick;
    }
    bool distinct = true;
#pragma unroll
    for (int a = 1; a < drawN; a++)
#pragma unroll
        for (int b = 0; b < a; b++) distinct = distinct && (sd[a] != sd[b]);
    return distinct;
}

// crec[r] = rec[sel[r]]: the enabled points' records in rank order
__global__ void __launch_bounds__(256)
compact_records_kernel(const double *__restrict__ rec, const int32_t *__restrict__ sel, int64_t n_enabled,
                       double *__restrict__ crec)
{
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_enabled) return;
    const f64x2 *s = (const f64x2 *)(rec + 8 * (int64_t)sel[r]);
    f64x2 *d = (f64x2 *)(crec + 8 * r);
    const f64x2 a = s[0], b = s[1], c = s[2];
    d[0] = a; d[1] = b; d[2] = c;
}

// Two kernels.  sample_sets_kernel is latency-bound (dependent random reads of enabled words, then
// the point gathers) and needs few registers, so it runs at full occupancy; it hands the gathered
// sets over in a coalesced [component][set] workspace.  fit_sets_kernel is the arithmetic (up to
// ~400 VGPRs with the cone fit's two 3x3 SVDs inlined) and reads that workspace with unit stride.
template <int DN>
__global__ void __launch_bounds__(256)
sample_sets_kernel(const double *__restrict__ rec, int64_t n, DevEnabled en, int32_t n_enabled,
                   const rhfit::OctView oc, const double *__restrict__ Pwin, int32_t drawN_rt, int32_t minsubsetN,
                   uint64_t seed, int64_t k0, int32_t n_iters, double *__restrict__ ws, int32_t *__restrict__ set_level,
                   unsigned long long *__restrict__ draws_per_iter, int32_t *__restrict__ gave_up_flag,
                   const double *__restrict__ crec, SetShard sh, const rh_oct_state *__restrict__ ost)
{
    if (ost != nullptr) {   // an iteration of a chained octree window (n_iters = 1; draws_per_iter already points at its counter)
        if (ost->stop != 0) return;
        Pwin = ost->P;
    }
    // thread t = (iteration, local set): this rank's sets of an iteration are j = lo + jl * step (one process: 0, 1)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n_iters * sh.m_local;
    if (t >= total) return;
    const int32_t it = (int32_t)(t / sh.m_local);
    const int32_t j = sh.lo + (int32_t)(t - (int64_t)it * sh.m_local) * sh.step;
    uint64_t x = rhfit::set_stream_init(seed, (uint64_t)(k0 + it), (uint64_t)j);
    constexpr int CAP = DN > 0 ? DN : RH_MAX_DRAWN;
    int64_t sd[CAP];
    uint32_t nd = 0;
    bool gave_up = false;
    const int drawN = DN > 0 ? DN : drawN_rt;
    int level = 1;
    // crec != null (root-cell sampling, long windows): sd[] holds RANKS among the enabled points and the
    // points come from the rank-ordered compact records -- one 64-byte line per point instead of a select-list
    // line plus a record line.  rank <-> index is a bijection on the enabled points, so "same point" and
    // "all different" mean the same on ranks (fitting.jl:416-428).
    const bool ok = Pwin != nullptr
                        ? rhfit::sample_minimal_set_octree<DN>(en, oc, Pwin + (int64_t)it * oc.depth, n, (int64_t)n_enabled,
                                                               drawN, &x, sd, &nd, &gave_up, &level)
                        : (crec != nullptr ? sample_ranks<DN>(en, n, (int64_t)n_enabled, drawN, &x, sd, &nd, &gave_up)
                                           : rhfit::sample_minimal_set<DN>(en, n, (int64_t)n_enabled, drawN, &x, sd, &nd, &gave_up));
    const double *src = crec != nullptr ? crec : rec;
    // draws per iteration: one atomic per (wave, iteration) instead of one per set -- thousands of
    // same-address atomics serialise in L2 and dominated the kernel
    {
        uint64_t todo = __builtin_amdgcn_ballot_w64(true);
        const int lane = threadIdx.x & 63;
        while (todo != 0) {
            const int leader = __builtin_ctzll(todo);
            const int32_t it0 = __shfl(it, leader);
            const uint64_t grp = __builtin_amdgcn_ballot_w64(it == it0) & todo;
            unsigned v = (it == it0) ? nd : 0u;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == leader) atomicAdd(&draws_per_iter[it0], (unsigned long long)v);
            todo &= ~grp;
        }
    }
    if (gave_up) atomicExch(gave_up_flag, 1);
    set_level[t] = ok ? level : 0;
#ifdef RH_OCT_TIMING
    unsigned long long t0_ = wall_clock64();
    if (Pwin != nullptr) atomicAdd(&rhfit::rh_oct_t[8], 1ULL);
#endif
    if (!ok) return;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < drawN; q++) {
        const f64x2 *r = (const f64x2 *)(src + 8 * (sd[q] - 1));   // one 64-byte record per point
        const f64x2 a = r[0], b = r[1], c = r[2];
        double *w = ws + (int64_t)(6 * q) * total + t;
        w[0] = a.x; w[total] = a.y; w[2 * total] = b.x;
        w[3 * total] = b.y; w[4 * total] = c.x; w[5 * total] = c.y;
    }
#ifdef RH_OCT_TIMING
    if (Pwin != nullptr) { RH_OCT_T(5); }
#endif
}



// fit(T, p, n, pc, params) for one shape type on a gathered minimal set (forcefitshapes!, fitting.jl:165-173).  f32: the cloud
// is a Float32 cloud -- all four fits in binary32 (fit_shared.h).
template <bool CONE>
static __device__ __forceinline__ bool fit_kind(int kind, const double *fp, const double *fn, int drawN, const rh_params &prm, int f32, rh_shape *s)
{
    switch (kind) {
    case RH_PLANE: return f32 ? rhfit::fit_plane32(fp, fn, drawN, prm, s) : rhfit::fit_plane(fp, fn, drawN, prm, s);
    case RH_SPHERE: return f32 ? rhfit::fit_sphere32(fp, fn, drawN, prm, s) : rhfit::fit_sphere(fp, fn, drawN, prm, s);
    case RH_CYLINDER: return f32 ? rhfit::fit_cylinder32(fp, fn, drawN, prm, s) : rhfit::fit_cylinder(fp, fn, drawN, prm, s);
    case RH_CONE: return CONE ? (f32 ? rhfit::fit_cone32(fp, fn, drawN, prm, s) : rhfit::fit_cone(fp, fn, drawN, prm, s)) : false;
    default: return false;
    }
}

template <int DN, bool CONE>
__global__ void __launch_bounds__(128)
fit_sets_kernel(const double *__restrict__ ws, const int32_t *__restrict__ set_level, int64_t total, const rh_params prm,
                rh_cand_entry *__restrict__ out, int32_t cap, int32_t *__restrict__ out_count, int32_t *__restrict__ nk_zero,
                SetShard sh, int32_t it0, const rh_oct_state *__restrict__ ost, int f32)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 4 && nk_zero != nullptr) nk_zero[t] = 0;   // the kind bins prep_entries_kernel fills next
    if (ost != nullptr && ost->stop != 0) return;
    if (t >= total) return;
    const int level = set_level[t];
    if (level == 0) return;
    constexpr int CAP = DN > 0 ? DN : RH_MAX_DRAWN;
    const int drawN = DN > 0 ? DN : prm.drawN;
    double fp[3 * CAP], fn[3 * CAP];
#pragma unroll
    for (int q = 0; q < drawN; q++) {
        const double *w = ws + (int64_t)(6 * q) * total + t;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            fp[3 * q + k] = w[k * total];
            fn[3 * q + k] = w[(3 + k) * total];
        }
    }
    for (int ti = 0; ti < prm.n_shape_types; ti++) {
        rh_shape s;
        for (int q = 0; q < 10; q++) s.v[q] = 0.0;
        s.kind = -1;
        s.outwards = 0;
        const bool fitted = fit_kind<CONE>(prm.shape_types[ti], fp, fn, drawN, prm, f32, &s);
        if (!fitted) continue;
        const int32_t pos = atomicAdd(out_count, 1);
        if (pos < cap) {
            // the slot is GLOBAL (iteration, set, type): the same whatever the number of processes sharing the window
            const int64_t it = t / sh.m_local;
            const int64_t tg = (it0 + it) * prm.minsubsetN + sh.lo + (t - it * sh.m_local) * sh.step;
            out[pos].slot = tg * prm.n_shape_types + ti;
            out[pos].level = level;
            out[pos].pad = 0;
            out[pos].shape = s;
        }
    }
}

// Octree sampling without cones: sample_sets_kernel and fit_sets_kernel in ONE launch (an iteration of a chained window is a
// chain of dependent launches -- sample, fit, prepare, score, advance -- and every launch less is ~5 us of its ~180).  A
// thread samples its set exactly like sample_sets_kernel (same stream, same draws), gathers the points into registers and
// fits every shape type like fit_sets_kernel; no hand-over workspace.  Blocks of one wave, like the octree sampler's.
template <int DN>
__global__ void __launch_bounds__(64)
sample_fit_oct_kernel(const double *__restrict__ rec, int64_t n, DevEnabled en, int32_t n_enabled, const rhfit::OctView oc,
                      const double *__restrict__ Pwin, const rh_params prm, uint64_t seed, int64_t k0, int32_t n_iters,
                      unsigned long long *__restrict__ draws_per_iter, int32_t *__restrict__ gave_up_flag, SetShard sh,
                      const rh_oct_state *__restrict__ ost, rh_cand_entry *__restrict__ out, int32_t cap, int32_t *__restrict__ out_count,
                      int32_t *__restrict__ nk_zero, int32_t it0, int f32)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 4 && nk_zero != nullptr) nk_zero[t] = 0;   // the kind bins prep_entries_kernel fills next
    if (ost != nullptr) {   // an iteration of a chained octree window (n_iters = 1; draws_per_iter already points at its counter)
        if (ost->stop != 0) return;
        Pwin = ost->P;
    }
    const int64_t total = (int64_t)n_iters * sh.m_local;
    if (t >= total) return;
    const int32_t it = (int32_t)(t / sh.m_local);
    const int32_t j = sh.lo + (int32_t)(t - (int64_t)it * sh.m_local) * sh.step;
    uint64_t x = rhfit::set_stream_init(seed, (uint64_t)(k0 + it), (uint64_t)j);
    constexpr int CAP = DN > 0 ? DN : RH_MAX_DRAWN;
    int64_t sd[CAP];
    uint32_t nd = 0;
    bool gave_up = false;
    const int drawN = DN > 0 ? DN : prm.drawN;
    int level = 1;
    const bool ok = rhfit::sample_minimal_set_octree<DN>(en, oc, Pwin + (int64_t)it * oc.depth, n, (int64_t)n_enabled, drawN, &x, sd, &nd,
                                                         &gave_up, &level);
    {   // draws per iteration: one atomic per (wave, iteration)
        uint64_t todo = __builtin_amdgcn_ballot_w64(true);
        const int lane = threadIdx.x & 63;
        while (todo != 0) {
            const int leader = __builtin_ctzll(todo);
            const int32_t itl = __shfl(it, leader);
            const uint64_t grp = __builtin_amdgcn_ballot_w64(it == itl) & todo;
            unsigned v = (it == itl) ? nd : 0u;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == leader) atomicAdd(&draws_per_iter[itl], (unsigned long long)v);
            todo &= ~grp;
        }
    }
    if (gave_up) atomicExch(gave_up_flag, 1);
    if (!ok) return;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    double fp[3 * CAP], fn[3 * CAP];
#pragma unroll
    for (int q = 0; q < drawN; q++) {
        const f64x2 *r = (const f64x2 *)(rec + 8 * (sd[q] - 1));   // one 64-byte record per point
        const f64x2 a = r[0], b = r[1], c = r[2];
        fp[3 * q] = a.x; fp[3 * q + 1] = a.y; fp[3 * q + 2] = b.x;
        fn[3 * q] = b.y; fn[3 * q + 1] = c.x; fn[3 * q + 2] = c.y;
    }
    for (int ti = 0; ti < prm.n_shape_types; ti++) {
        rh_shape s;
        for (int q = 0; q < 10; q++) s.v[q] = 0.0;
        s.kind = -1;
        s.outwards = 0;
        const bool fitted = fit_kind<false>(prm.shape_types[ti], fp, fn, drawN, prm, f32, &s);
        if (!fitted) continue;
        const int32_t pos = atomicAdd(out_count, 1);
        if (pos < cap) {
            const int64_t tg = (int64_t)(it0 + it) * prm.minsubsetN + j;   // the slot is GLOBAL (iteration, set, type)
            out[pos].slot = tg * prm.n_shape_types + ti;
            out[pos].level = level;
            out[pos].pad = 0;
            out[pos].shape = s;
        }
    }
}

// End of a window's chain: one block copies the status block, the head of the candidate list and the
// head of its counts straight into pinned host memory (three copy-engine transfers cost ~15 us per
// window), then zeroes the status block for the buffer's next window.
__global__ void __launch_bounds__(1024)
pack_window_kernel(uint64_t *__restrict__ status, int32_t status_words, const rh_cand_entry *__restrict__ entries,
                   const int32_t *__restrict__ counts, int32_t head_cap, uint64_t *__restrict__ h_status,
                   uint64_t *__restrict__ h_entries, int32_t *__restrict__ h_counts)
{
    const int tid = threadIdx.x;
    const int32_t cnt = min(((const int32_t *)status)[0], head_cap);
    for (int i = tid; i < status_words; i += 1024) h_status[i] = status[i];
    constexpr int EW = (int)(sizeof(rh_cand_entry) / 8);
    const uint64_t *src = (const uint64_t *)entries;
    for (int i = tid; i < cnt * EW; i += 1024) h_entries[i] = src[i];
    if (counts != nullptr)
        for (int i = tid; i < cnt; i += 1024) h_counts[i] = counts[i];
    __syncthreads();
    for (int i = tid; i < status_words; i += 1024) status[i] = 0;
}

// Root-cell windows without cones, sampled in rank space: ONE kernel.  With the compact records every point
// is a single 64-byte line from a set that fits the memory-side cache, so the hand-over workspace of the
// two-kernel form (144 B written + read per set) costs more than the lower occupancy of a kernel that also
// carries the plane / sphere / cylinder fits (~160 VGPRs, 3 waves per SIMD).
template <int DN>
__global__ void __launch_bounds__(128)
sample_fit_ranks_kernel(const double *__restrict__ crec, const double *__restrict__ rec, int64_t n, DevEnabled en,
                        int32_t n_enabled, const rh_params prm,
                        uint64_t seed, int64_t k0, int32_t n_iters, rh_cand_entry *__restrict__ out, int32_t cap,
                        int32_t *__restrict__ out_count, unsigned long long *__restrict__ draws_per_iter,
                        int32_t *__restrict__ gave_up_flag, int32_t *__restrict__ nk_zero, SetShard sh, int f32)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 4 && nk_zero != nullptr) nk_zero[t] = 0;
    const int64_t total = (int64_t)n_iters * sh.m_local;
    if (t >= total) return;
    const int32_t it = (int32_t)(t / sh.m_local);
    const int32_t j = sh.lo + (int32_t)(t - (int64_t)it * sh.m_local) * sh.step;
    uint64_t x = rhfit::set_stream_init(seed, (uint64_t)(k0 + it), (uint64_t)j);
    constexpr int CAP = DN > 0 ? DN : RH_MAX_DRAWN;
    int64_t sd[CAP];
    uint32_t nd = 0;
    bool gave_up = false;
    const int drawN = DN > 0 ? DN : prm.drawN;
    int64_t first_index = 0;
    const bool ok = sample_ranks<DN>(en, n, (int64_t)n_enabled, drawN, &x, sd, &nd, &gave_up, &first_index);
    {
        uint64_t todo = __builtin_amdgcn_ballot_w64(true);
        const int lane = threadIdx.x & 63;
        while (todo != 0) {
            const int leader = __builtin_ctzll(todo);
            const int32_t it0 = __shfl(it, leader);
            const uint64_t grp = __builtin_amdgcn_ballot_w64(it == it0) & todo;
            unsigned v = (it == it0) ? nd : 0u;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == leader) atomicAdd(&draws_per_iter[it0], (unsigned long long)v);
            todo &= ~grp;
        }
    }
    if (gave_up) atomicExch(gave_up_flag, 1);
    if (!ok) return;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    double fp[3 * CAP], fn[3 * CAP];
    // without the compact records (short windows) the ranks go back to indices through the select directory
    if (crec == nullptr) {
        sd[0] = first_index;
#pragma unroll
        for (int q = 1; q < drawN; q++) sd[q] = en.select(sd[q]);
    }
    const double *__restrict__ src = crec != nullptr ? crec : rec;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const f64x2 *r = (const f64x2 *)(src + 8 * (sd[q] - 1));
        const f64x2 a = r[0], b = r[1], c = r[2];
        fp[3 * q] = a.x; fp[3 * q + 1] = a.y; fp[3 * q + 2] = b.x;
        fn[3 * q] = b.y; fn[3 * q + 1] = c.x; fn[3 * q + 2] = c.y;
    }
    // Can this set still give a candidate?  Decided on its first two points, before the others are fetched
    // (a third of the random reads of a window of outliers):
    //  - sphere / cylinder are built from points 1 and 2 and then checked against every point with the
    //    conditions and-ed together, so fit(..., 2 points) = false implies fit(..., all points) = false --
    //    the same code, the same bits;
    //  - a plane needs every normal within alpha of +u or every one within alpha of -u for the unit normal u,
    //    hence angle(n1, n2) <= 2 alpha: with t = cos alpha > 0 it cannot pass when
    //    dot(n1^, n2^) < 2 t^2 - 1 (1e-9 below, far beyond the rounding of the dots).
    if (drawN > 2) {
        bool maybe = false;
        rh_shape tmp;
        for (int ti = 0; ti < prm.n_shape_types; ti++) {
            switch (prm.shape_types[ti]) {
            case RH_PLANE: {
                // (Float32 cloud: the fit's dots are binary32, a few 1e-7 from these: 1e-5 below)
                const double tpl = prm.cos_alpha[RH_PLANE];
                const double d12 = rhfit::dot(rhfit::normalize(rhfit::Vec(fn)), rhfit::normalize(rhfit::Vec(fn + 3)));
                maybe = maybe || !(tpl > 0.0 && d12 < 2.0 * tpl * tpl - 1.0 - (f32 ? 1e-5 : 1e-9));
                break;
            }
            case RH_SPHERE: maybe = maybe || fit_kind<false>(RH_SPHERE, fp, fn, 2, prm, f32, &tmp); break;
            case RH_CYLINDER: maybe = maybe || fit_kind<false>(RH_CYLINDER, fp, fn, 2, prm, f32, &tmp); break;
            default: maybe = true; break;
            }
        }
        if (!maybe) return;
    }
#pragma unroll
    for (int q = 2; q < drawN; q++) {
        const f64x2 *r = (const f64x2 *)(src + 8 * (sd[q] - 1));
        const f64x2 a = r[0], b = r[1], c = r[2];
        fp[3 * q] = a.x; fp[3 * q + 1] = a.y; fp[3 * q + 2] = b.x;
        fn[3 * q] = b.y; fn[3 * q + 1] = c.x; fn[3 * q + 2] = c.y;
    }
    for (int ti = 0; ti < prm.n_shape_types; ti++) {
        rh_shape s;
        for (int q = 0; q < 10; q++) s.v[q] = 0.0;
        s.kind = -1;
        s.outwards = 0;
        const bool fitted = fit_kind<false>(prm.shape_types[ti], fp, fn, drawN, prm, f32, &s);   // (cones take the two-kernel path)
        if (!fitted) continue;
        const int32_t pos = atomicAdd(out_count, 1);
        if (pos < cap) {
            out[pos].slot = ((int64_t)it * prm.minsubsetN + j) * prm.n_shape_types + ti;   // global (iteration, set, type)
            out[pos].level = 1;
            out[pos].pad = 0;
            out[pos].shape = s;
        }
    }
}


// ---- chained octree windows: the end of an iteration on the device (rh_internal.h, rh_oct_state) -------------------------
// E of estimatescore (confidenceintervals.jl:71-74, 53-59): the operations of rh_estimatescore (fit.cpp), one for one;
// (lo + hi) / 2 = (a + b) / 2 whichever of the two is the smaller.
__device__ __forceinline__ double oct_estimate_E(int64_t S1length, int64_t Plength, int64_t sigma, int32_t score_mode)
{
    const int64_t N = -2 - S1length, x = -2 - Plength, n = -1 - sigma;
    double sq_, xn;
    if (score_mode == RH_SCORE_INT64_WRAP) {
        const uint64_t xn_u = (uint64_t)x * (uint64_t)n;
        const uint64_t prod = xn_u * (uint64_t)(N - x) * (uint64_t)(N - n);
        sq_ = (double)(int64_t)prod / (double)(N - 1);
        xn = (double)(int64_t)xn_u;
    } else {
        const double xd = (double)x, nd = (double)n, Nd = (double)N;
        sq_ = (xd * nd * (Nd - xd) * (Nd - nd)) / (Nd - 1);
        xn = xd * nd;
    }
    const double sq = sq_ < 0 ? 0.0 : sqrt(sq_);
    const double a = -1 - (xn + sq) / (double)N, b = -1 - (xn - sq) / (double)N;
    return (a + b) / 2;
}

// One block.  The candidates of the iteration sit in the list at [ost->start, count) in the order the fits were
// appended (atomics); the reference adds their scores to the level scores in CANDIDATE order, and a sum of doubles
// depends on its order.  So: every candidate sets its bit in a zeroed (level, slot) bitmap and drops its list position
// into the table cell of the same key; an ordered compaction of the bitmap (a popcount scan over the block) yields the
// scores sorted by (level, slot); lane l of the first wave then adds level l's run to S[l] one by one, in order -- od
// independent serial chains.  A second bitmap over the slots alone gives every candidate its rank in candidate order
// (the host replays the iteration in that order without sorting).  The kernel also ships the iteration to the host
// (its slice of the list and of the counts, pinned memory) and appends the prepared records of its candidates to the
// device store (driver.hip), telling the host the store slot of each.  Thread 0 finishes the iteration:
// updatelevelweight, the counters, the extraction test.
constexpr int OA_THREADS = 1024, OA_WAVES = OA_THREADS / 64, OA_ES = 6144;

struct OctAdvArgs {
    rh_oct_state *ost;
    const rh_cand_entry *entries;
    const unsigned long long *status;    // int32 count, int32 gave_up, u64 draws[]
    const int32_t *counts;
    int32_t cap, it, od, score_mode, extract_s, minsubsetN, drawN;
    int64_t k, per_it, S1length, Plength;
    double prob_det;
    double *Etab;                        // [od * per_it] score per (level, slot); stale outside the set bits
    unsigned long long *bits, *sbits;    // (level, slot) bitmap, slot bitmap: zero on entry and on exit
    int32_t *spref;                      // [words of sbits] scratch
    double *Esort;                       // [per_it] scratch
    const rh_prep *bin_prep;             // the iteration's candidates as prep_entries_kernel binned them: records,
    const int32_t *bin_orig, *bin_nk;    //   list positions, numbers per kind; bins bin_cap apart
    int64_t bin_cap;
    rh_cand_entry *h_entries;            // pinned: the list, its counts, every entry's rank in candidate order within its
    int32_t *h_counts, *h_rank, *h_slot; //   iteration and its slot in the device store of its kind
    rh_oct_iter_hdr *h_hdr;
};

// copies n words, the block's threads t0 .. t0 + nt - 1 taking eight each per round (eight loads in flight)
typedef unsigned int oa_u32x4 __attribute__((ext_vector_type(4)));
template <class W>
static __device__ __forceinline__ void oa_copy(W *__restrict__ dst, const W *__restrict__ src, int64_t n, int t, int nt)
{
    for (int64_t i0 = (int64_t)t; i0 < n; i0 += (int64_t)nt * 8) {
        W v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int64_t i = i0 + (int64_t)q * nt; if (i < n) v[q] = src[i]; }
#pragma unroll
        for (int q = 0; q < 8; q++) { const int64_t i = i0 + (int64_t)q * nt; if (i < n) dst[i] = v[q]; }
    }
}

__global__ void __launch_bounds__(OA_THREADS)
oct_advance_kernel(const OctAdvArgs A)
{
    rh_oct_state *__restrict__ ost = A.ost;
    const int it = A.it;
    if (ost->stop != 0) {
        if (threadIdx.x == 0) A.h_hdr[it].skipped = 1;
        return;
    }
    __shared__ int32_t hist[33], seg[33], wave_tot[2][OA_WAVES], wave_base[2][OA_WAVES];
    __shared__ double bestw[OA_WAVES], lP[32], lS[32], lr[32], lw;
    __shared__ int32_t anyw[OA_WAVES], store_over, lstop, lkeep;
    __shared__ double Es[OA_ES];          // the sorted scores of an iteration of up to OA_ES candidates (else: A.Esort)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const rh_cand_entry *__restrict__ entries = A.entries;
    const int64_t per_it = A.per_it;
    const int od = A.od;
    const int32_t count = (int32_t)(A.status[0] & 0xffffffffULL);
    const int32_t start = ost->start, end = max(start, min(count, A.cap)), m = end - start;
    int32_t nkq[4], sbase[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { nkq[q] = A.bin_nk[q]; sbase[q] = ost->store_n[q]; }
    if (tid < 33) hist[tid] = 0;
    if (tid < 32) { lP[tid] = ost->P[tid]; lS[tid] = ost->S[tid]; }
    if (tid == 0) { store_over = 0; lstop = 0; lkeep = 0; }
#ifdef RH_OCT_TIMING
    unsigned long long tq[8];
    int tqi = 0;
#define OA_T() do { if (tid == 0) tq[tqi] = wall_clock64(); tqi++; } while (0)
#else
#define OA_T() do { } while (0)
#endif
    OA_T();
    __syncthreads();
    // 1. every candidate: its score into the (level, slot) table, its bits, the level histogram, the best score
    double bE = 0.0;
    bool any = false;
    for (int32_t e = start + tid; e < end; e += OA_THREADS) {
        const int64_t ls = entries[e].slot - (int64_t)it * per_it;
        const int32_t lv = entries[e].level;
        if (ls >= 0 && ls < per_it && lv >= 1 && lv <= od) {
            const double E = oct_estimate_E(A.S1length, A.Plength, (int64_t)A.counts[e], A.score_mode);
            const int64_t key = (int64_t)(lv - 1) * per_it + ls;
            A.Etab[key] = E;
            atomicOr(&A.bits[key >> 6], 1ULL << (key & 63));
            atomicOr(&A.sbits[ls >> 6], 1ULL << (ls & 63));
            atomicAdd(&hist[lv - 1], 1);
            if (!any || E > bE) bE = E;
            any = true;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {   // (a maximum: any order)
        const double o = __shfl_xor(bE, off);
        const int oa = __shfl_xor((int)any, off);
        if (oa && (!any || o > bE)) bE = o;
        any = any || oa;
    }
    if (lane == 0) { bestw[wv] = bE; anyw[wv] = any ? 1 : 0; }
    __syncthreads();
    OA_T();
    if (tid == 0) {
        int32_t a = 0;
        for (int l = 0; l < od; l++) { seg[l] = a; a += hist[l]; }
        seg[od] = a;
    }
    // 2. ordered compaction: thread t owns the words [t * wpt, (t + 1) * wpt) of either bitmap
    const int64_t nw = ((int64_t)od * per_it + 63) >> 6, nws = (per_it + 63) >> 6;
    const int64_t wpt = (nw + OA_THREADS - 1) / OA_THREADS, wpts = (nws + OA_THREADS - 1) / OA_THREADS;
    const int64_t w0 = min(nw, (int64_t)tid * wpt), w1 = min(nw, w0 + wpt);
    const int64_t v0 = min(nws, (int64_t)tid * wpts), v1 = min(nws, v0 + wpts);
    int32_t mine = 0, mines = 0;
    for (int64_t w = w0; w < w1; w++) mine += __popcll(A.bits[w]);
    for (int64_t w = v0; w < v1; w++) mines += __popcll(A.sbits[w]);
    int32_t incl = mine, incls = mines;   // inclusive scans over the wave, then over the waves
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t o = __shfl_up(incl, off), os = __shfl_up(incls, off);
        if (lane >= off) { incl += o; incls += os; }
    }
    if (lane == 63) { wave_tot[0][wv] = incl; wave_tot[1][wv] = incls; }
    __syncthreads();
    if (tid < 2) {
        int32_t a = 0;
        for (int w = 0; w < OA_WAVES; w++) { wave_base[tid][w] = a; a += wave_tot[tid][w]; }
    }
    __syncthreads();
    OA_T();
    {
        int32_t ps = wave_base[1][wv] + incls - mines;
        for (int64_t w = v0; w < v1; w++) { A.spref[w] = ps; ps += __popcll(A.sbits[w]); }
    }
    const bool lds = m <= OA_ES;
    int32_t pos = wave_base[0][wv] + incl - mine;
    for (int64_t w = w0; w < w1; w++) {
        unsigned long long bm = A.bits[w];
        if (bm == 0) continue;
        A.bits[w] = 0;                           // the bitmap is zero again for the next iteration
        while (bm != 0) {
            const int bit = __builtin_ctzll(bm);
            bm &= bm - 1;
            const double E = A.Etab[(w << 6) + bit];
            if (lds) Es[pos++] = E; else A.Esort[pos++] = E;
        }
    }
    __syncthreads();
    OA_T();
    // 3. first wave: the level scores in candidate order, then updatelevelweight; second wave: the extraction test;
    // the others: the iteration -> the host, its prepared records -> the device store
    if (wv == 0) {
        if (lane < od) {
            // (the chain of additions is the critical path: the next eight scores are fetched while the current eight are added)
            double acc = lS[lane];
            const int32_t a = seg[lane], b = seg[lane + 1];
            auto at = [&](int32_t j) { return lds ? Es[j] : A.Esort[j]; };
            int32_t j = a;
            if (j + 8 <= b) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = at(j + q);
                for (; j + 16 <= b; j += 8) {
                    double nx[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) nx[q] = at(j + 8 + q);
#pragma unroll
                    for (int q = 0; q < 8; q++) acc += v[q];
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = nx[q];
                }
#pragma unroll
                for (int q = 0; q < 8; q++) acc += v[q];
                j += 8;
            }
            for (; j < b; j++) acc += at(j);
            lS[lane] = acc;
            ost->S[lane] = acc;
        }
        // updatelevelweight (octree.jl:198-205): the operations of update_level_probs (fit_shared.h) -- the quotients
        // and the new weights lane-parallel, the two sums one term after the other in level order
        if (od <= 32) {
            if (lane < od) lr[lane] = lS[lane] / lP[lane];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                double w = 0;
                for (int i = 0; i < od; i++) w += lr[i];
                lw = w;
            }
            __builtin_amdgcn_wave_barrier();
            const double w = lw;
            double pn = 0.0;
            if (lane < od) { pn = 0.9 * lS[lane] / (w * lP[lane]) + (1 - 0.9) / od; lr[lane] = pn; }
            const bool bad = lane < od && !(pn >= 0);
            const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                double sum = 0;
                for (int i = 0; i < od; i++) sum += lr[i];
                lkeep = ((w > 0) && !any_bad && (sum > 0)) ? 1 : 0;
            }
            __builtin_amdgcn_wave_barrier();
            if (lkeep && lane < od) lP[lane] = pn;
        }
    } else if (wv == 1) {
        if (lane == 0) {
            double best = ost->best_E;
            int32_t has = ost->has_best;
            for (int w = 0; w < OA_WAVES; w++)
                if (anyw[w] && (!has || bestw[w] > best)) { best = bestw[w]; has = 1; }
            ost->best_E = best;
            ost->has_best = has;
            const int32_t scored = seg[od];           // (= m unless an entry was malformed)
            const long long store_n = ost->store_count + scored, cc2 = ost->cc2 + scored;
            ost->store_count = store_n;
            ost->cc2 = cc2;
            // the extraction test of iterations.jl:114-123 -- with the device's pow, which may differ from the host's in
            // the last place: this only ends the window (the host replays the iterations and decides)
            if (has) {
                const long long sl[4] = { 0, store_n, cc2, (long long)A.k * A.minsubsetN };
                const double ppp = 1 - pow(1 - pow(best / (double)A.Plength, (double)A.drawN), (double)sl[A.extract_s & 3]);
                if (ppp > A.prob_det) lstop = 1;
            }
        }
    } else {
        const int t = tid - 128, nt = OA_THREADS - 128;
        static_assert(sizeof(rh_cand_entry) % 8 == 0 && sizeof(rh_prep) % 16 == 0, "copied as 8- / 16-byte words");
        oa_copy((unsigned long long *)(A.h_entries + start), (const unsigned long long *)(entries + start),
                (int64_t)m * (int64_t)(sizeof(rh_cand_entry) / 8), t, nt);
        for (int32_t i = start + t; i < end; i += nt) {
            A.h_counts[i] = A.counts[i];
            const int64_t ls = entries[i].slot - (int64_t)it * per_it;
            int32_t r = -1;
            if (ls >= 0 && ls < per_it) r = A.spref[ls >> 6] + __popcll(A.sbits[ls >> 6] & ((1ULL << (ls & 63)) - 1ULL));
            A.h_rank[i] = r;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (nkq[q] == 0) continue;
            if ((int64_t)sbase[q] + nkq[q] > ost->store_cap[q]) { if (t == 0) store_over = 1; continue; }
            oa_copy((oa_u32x4 *)(ost->store_prep[q] + sbase[q]), (const oa_u32x4 *)(A.bin_prep + (int64_t)q * A.bin_cap),
                      (int64_t)nkq[q] * (int64_t)(sizeof(rh_prep) / 16), t, nt);
            for (int32_t j = t; j < nkq[q]; j += nt) {
                const int32_t e = A.bin_orig[(int64_t)q * A.bin_cap + j];
                int32_t id = -1;
                if (e >= start && e < end) {
                    A.h_slot[e] = sbase[q] + j;
                    const int64_t ls = entries[e].slot - (int64_t)it * per_it;
                    if (ls >= 0 && ls < per_it)
                        id = (int32_t)ost->appended + A.spref[ls >> 6] + __popcll(A.sbits[ls >> 6] & ((1ULL << (ls & 63)) - 1ULL));
                }
                ost->store_id[q][sbase[q] + j] = id;
                ost->store_E[q][sbase[q] + j] = (e >= start && e < end) ? oct_estimate_E(A.S1length, A.Plength, (int64_t)A.counts[e], A.score_mode) : 0.0;
            }
        }
    }
    OA_T();
    __syncthreads();
    OA_T();
    for (int64_t w = v0; w < v1; w++) A.sbits[w] = 0;
    if (tid < 32) {
        ost->P[tid] = lP[tid];
        A.h_hdr[it].P[tid] = lP[tid];
    }
    if (tid == 0) {
        ost->start = end;
        ost->it_done = it + 1;
#pragma unroll
        for (int q = 0; q < 4; q++) ost->store_n[q] = sbase[q] + nkq[q];
        ost->appended += seg[od];
        const int32_t over = (count > A.cap ? 1 : 0) | (store_over != 0 ? 2 : 0);   // bit 0: the list is full, bit 1: the store
        const int32_t stop = (over || lstop) ? 1 : 0;
        if (stop) ost->stop = 1;
        rh_oct_iter_hdr &H = A.h_hdr[it];
        H.skipped = 0;
        H.overflow = over;
        H.gave_up = (int32_t)(A.status[0] >> 32);
        H.start = start;
        H.end = end;
        H.stop_after = stop;
        H.draws = A.status[1 + it];
#ifdef RH_OCT_TIMING
        tq[7] = wall_clock64();
        for (int i = 0; i < 7; i++) H.t[i] = tq[i + 1] - tq[i];
#endif
    }
}

}  // namespace

#ifdef RH_OCT_TIMING
extern "C" int rh_dbg_oct_timing(unsigned long long *out)
{
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rhfit::rh_oct_t), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

int rhk_pack_window(rh_cloud *c, void *d_status, int32_t n_iters, const rh_cand_entry *d_entries, const int32_t *d_counts,
                    int32_t head_cap, void *h_status, void *h_entries, int32_t *h_counts)
{
    static_assert(sizeof(rh_cand_entry) % 8 == 0, "entries are copied as 64-bit words");
    const int32_t words = (int32_t)((8 + 8 * (int64_t)n_iters + 63) / 64 * 8);
    hipLaunchKernelGGL(pack_window_kernel, dim3(1), dim3(1024), 0, c->stream, (uint64_t *)d_status, words, d_entries, d_counts,
                       head_cap, (uint64_t *)h_status, (uint64_t *)h_entries, h_counts);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_sample_fit(rh_cloud *c, const rh_params *prm, uint64_t seed, int64_t k0, int32_t n_iters, int32_t n_enabled,
                   const double *d_P, rh_cand_entry *d_out, int32_t cap, void *d_status, int status_is_zero,
                   int32_t *d_nk_zero, int32_t it0, const rh_oct_state *ost)
{
    if (ost != nullptr && (n_iters != 1 || d_P == nullptr)) { rh_set_error("rhk_sample_fit: a chained window samples one octree iteration per launch"); return RH_E_INTERNAL; }
    // status block: int32 count, int32 gave_up, u64 draws[n_iters]
    int32_t *d_count = (int32_t *)d_status, *d_gave_up = d_count + 1;
    unsigned long long *d_draws = (unsigned long long *)((char *)d_status + 8) + it0;
    if (prm->drawN > RH_MAX_DRAWN) { rh_set_error("device sampler supports drawN <= %d", RH_MAX_DRAWN); return RH_E_INVALID; }
    if (!c->select_valid) RH_TRY(rhk_build_select(c));
    // (the block is a multiple of 64 bytes: one aligned fill; the driver's windows keep it zero themselves)
    if (!status_is_zero) RH_HIP(hipMemsetAsync(d_status, 0, (size_t)((8 + 8 * (int64_t)n_iters + 63) / 64 * 64), c->stream));
    SetShard sh;
    sh.lo = c->mp_rank;
    sh.step = c->mp_world > 0 ? c->mp_world : 1;
    sh.m_local = prm->minsubsetN > sh.lo ? (prm->minsubsetN - sh.lo + sh.step - 1) / sh.step : 0;
    const int64_t total = (int64_t)n_iters * sh.m_local;
    if (total == 0) return RH_OK;
    // hand-over workspace (grown on demand; a window of 128 x 4096 sets of 3 points is 75 MB)
    const int64_t need = total * 6 * prm->drawN;
    if (c->set_ws_doubles < need) {
        (void)hipFree(c->set_ws);
        c->set_ws = nullptr; c->set_ws_doubles = 0;
        RH_HIP(hipMalloc((void **)&c->set_ws, sizeof(double) * (size_t)need));
        c->set_ws_doubles = need;
    }
    if (c->set_ws_sets < total) {
        (void)hipFree(c->set_level);
        c->set_level = nullptr; c->set_ws_sets = 0;
        RH_HIP(hipMalloc((void **)&c->set_level, sizeof(int32_t) * (size_t)total));
        c->set_ws_sets = total;
    }
    bool cone = false;
    for (int i = 0; i < prm->n_shape_types; i++) cone |= prm->shape_types[i] == RH_CONE;
    const bool no_fused = rh_opt_on(c, RH_OPT_NO_FUSED_SAMPLER);   // read per call: the tests flip it
    // Windows with many root-cell sets pay for two structures that are rebuilt after every extraction: the
    // flat select list (4 B per enabled point) and, from it, the rank-ordered compact records (64 B per
    // enabled point).  Short windows -- the ones between extractions -- search the directory instead.
    const bool no_crec = rh_opt_on(c, RH_OPT_NO_CREC);
    const int64_t long_sets = rh_opt_int(c, RH_OPT_LONG_WINDOW_SETS, (int64_t)1 << 16);
    const bool long_window = d_P == nullptr && n_enabled > 0 && total >= long_sets;
    const double *crec = nullptr;
    if (long_window) {
        RH_TRY(rhk_build_sel_list(c));
        // the compact records cost a gather of the whole enabled cloud (0.12-0.33 ms at 10M points, 0.8 ms at
        // 50M): only for windows 8 x longer still, and only from the second such window on the same enabled bits
        // -- one that follows an extraction directly is usually cut short by the next one
        if (total >= 8 * long_sets) c->very_long_windows++;
        if (!no_crec && (c->crec_valid || (total >= 8 * long_sets && c->very_long_windows >= 2))) {
            if (!c->crec_valid) {
                if (c->crec_cap < n_enabled) {
                    (void)hipFree(c->crec);
                    c->crec = nullptr; c->crec_cap = 0;
                    RH_HIP(hipMalloc((void **)&c->crec, sizeof(double) * 8 * (size_t)n_enabled));
                    c->crec_cap = n_enabled;
                }
                hipLaunchKernelGGL(compact_records_kernel, dim3((unsigned)(((int64_t)n_enabled + 255) / 256)), dim3(256), 0,
                                   c->stream, c->rec, c->sel_list, (int64_t)n_enabled, c->crec);
                c->crec_valid = true;
            }
            crec = c->crec;
        }
    }
    DevEnabled en;
    en.w = c->enabled;
    en.prefix = c->word_prefix;
    en.nwords = c->nwords;
    en.total = n_enabled;
    en.sel = c->sel_valid ? c->sel_list : nullptr;
    rhfit::OctView oc;
    oc.code = c->oct_code; oc.perm = c->oct_perm; oc.pos = c->oct_pos; oc.men = c->oct_men; oc.prefix = c->oct_prefix;
    oc.n = c->n; oc.nwords = c->nwords; oc.depth = c->oct_depth;
    if (!rh_opt_on(c, RH_OPT_NO_OCT_TAB)) { oc.tab = c->oct_tab; oc.tab_level = c->oct_tab_level; oc.code_o = c->oct_code_o; }
    // (octree sampling is a chain of dependent random reads per set: one wave per block spreads a window of a few thousand
    // sets over four times as many compute units -- each with its own address translation -- as 256-thread blocks would)
    const int sblock = d_P != nullptr ? 64 : 256;
    const dim3 gs((unsigned)((total + sblock - 1) / sblock)), gf((unsigned)((total + 127) / 128));
    if (d_P == nullptr && n_enabled > 0 && !cone && !no_fused) {   // rank-space sampling + fits in one kernel, no hand-over
        const dim3 gk((unsigned)((total + 127) / 128));
        if (prm->drawN == 3)
            hipLaunchKernelGGL(sample_fit_ranks_kernel<3>, gk, dim3(128), 0, c->stream, crec, c->rec, c->n, en, n_enabled, *prm,
                               seed, k0, n_iters, d_out, cap, d_count, d_draws, d_gave_up, d_nk_zero, sh, c->f32 ? 1 : 0);
        else
            hipLaunchKernelGGL(sample_fit_ranks_kernel<0>, gk, dim3(128), 0, c->stream, crec, c->rec, c->n, en, n_enabled, *prm,
                               seed, k0, n_iters, d_out, cap, d_count, d_draws, d_gave_up, d_nk_zero, sh, c->f32 ? 1 : 0);
        RH_HIP(hipGetLastError());
        return RH_OK;
    }
    if (d_P != nullptr && !cone && !no_fused) {   // octree sampling: sampler + fits in one launch (sample_fit_oct_kernel)
        const dim3 go((unsigned)((total + 63) / 64));
        if (prm->drawN == 3)
            hipLaunchKernelGGL(sample_fit_oct_kernel<3>, go, dim3(64), 0, c->stream, c->rec, c->n, en, n_enabled, oc, d_P, *prm, seed, k0,
                               n_iters, d_draws, d_gave_up, sh, ost, d_out, cap, d_count, d_nk_zero, it0, c->f32 ? 1 : 0);
        else
            hipLaunchKernelGGL(sample_fit_oct_kernel<0>, go, dim3(64), 0, c->stream, c->rec, c->n, en, n_enabled, oc, d_P, *prm, seed, k0,
                               n_iters, d_draws, d_gave_up, sh, ost, d_out, cap, d_count, d_nk_zero, it0, c->f32 ? 1 : 0);
        RH_HIP(hipGetLastError());
        return RH_OK;
    }
#define RH_SAMPLE(DN)                                                                                                  \
    hipLaunchKernelGGL(sample_sets_kernel<DN>, gs, dim3((unsigned)sblock), 0, c->stream, c->rec, c->n, en, n_enabled, oc, d_P,      \
                       prm->drawN, prm->minsubsetN, seed, k0, n_iters, c->set_ws, c->set_level, d_draws, d_gave_up, crec, sh, ost)
#define RH_FIT(DN, CONE)                                                                                               \
    hipLaunchKernelGGL((fit_sets_kernel<DN, CONE>), gf, dim3(128), 0, c->stream, c->set_ws, c->set_level, total, *prm, \
                       d_out, cap, d_count, d_nk_zero, sh, it0, ost, c->f32 ? 1 : 0)
    if (prm->drawN == 3) {   // the reference's default: fully unrolled, no scratch
        RH_SAMPLE(3);
        if (cone) RH_FIT(3, true); else RH_FIT(3, false);
    } else {
        RH_SAMPLE(0);
        if (cone) RH_FIT(0, true); else RH_FIT(0, false);
    }
#undef RH_SAMPLE
#undef RH_FIT
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_oct_advance(rh_cloud *c, const rh_params *prm, rh_oct_state *ost, const rh_cand_entry *d_entries, const void *d_status,
                    int32_t cap, const int32_t *d_counts, int32_t it, int64_t k, rh_cand_entry *h_entries, int32_t *h_counts,
                    int32_t *h_rank, int32_t *h_slot, rh_oct_iter_hdr *h_hdr)
{
    const int64_t per_it = (int64_t)prm->minsubsetN * prm->n_shape_types;
    const int od = c->oct_depth;
    if (od < 1 || od > 32) { rh_set_error("rhk_oct_advance: octree depth %d", od); return RH_E_INTERNAL; }
    const int64_t cells = (int64_t)od * per_it;
    const int64_t nw = (cells + 63) / 64, nws = (per_it + 63) / 64;
    if (c->oct_adv_cells < cells) {
        RH_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(c->oct_adv_tab); (void)hipFree(c->oct_adv_bits); (void)hipFree(c->oct_adv_E);
        c->oct_adv_tab = nullptr; c->oct_adv_bits = nullptr; c->oct_adv_E = nullptr; c->oct_adv_cells = 0;
        RH_HIP(hipMalloc((void **)&c->oct_adv_tab, sizeof(double) * (size_t)cells + sizeof(int32_t) * (size_t)nws));   // score table, then the slot prefixes
        RH_HIP(hipMalloc((void **)&c->oct_adv_bits, sizeof(unsigned long long) * (size_t)(nw + nws)));   // both bitmaps
        RH_HIP(hipMemsetAsync(c->oct_adv_bits, 0, sizeof(unsigned long long) * (size_t)(nw + nws), c->stream));   // the kernel leaves them zero
        RH_HIP(hipMalloc((void **)&c->oct_adv_E, sizeof(double) * (size_t)per_it));
        c->oct_adv_cells = cells;
    }
    OctAdvArgs A;
    A.ost = ost; A.entries = d_entries; A.status = (const unsigned long long *)d_status; A.counts = d_counts;
    A.cap = cap; A.it = it; A.od = od; A.score_mode = prm->score_mode; A.extract_s = prm->extract_s; A.minsubsetN = prm->minsubsetN;
    A.drawN = prm->drawN; A.k = k; A.per_it = per_it; A.S1length = c->s; A.Plength = c->n; A.prob_det = prm->prob_det;
    A.Etab = (double *)c->oct_adv_tab; A.spref = (int32_t *)((double *)c->oct_adv_tab + cells); A.bits = c->oct_adv_bits; A.sbits = c->oct_adv_bits + nw; A.Esort = c->oct_adv_E;
    A.bin_prep = c->d_prep; A.bin_orig = c->d_orig; A.bin_nk = c->d_nk; A.bin_cap = c->batch_cap;
    A.h_entries = h_entries; A.h_counts = h_counts; A.h_rank = h_rank; A.h_slot = h_slot; A.h_hdr = h_hdr;
    hipLaunchKernelGGL(oct_advance_kernel, dim3(1), dim3(OA_THREADS), 0, c->stream, A);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

namespace {
__global__ void oct_window_begin_kernel(rh_oct_state *ost)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { ost->start = 0; ost->it_done = 0; }
}
}  // namespace

// a chained octree window that continues from the device's state (no upload): its candidate list starts at position 0
int rhk_oct_window_begin(rh_cloud *c, rh_oct_state *ost)
{
    hipLaunchKernelGGL(oct_window_begin_kernel, dim3(1), dim3(64), 0, c->stream, ost);
    RH_HIP(hipGetLastError());
    return RH_OK;
}
