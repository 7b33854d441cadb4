// fit.cpp -- host-side pieces of the plugin API: minimal-set fits, score statistics,
// parameters, RNG.  O(1) per minimal set, so they stay on the host like in the reference.
// binary64, the reference's operation order, built with -ffp-contract=off.
// Paths in comments are under /root/reference/src.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "fit_shared.h"
#include "rh_internal.h"

namespace {

using namespace rhfit;

constexpr double kPi = 3.14159265358979323846;

inline double julia_min(double x, double y) { return x != x ? x : (y != y ? y : (y < x ? y : x)); }
inline double julia_max(double x, double y) { return x != x ? x : (y != y ? y : (x < y ? y : x)); }

uint64_t splitmix(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

}  // namespace

// ------------------------------------------------------------------- ABI ----
extern "C" void rh_params_finalize(rh_params *p)
{
    for (int k = 0; k < 4; k++) p->cos_alpha[k] = cos(p->alpha[k]);
    p->cos_parallelthr = cos(p->parallelthrdeg * kPi / 180.0);   // cosd: sphere.jl:37, cylinder.jl:40
}

// utilities.jl:345,371; plane.jl:22; sphere.jl:25; cylinder.jl:27; cone.jl:30; RANSAC.jl:94
extern "C" void rh_default_params(rh_params *p)
{
    memset(p, 0, sizeof *p);
    for (int k = 0; k < 4; k++) { p->eps[k] = 0.3; p->alpha[k] = 5.0 * kPi / 180.0; }
    p->collin_threshold = 0.2;
    p->parallelthrdeg = 1.0;
    p->sphere_par = 0.02;
    p->minconeopang = 2.0 * kPi / 180.0;
    p->prob_det = 0.9;
    p->tau = 900;
    p->itermax = 1000;
    p->drawN = 3;
    p->minsubsetN = 15;
    p->extract_s = RH_S_NOFMINSET;
    p->terminate_s = RH_S_NOFMINSET;
    p->n_shape_types = 4;
    p->shape_types[0] = RH_PLANE;
    p->shape_types[1] = RH_CONE;
    p->shape_types[2] = RH_CYLINDER;
    p->shape_types[3] = RH_SPHERE;
    p->score_mode = RH_SCORE_INT64_WRAP;
    p->sphere_uses_enabled = 0;
    p->octree_max_depth = 10;
    rh_params_finalize(p);
}

extern "C" void rh_shape_finalize(rh_shape *s)
{
    if (s->kind == RH_CONE) rhfit::cone_finalize(s);   // deterministic cos / sin (det_math.h)
}

extern "C" int rh_fit(int kind, const double *p, const double *n, int32_t lp, const rh_params *prm, rh_shape *out,
                      int32_t *fitted)
{
    if (!p || !n || !prm || !out || !fitted) { rh_set_error("rh_fit: NULL argument"); return RH_E_INVALID; }
    if (lp < 3) { rh_set_error("rh_fit: at least 3 points are needed (got %d)", lp); return RH_E_INVALID; }   // @assert
    rh_shape s;
    memset(&s, 0, sizeof s);
    bool ok;
    switch (kind) {
    case RH_PLANE: ok = fit_plane(p, n, lp, *prm, &s); break;
    case RH_SPHERE: ok = fit_sphere(p, n, lp, *prm, &s); break;
    case RH_CYLINDER: ok = fit_cylinder(p, n, lp, *prm, &s); break;
    case RH_CONE: ok = fit_cone(p, n, lp, *prm, &s); break;
    default: rh_set_error("rh_fit: unknown kind %d", kind); return RH_E_INVALID;
    }
    *fitted = ok ? 1 : 0;
    if (ok) *out = s;
    return RH_OK;
}

// fit on a Float32 cloud (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109): p / n hold the Float32 values (as
// doubles, exactly); the fit runs in binary32 (fit_shared.h) and its shape holds binary32 numbers
extern "C" int rh_fit_f32(int kind, const double *p, const double *n, int32_t lp, const rh_params *prm, rh_shape *out,
                          int32_t *fitted)
{
    if (!p || !n || !prm || !out || !fitted) { rh_set_error("rh_fit_f32: NULL argument"); return RH_E_INVALID; }
    if (lp < 3) { rh_set_error("rh_fit_f32: at least 3 points are needed (got %d)", lp); return RH_E_INVALID; }
    rh_shape s;
    memset(&s, 0, sizeof s);
    bool ok;
    switch (kind) {
    case RH_PLANE: ok = fit_plane32(p, n, lp, *prm, &s); break;
    case RH_SPHERE: ok = fit_sphere32(p, n, lp, *prm, &s); break;
    case RH_CYLINDER: ok = fit_cylinder32(p, n, lp, *prm, &s); break;
    case RH_CONE: ok = fit_cone32(p, n, lp, *prm, &s); break;
    default: rh_set_error("rh_fit_f32: unknown kind %d", kind); return RH_E_INVALID;
    }
    *fitted = ok ? 1 : 0;
    if (ok) *out = s;
    return RH_OK;
}

// estimatescore + hypergeomdev + notsoconfident: confidenceintervals.jl:71-74, 53-59, 20-22
extern "C" int rh_estimatescore(int64_t S1length, int64_t Plength, int64_t sigma, int32_t score_mode, double *ci_min,
                                double *ci_max, double *ci_E)
{
    const int64_t N = -2 - S1length, x = -2 - Plength, n = -1 - sigma;
    double sq_, xn;
    if (score_mode == RH_SCORE_INT64_WRAP) {
        // Julia Int64 arithmetic wraps silently; unsigned multiplication has the same bits
        const uint64_t xn_u = (uint64_t)x * (uint64_t)n;
        const uint64_t prod = xn_u * (uint64_t)(N - x) * (uint64_t)(N - n);
        sq_ = (double)(int64_t)prod / (double)(N - 1);
        xn = (double)(int64_t)xn_u;
    } else if (score_mode == RH_SCORE_F64) {
        const double xd = (double)x, nd = (double)n, Nd = (double)N;
        sq_ = (xd * nd * (Nd - xd) * (Nd - nd)) / (Nd - 1);
        xn = xd * nd;
    } else {
        rh_set_error("rh_estimatescore: unknown score_mode %d", score_mode);
        return RH_E_INVALID;
    }
    const double sq = sq_ < 0 ? 0.0 : sqrt(sq_);
    const double a = -1 - (xn + sq) / (double)N, b = -1 - (xn - sq) / (double)N;
    const double lo = julia_min(a, b), hi = julia_max(a, b);
    if (ci_min) *ci_min = lo;
    if (ci_max) *ci_max = hi;
    if (ci_E) *ci_E = (lo + hi) / 2;
    return RH_OK;
}

// prob(n, s, N, k) = 1-(1-(n/N)^k)^s: utilities.jl:262
extern "C" double rh_prob(double n, int64_t s, int64_t N, int64_t k)
{
    return 1 - pow(1 - pow(n / (double)N, (double)k), (double)s);
}

extern "C" void rh_rng_seed(rh_rng *r, uint64_t seed)
{
    memset(r, 0, sizeof *r);
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) r->s[i] = splitmix(&x);
}

static uint64_t rng_next(rh_rng *r)
{
    r->draws++;
    if (r->stream && r->stream_pos < r->stream_len) return r->stream[r->stream_pos++];
    uint64_t *s = r->s;
    const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

// forcefitshapes! (fitting.jl:165-173) for the minimal sets of an iteration in one call (the sets rh_sample_sets drew): fit for
// every type of p->shape_types in order on every set with ok != 0; what fits is appended with the index of its set.  The sets
// are independent: from 512 sets on they are dealt to host threads in contiguous ranges and the ranges' results joined in
// order, so the output does not depend on the number of threads.
namespace {
struct FitSetsOut { std::vector<rh_shape> shapes; std::vector<int32_t> sets; int rc = RH_OK; };
void fit_sets_range(const double *xyz, const double *nrm, const int64_t *idx, const int32_t *ok, int32_t j0, int32_t j1, int32_t drawN,
                    const rh_params *prm, int32_t f32, FitSetsOut *o)
{
    double fp[48], fn[48];
    for (int32_t j = j0; j < j1; j++) {
        if (ok != nullptr && ok[j] == 0) continue;
        for (int q = 0; q < drawN; q++) {
            const int64_t i = idx[(int64_t)j * drawN + q];
            if (i < 1) { o->rc = RH_E_INVALID; return; }
            for (int a = 0; a < 3; a++) { fp[3 * q + a] = xyz[3 * (i - 1) + a]; fn[3 * q + a] = nrm[3 * (i - 1) + a]; }
        }
        for (int t = 0; t < prm->n_shape_types; t++) {
            rh_shape s;
            memset(&s, 0, sizeof s);
            bool fitted = false;
            switch (prm->shape_types[t]) {
            case RH_PLANE: fitted = f32 ? fit_plane32(fp, fn, drawN, *prm, &s) : fit_plane(fp, fn, drawN, *prm, &s); break;
            case RH_SPHERE: fitted = f32 ? fit_sphere32(fp, fn, drawN, *prm, &s) : fit_sphere(fp, fn, drawN, *prm, &s); break;
            case RH_CYLINDER: fitted = f32 ? fit_cylinder32(fp, fn, drawN, *prm, &s) : fit_cylinder(fp, fn, drawN, *prm, &s); break;
            case RH_CONE: fitted = f32 ? fit_cone32(fp, fn, drawN, *prm, &s) : fit_cone(fp, fn, drawN, *prm, &s); break;
            default: o->rc = RH_E_INVALID; return;
            }
            if (fitted) { o->shapes.push_back(s); o->sets.push_back(j); }
        }
    }
}
}  // namespace

extern "C" int rh_fit_sets(const double *xyz, const double *nrm, const int64_t *idx, const int32_t *ok, int32_t k, int32_t drawN,
                           const rh_params *prm, int32_t f32, rh_shape *shapes_out, int32_t *set_out, int32_t cap, int32_t *n_out)
{
    if (!xyz || !nrm || (k > 0 && !idx) || !prm || !n_out || k < 0 || cap < 0 || (cap > 0 && !shapes_out)) { rh_set_error("rh_fit_sets: bad arguments"); return RH_E_INVALID; }
    if (drawN < 3 || drawN > 16) { rh_set_error("rh_fit_sets: drawN=%d outside 3..16", drawN); return RH_E_INVALID; }
    if (prm->n_shape_types < 0 || prm->n_shape_types > 8) { rh_set_error("rh_fit_sets: bad n_shape_types"); return RH_E_INVALID; }
    *n_out = 0;
    try {
        unsigned nt = 1;
        if (k >= 512) {   // (at least 128 sets per thread: a thread's start costs about as much as fitting fifty sets)
            nt = std::thread::hardware_concurrency();
            nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
            if (nt > (unsigned)k / 128) nt = (unsigned)k / 128;
        }
        std::vector<FitSetsOut> outs(nt);
        if (nt == 1) {
            fit_sets_range(xyz, nrm, idx, ok, 0, k, drawN, prm, f32, &outs[0]);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) {
                const int32_t j0 = (int32_t)((int64_t)k * t / nt), j1 = (int32_t)((int64_t)k * (t + 1) / nt);
                th.emplace_back(fit_sets_range, xyz, nrm, idx, ok, j0, j1, drawN, prm, f32, &outs[t]);
            }
            for (auto &x : th) x.join();
        }
        int32_t nout = 0;
        for (const FitSetsOut &o : outs) {
            if (o.rc != RH_OK) { rh_set_error("rh_fit_sets: a set holds an index below 1, or shape_types an unknown kind"); return o.rc; }
            for (size_t i = 0; i < o.shapes.size(); i++, nout++)
                if (nout < cap) { shapes_out[nout] = o.shapes[i]; if (set_out) set_out[nout] = o.sets[i]; }
        }
        *n_out = nout;
        if (nout > cap) { rh_set_error("rh_fit_sets: %d shapes fitted, capacity %d", nout, cap); return RH_E_CAPACITY; }
    } catch (const std::exception &) {
        rh_set_error("rh_fit_sets: out of host memory or threads");
        return RH_E_NOMEM;
    }
    return RH_OK;
}

uint64_t rh_rng_next_raw(rh_rng *r) { return rng_next(r); }   // rh_sample_sets (cloud.hip): the raw draws rh_rng_range scales

extern "C" int64_t rh_rng_range(rh_rng *r, int64_t n)
{
    return 1 + (int64_t)(((unsigned __int128)rng_next(r) * (unsigned __int128)(uint64_t)n) >> 64);
}

#ifdef RH_DIAG
// ---- self-check of the device sampler's search routines (ransac_hip.h), on the host --------------------------------
extern "C" int rh_dbg_oct_search_selftest(int64_t n, uint64_t seed, int64_t queries, int64_t *mismatches)
{
    if (n < 1 || queries < 0 || !mismatches) { rh_set_error("rh_dbg_oct_search_selftest: bad arguments"); return RH_E_INVALID; }
    uint64_t x = seed;
    auto rnd = [&]() { return splitmix(&x); };
    // a Morton order with clusters: a few thousand distinct deep cells, many points per cell, some exact duplicates
    std::vector<uint64_t> code((size_t)n);
    const int ncell = (int)std::min<int64_t>(n, 1 + (int64_t)(rnd() % 4096));
    std::vector<uint64_t> cell((size_t)ncell);
    for (uint64_t &c : cell) c = rnd() >> 1;   // 63 bits
    for (int64_t i = 0; i < n; i++) {
        const uint64_t c = cell[(size_t)(rnd() % (uint64_t)ncell)];
        const int keep = 3 * (int)(rnd() % 22);   // low bits to redraw: 0 (a duplicate of the cell's code) .. 63
        const uint64_t mask = keep >= 63 ? ~0ULL >> 1 : ((1ULL << keep) - 1ULL);
        code[(size_t)i] = (c & ~mask) | (rnd() & mask);
    }
    std::sort(code.begin(), code.end());
    const int64_t nwords = (n + 63) / 64;
    std::vector<uint64_t> men((size_t)nwords, 0);
    std::vector<int32_t> prefix((size_t)nwords + 1, 0);
    const uint64_t density = rnd() % 5;   // 0: every bit, else about 1/2, 1/4, 1/8, 1/16 of them, in runs
    for (int64_t i = 0; i < n; i++) {
        bool on = true;
        if (density) on = (rnd() & ((1ULL << density) - 1ULL)) == 0 || ((i >> 7) % 5 == 0);
        if (on) men[(size_t)(i >> 6)] |= 1ULL << (i & 63);
    }
    for (int64_t w = 0; w < nwords; w++) prefix[(size_t)w + 1] = prefix[(size_t)w] + (int32_t)__builtin_popcountll(men[(size_t)w]);
    const int64_t total = prefix[(size_t)nwords];
    std::vector<int32_t> perm((size_t)n), pos((size_t)n);
    for (int64_t i = 0; i < n; i++) { perm[(size_t)i] = (int32_t)i; pos[(size_t)i] = (int32_t)i; }
    rhfit::OctView plain;
    plain.code = code.data(); plain.perm = perm.data(); plain.pos = pos.data(); plain.men = men.data(); plain.prefix = prefix.data();
    plain.n = n; plain.nwords = nwords; plain.depth = 22;
    int64_t bad = 0;
    for (int tab_level = 1; tab_level <= 6; tab_level++) {
        const int64_t entries = ((int64_t)1 << (3 * (tab_level - 1))) + 1;
        const int tshift = 3 * (21 - (tab_level - 1));
        std::vector<int32_t> tab((size_t)entries);
        for (int64_t k = 0; k < entries; k++)
            tab[(size_t)k] = k == entries - 1 ? (int32_t)n : (int32_t)plain.lower_bound((uint64_t)k << tshift);
        rhfit::OctView fast = plain;
        fast.tab = tab.data(); fast.tab_level = tab_level; fast.code_o = code.data();   // (identity permutation: code_o = code)
        for (int64_t qi = 0; qi < queries; qi++) {
            const int64_t q0 = (int64_t)(rnd() % (uint64_t)n);
            const int level = 1 + (int)(rnd() % 22);
            int64_t lo0, hi0, lo1, hi1, lo2, hi2;
            plain.cell_bounds(level, q0, &lo0, &hi0);
            fast.cell_bounds(level, q0, &lo1, &hi1);
            fast.cell_bounds_code(level, code[(size_t)q0], &lo2, &hi2);
            bad += (lo0 != lo1) + (hi0 != hi1) + (lo0 != lo2) + (hi0 != hi2);
            if (!(lo0 <= q0 && q0 < hi0)) bad++;
            const int64_t base = plain.rank(lo0), ne = plain.rank(hi0) - base;
            if (ne < 1) continue;
            const int64_t wlo = lo0 >> 6, whi = (hi0 - 1) >> 6;
            int64_t r2[2] = { base + 1 + (int64_t)(rnd() % (uint64_t)ne), base + 1 + (int64_t)(rnd() % (uint64_t)ne) };
            const int64_t s0 = plain.select(r2[0]), s1 = plain.select(r2[1]);
            bad += (fast.select_in(r2[0], wlo, whi) != s0) + (fast.select_in(r2[1], wlo, whi) != s1);
            bad += (fast.select_in(r2[0], 0, nwords - 1) != s0);
            fast.select_in_many<2>(r2, wlo, whi);
            bad += (r2[0] != s0) + (r2[1] != s1);
            if (!(s0 >= lo0 && s0 < hi0)) bad++;
        }
    }
    for (int64_t qi = 0; qi < queries; qi++) {   // select_bit against clearing bits one by one
        const uint64_t m = rnd() | (1ULL << (rnd() % 64));
        const int pc = __builtin_popcountll(m);
        const int k = (int)(rnd() % (uint64_t)pc);
        uint64_t t = m;
        for (int i = 0; i < k; i++) t &= t - 1;
        bad += rhfit::OctView::select_bit(m, k) != __builtin_ctzll(t);
    }
    (void)total;
    *mismatches = bad;
    return RH_OK;
}
#endif   // RH_DIAG
