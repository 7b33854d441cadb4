// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the RANSAC hot path.
//
// Numerics contract: IEEE binary64, the reference's operation order, NO fused
// multiply-add (built with -ffp-contract=off), correctly rounded sqrt and divide.
// Each per-point test cites the reference function it restates (paths under
// /root/reference/src).  StaticArrays semantics: dot = (a1*b1 + a2*b2) + a3*b3,
// norm = sqrt of the same sum of squares, normalize(a) = (1/norm(a)) * a.
//
// Layout: points are structure-of-arrays, six planes (x y z nx ny nz) of `stride`
// doubles; lane i of a wave reads element base+i of each plane (512-B coalesced
// loads).  Candidate constants are wave-uniform: they are read with scalar loads
// into SGPRs, the per-point math runs on the FP64 vector ALU, inlier counts come
// from 64-bit wave ballots + s_bcnt1 (no cross-lane reduction tree).

#include <stdlib.h>

#include "rh_internal.h"
#include "score_device.h"
#include "score_device32.h"
#include "score4_device.h"

namespace {

using namespace rhdev;
using namespace rhdev32;

// -------------------------------------------------------------- prep ------
// Per-candidate constants of the CONSERVATIVE stages (box tests, band prefilter), in the record's free slots so that
// stage 1 does not recompute them per (chunk, tile): f[11] = sum |f[0..6]| (the magnitude behind box_slack); cylinder:
// f[8] = 1 + |c0|_1, f[9] = k = 2 - |a|^2, f[10] = 1 + |k| |a|^2 (closed-form rho^2 of the prefilter, score_device.h).
__host__ __device__ inline void prep_derived(rh_prep &o, int kind)
{
    o.f[11] = (((((fabs(o.f[0]) + fabs(o.f[1])) + fabs(o.f[2])) + fabs(o.f[3])) + fabs(o.f[4])) + fabs(o.f[5])) + fabs(o.f[6]);
    if (kind == RH_CYLINDER) {
        const double a2 = (o.f[0] * o.f[0] + o.f[1] * o.f[1]) + o.f[2] * o.f[2];
        const double k = 2.0 - a2;
        o.f[8] = (1.0 + fabs(o.f[3])) + (fabs(o.f[4]) + fabs(o.f[5]));
        o.f[9] = k;
        o.f[10] = 1.0 + fabs(k) * a2;
    }
}

__device__ __forceinline__ void prep_one(const rh_shape &s, rh_prep &o)
{
#pragma unroll
    for (int i = 0; i < 12; i++) o.f[i] = 0.0;
    const double sgn = s.outwards ? 1.0 : -1.0;
    switch (s.kind) {
    case RH_PLANE: {
        for (int i = 0; i < 6; i++) o.f[i] = s.v[i];
        const double a = s.v[3], b = s.v[4], c = s.v[5];
        const double inv = 1.0 / sqrt((a * a + b * b) + c * c);   // o_z = normalize(plane.normal)
        o.f[6] = inv * a; o.f[7] = inv * b; o.f[8] = inv * c;
        break;
    }
    case RH_SPHERE:
        for (int i = 0; i < 4; i++) o.f[i] = s.v[i];
        o.f[4] = sgn;
        break;
    case RH_CYLINDER:
        for (int i = 0; i < 7; i++) o.f[i] = s.v[i];
        o.f[7] = sgn;
        break;
    default:
        for (int i = 0; i < 6; i++) o.f[i] = s.v[i];
        o.f[6] = s.v[7];   // cos(-opang/2)
        o.f[7] = s.v[8];   // sin(-opang/2)
        o.f[8] = sgn;
        break;
    }
    prep_derived(o, s.kind);
}

// thresholds and magnitudes behind the classifier / culling records (rh4::cls_make) the prep kernels leave beside the bins
struct PreArgs { double eps[4]; double cosa[4]; double coord_mag, nrm_mag; int f32; float *box; int64_t bstride; };

__global__ void prep_sorted_kernel(const rh_shape *__restrict__ shapes, int32_t b, rh_prep *__restrict__ prep,
                                   int32_t *__restrict__ counts_zero, rh4::rh_cls *__restrict__ cls, const PreArgs QA)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b) {
        rh_prep P;
        const int kind = shapes[i].kind;
        prep_one(shapes[i], P);
        prep[i] = P;
        if (counts_zero != nullptr) counts_zero[i] = 0;
        if (cls != nullptr && kind >= 0 && kind <= 3)
            rh4::cls_make(P, kind, QA.eps[kind], QA.cosa[kind], QA.coord_mag, QA.nrm_mag, cls[i], QA.box + i, QA.bstride, nullptr, QA.f32 != 0);
    }
}

// unknown kinds (device-resident batch): bin by kind, one atomic per (wave, kind)
// counts_zero (optional): the batch's counts array, zeroed here; nk_other (optional): the OTHER half of a
// double-buffered bin-size array, zeroed for the next batch -- a scoring step then needs no memset at all
// `spread`: thread t takes candidate (t * spread) mod b (spread coprime to b, rh_spread_multiplier): neighbours
// in the batch -- often hypotheses of the same primitive -- end up in different 64-candidate chunks.  A chunk of
// near-identical candidates makes the tiles it touches 64 x heavier than the rest (measured on the cfg3 batch
// sorted by primitive: 0.231 ms instead of 0.180).
// cls (optional): the binary32 classifier record (rh4::rh_cls, score4_device.h) of every binned candidate, slot for slot,
// and its culling record in QA.box -- what the culled score kernel (score4.hip) reads

__global__ void prep_binned_kernel(const rh_shape *__restrict__ shapes, int32_t b, rh_prep *__restrict__ prep,
                                   int32_t *__restrict__ orig, int32_t *__restrict__ nk, int64_t cap,
                                   int32_t *__restrict__ counts_zero, int32_t *__restrict__ nk_other, int32_t spread,
                                   rh4::rh_cls *__restrict__ cls, const PreArgs QA, int32_t *__restrict__ zero_extra, int32_t zero_extra_n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // (a wider range of counts to zero: the other ranks' slices of a sharded batch, rh_score_batch_allreduce_dev)
    for (int i = t; i < zero_extra_n; i += gridDim.x * blockDim.x) zero_extra[i] = 0;
    const int lane = threadIdx.x & 63;
    rh_shape s;
    int kind = -1;
    if (t < 4 && nk_other != nullptr) nk_other[t] = 0;
    const int i = t < b ? (int)(((int64_t)t * spread) % b) : 0;
    if (t < b) {
        if (counts_zero != nullptr) counts_zero[t] = 0;
        s = shapes[i];
        if (s.kind >= 0 && s.kind <= 3) kind = s.kind;   // anything else: counts[i] stays 0
    }
    // slot in the kind's bin: the first lane of every kind present in the wave reserves the wave's share -- ONE atomic
    // instruction (up to four lanes, four counters), one round trip
    int slot = 0;
    {
        uint64_t mk = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t m = WB(kind == k);
            if (kind == k) mk = m;
        }
        const int leader = mk != 0 ? __builtin_ctzll(mk) : lane;
        int base = 0;
        if (kind >= 0 && lane == leader) base = atomicAdd(&nk[kind], __popcll(mk));
        base = __shfl(base, leader);
        slot = base + __popcll(mk & ((1ULL << lane) - 1ULL));
    }
    if (kind < 0) return;
    rh_prep P;
    prep_one(s, P);
    prep[(int64_t)kind * cap + slot] = P;
    orig[(int64_t)kind * cap + slot] = i;
    if (cls != nullptr)
        rh4::cls_make(P, kind, QA.eps[kind], QA.cosa[kind], QA.coord_mag, QA.nrm_mag, cls[(int64_t)kind * cap + slot],
                      QA.box + ((int64_t)kind * cap + slot), QA.bstride, nullptr, QA.f32 != 0);
}

// the sampler's candidate list (count on the device): bin by kind like prep_binned_kernel, zero the counts
__global__ void prep_entries_kernel(const rh_cand_entry *__restrict__ entries, const int32_t *__restrict__ count_ptr,
                                    int32_t cap_entries, rh_prep *__restrict__ prep, int32_t *__restrict__ orig,
                                    int32_t *__restrict__ nk, int64_t cap, int32_t *__restrict__ counts,
                                    rh4::rh_cls *__restrict__ cls, const PreArgs QA, const rh_oct_state *__restrict__ ost)
{
    // (an iteration of a chained octree window takes the entries from its start in the list on; nothing after a stop)
    if (ost != nullptr && ost->stop != 0) return;
    const int i = (ost != nullptr ? ost->start : 0) + blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int b = min(*count_ptr, cap_entries);
    rh_shape s;
    int kind = -1;
    if (i < b) {
        s = entries[i].shape;
        counts[i] = 0;
        if (s.kind >= 0 && s.kind <= 3) kind = s.kind;
    }
    // slot in the kind's bin: the first lane of every kind present in the wave reserves the wave's share -- ONE atomic
    // instruction (up to four lanes, four counters), one round trip
    int slot = 0;
    {
        uint64_t mk = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t m = WB(kind == k);
            if (kind == k) mk = m;
        }
        const int leader = mk != 0 ? __builtin_ctzll(mk) : lane;
        int base = 0;
        if (kind >= 0 && lane == leader) base = atomicAdd(&nk[kind], __popcll(mk));
        base = __shfl(base, leader);
        slot = base + __popcll(mk & ((1ULL << lane) - 1ULL));
    }
    if (kind < 0) return;
    rh_prep P;
    prep_one(s, P);
    prep[(int64_t)kind * cap + slot] = P;
    orig[(int64_t)kind * cap + slot] = i;
    if (cls != nullptr)
        rh4::cls_make(P, kind, QA.eps[kind], QA.cosa[kind], QA.coord_mag, QA.nrm_mag, cls[(int64_t)kind * cap + slot],
                      QA.box + ((int64_t)kind * cap + slot), QA.bstride, nullptr, QA.f32 != 0);
}

// ------------------------------------------------------------- score ------
// grid.x = point splits, grid.y = candidate tiles of RH_SC_CT.  A wave owns
// RH_SC_WAVE_PTS contiguous points per tile (PPT x 64, PPT points per lane in
// registers) and walks the block's candidates; per candidate it issues PPT tests per
// lane, PPT ballots, and one LDS atomic with the wave's popcount.
// F32: the points are a Float32 cloud's (exactly converted), the test is the binary32 one on the float record that follows
// from the candidate's binary64 record (prepf_of, score_device32.h) -- rh_ransac's liveness passes and small subsets
template <int KIND, bool MASK, bool F32 = false>
__global__ void __launch_bounds__(RH_SC_THREADS)
score_kernel(const double *__restrict__ pts, int64_t stride, int64_t s,
             const uint64_t *__restrict__ enabled_words, const rh_prep *__restrict__ prep,
             const int32_t *__restrict__ orig, const int32_t *__restrict__ nk_ptr, double eps, double cosa,
             int32_t *__restrict__ counts, uint64_t *__restrict__ masks, int64_t mask_stride)
{
    const int nk = *nk_ptr;
    const int c0 = blockIdx.y * RH_SC_CT;
    if (c0 >= nk) return;
    const int nc = min(RH_SC_CT, nk - c0);

    __shared__ int32_t lcnt[RH_SC_CT];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < RH_SC_CT) lcnt[tid] = 0;
    __syncthreads();

    const double *__restrict__ X = pts, *__restrict__ Y = pts + stride, *__restrict__ Z = pts + 2 * stride;
    const double *__restrict__ NX = pts + 3 * stride, *__restrict__ NY = pts + 4 * stride,
                 *__restrict__ NZ = pts + 5 * stride;
    const int64_t ntiles = (s + RH_SC_TILE - 1) / RH_SC_TILE;
    const int64_t swords = (s + 63) >> 6;
    static_assert(RH_SC_CT == 64, "one lane per candidate of the tile");
    int acc = 0;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * RH_SC_TILE + (int64_t)wave * RH_SC_WAVE_PTS;
        double px[RH_SC_PPT], py[RH_SC_PPT], pz[RH_SC_PPT], qx[RH_SC_PPT], qy[RH_SC_PPT], qz[RH_SC_PPT];
        uint64_t en[RH_SC_PPT];
#pragma unroll
        for (int p = 0; p < RH_SC_PPT; p++) {
            const int64_t i = wbase + p * 64 + lane;
            px[p] = X[i]; py[p] = Y[i]; pz[p] = Z[i];
            qx[p] = NX[i]; qy[p] = NY[i]; qz[p] = NZ[i];
            const int64_t gb = wbase + p * 64;
            uint64_t v = valid_mask(gb, s);
            if (enabled_words != nullptr && v != 0) v &= enabled_words[gb >> 6];
            en[p] = v;
        }
        // candidate constants: scalar loads, software-prefetched one candidate ahead
        rh_prep P = prep[c0];
        for (int c = 0; c < nc; c++) {
            const rh_prep Pn = prep[c0 + min(c + 1, nc - 1)];
            uint64_t bw[RH_SC_PPT];
            int n = 0;
#pragma unroll
            for (int p = 0; p < RH_SC_PPT; p++) {
                if (F32) bw[p] = test_point32<KIND>(prepf_of<KIND>(P), (float)px[p], (float)py[p], (float)pz[p], (float)qx[p], (float)qy[p],
                                                    (float)qz[p], eps, cosa) & en[p];
                else bw[p] = test_point<KIND>(P, px[p], py[p], pz[p], qx[p], qy[p], qz[p], eps, cosa) & en[p];
                n += __popcll(bw[p]);
            }
            acc += (lane == c) ? n : 0;   // lane c keeps candidate c0+c's count (RH_SC_CT == 64)
            if (MASK) {
                const int64_t w0 = wbase >> 6;
                if (lane < RH_SC_PPT && w0 + lane < swords) {
                    uint64_t v = bw[0];
#pragma unroll
                    for (int p = 1; p < RH_SC_PPT; p++) v = (lane == p) ? bw[p] : v;
                    masks[(int64_t)orig[c0 + c] * mask_stride + w0 + lane] = v;
                }
            }
            P = Pn;
        }
    }
    if (acc != 0) atomicAdd(&lcnt[lane], acc);
    __syncthreads();
    if (tid < nc) {
        const int v = lcnt[tid];
        if (v != 0) atomicAdd(&counts[orig[c0 + tid]], v);
    }
}

// one wave per 64-point group: axis-aligned box of its valid points
__global__ void __launch_bounds__(256)
group_bounds_kernel(const double *__restrict__ pts, int64_t stride, int64_t s, int64_t ngroups,
                    double *__restrict__ gb, int64_t gstride)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ngroups) return;
    const int64_t i = (g << 6) + lane;
    const bool valid = i < s;
    const int64_t ii = valid ? i : (g << 6);   // the first point of a group is always valid
    double mn[3], mx[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double v = pts[k * stride + ii];
        mn[k] = v; mx[k] = v;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fmin(mn[k], __shfl_xor(mn[k], off));
            mx[k] = fmax(mx[k], __shfl_xor(mx[k], off));
        }
    }
    if (lane == 0) {
        double h2 = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double c = 0.5 * mn[k] + 0.5 * mx[k];
            const double h = fmax(mx[k] - c, c - mn[k]) * 1.0000000000000009;
            gb[k * gstride + g] = c;
            gb[(3 + k) * gstride + g] = h;
            h2 += h * h;
        }
        gb[6 * gstride + g] = sqrt(h2) * 1.0000000000000009;
    }
}

// masks in internal (k-d leaf) order -> subset order.  One block per (candidate row, output segment): the segment of
// the output row is assembled in LDS -- every set bit of the row's input words is one LDS atomicOr at its subset
// position -- and written out once, coalesced, so the output needs neither a memset nor global atomics (the first
// version issued one global atomicOr per inlier: 12.4M of them on the cfg3 batch, 0.48 ms).
constexpr int RH_UNPERM_SEG_WORDS = 16384;   // 128 KB of LDS per block of 1024 threads (one block per CU then: 16 waves)

__global__ void __launch_bounds__(1024)
unpermute_masks_kernel(const uint64_t *__restrict__ in, const int32_t *__restrict__ perm, int64_t swords,
                       uint64_t *__restrict__ out)
{
    extern __shared__ unsigned long long seg[];
    const int64_t row = blockIdx.x;
    const int64_t w0 = (int64_t)blockIdx.y * RH_UNPERM_SEG_WORDS;
    const int nw = (int)(swords - w0 < RH_UNPERM_SEG_WORDS ? swords - w0 : RH_UNPERM_SEG_WORDS);
    for (int t = threadIdx.x; t < nw; t += 1024) seg[t] = 0ULL;
    __syncthreads();
    const uint64_t *__restrict__ src = in + row * swords;
    const int64_t lo = w0 << 6, hi = lo + ((int64_t)nw << 6);
    for (int64_t w = threadIdx.x; w < swords; w += 1024) {
        uint64_t m = src[w];
        while (m != 0) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t j = perm[(w << 6) + b];
            if (j >= lo && j < hi) atomicOr(&seg[(j - lo) >> 6], 1ULL << (j & 63));
        }
    }
    __syncthreads();
    uint64_t *__restrict__ dst = out + row * swords + w0;
    for (int t = threadIdx.x; t < nw; t += 1024) dst[t] = seg[t];
}

// the same for rows too long for one LDS segment (cfg5: 24 415 words): one thread per input word, one global atomicOr per
// inlier into the zeroed output (at that size the masks are sparse and two full passes over the rows cost more)
__global__ void unpermute_masks_atomic_kernel(const uint64_t *__restrict__ in, const int32_t *__restrict__ perm, int64_t swords,
                                              int64_t total_words, uint64_t *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_words) return;
    uint64_t m = in[t];
    const int64_t row = t / swords, w = t - row * swords;
    while (m != 0) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        const int32_t j = perm[(w << 6) + b];
        atomicOr((unsigned long long *)&out[row * swords + (j >> 6)], 1ULL << (j & 63));
    }
}

// ------------------------------------------------------------- refit ------
// One candidate (kernarg -> SGPRs), the whole cloud in original order.  Each wave owns
// 64-point words; mask word = ballot & enabled word; all-disabled words are skipped
// without touching the point planes.  Per-1024-word popcount sums feed the compaction.
// 64-point words per wave per iteration (12 loads in flight per lane).  Measured on the plane scan, 10M / 50M points
// (real clouds, bench.py): 4 words 0.0804 / 0.369 ms, 2 words 0.0764 / 0.355, 1 word 0.0792 / 0.345 -- fewer words per
// wave mean more waves in flight, and with one word the compiler sinks the second half of the loads below the
// wave-level early-out of the exact test (a word whose 64 points all fail the first half never fetches the other
// three planes: 0.048 ms on a cloud of random normals).
constexpr int RH_RF_WPW = 2;

template <int KIND>
__global__ void __launch_bounds__(256)
refit_mask_kernel(const double *__restrict__ pts, int64_t stride, int64_t n, int64_t nwords,
                  const uint64_t *__restrict__ enabled, const rh_prep P, double eps, double cosa,
                  uint64_t *__restrict__ mask_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const double *__restrict__ X = pts, *__restrict__ Y = pts + stride, *__restrict__ Z = pts + 2 * stride;
    const double *__restrict__ NX = pts + 3 * stride, *__restrict__ NY = pts + 4 * stride,
                 *__restrict__ NZ = pts + 5 * stride;
    const int64_t ngroups = (nwords + RH_RF_WPW - 1) / RH_RF_WPW;
    for (int64_t g = wave0; g < ngroups; g += nwaves) {
        const int64_t w0 = g * RH_RF_WPW;
        uint64_t en[RH_RF_WPW];
        double px[RH_RF_WPW], py[RH_RF_WPW], pz[RH_RF_WPW], qx[RH_RF_WPW], qy[RH_RF_WPW], qz[RH_RF_WPW];
#pragma unroll
        for (int k = 0; k < RH_RF_WPW; k++) {
            const int64_t w = w0 + k;
            en[k] = w < nwords ? (enabled[w] & valid_mask(w << 6, n)) : 0ULL;
        }
#pragma unroll
        for (int k = 0; k < RH_RF_WPW; k++) {
            // an all-disabled word re-reads the (L2-hot) first line of the planes instead of streaming its 3 KB
            const int64_t i = (en[k] != 0 ? ((w0 + k) << 6) : (int64_t)0) + lane;
            px[k] = X[i]; py[k] = Y[i]; pz[k] = Z[i];
            qx[k] = NX[i]; qy[k] = NY[i]; qz[k] = NZ[i];
        }
#pragma unroll
        for (int k = 0; k < RH_RF_WPW; k++) {
            const uint64_t b = test_point<KIND>(P, px[k], py[k], pz[k], qx[k], qy[k], qz[k], eps, cosa) & en[k];
            if (lane == 0 && w0 + k < nwords) mask_out[w0 + k] = b;
        }
    }
}

// popcount sums per RH_WORDS_PER_BLOCK words (used for the select directory)
__global__ void __launch_bounds__(256)
block_popc_kernel(const uint64_t *__restrict__ words, int64_t nwords, int32_t *__restrict__ block_sums)
{
    __shared__ int32_t red[4];
    const int64_t base = (int64_t)blockIdx.x * RH_WORDS_PER_BLOCK;
    int acc = 0;
    for (int k = threadIdx.x; k < RH_WORDS_PER_BLOCK; k += 256) {
        const int64_t w = base + k;
        if (w < nwords) acc += __popcll(words[w]);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// single-block exclusive scan of block_sums[0..nb) in place; block_sums[nb] and *total = grand total
__global__ void __launch_bounds__(1024)
scan_block_sums_kernel(int32_t *__restrict__ block_sums, int64_t nb, int32_t *__restrict__ total)
{
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int v = i < nb ? block_sums[i] : 0;
        int inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; k++) woff += wsum[k];
        const int carry = carry_s;
        if (i < nb) block_sums[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        block_sums[nb] = carry_s;
        *total = carry_s;
    }
}

// block b expands mask words [b*1024, (b+1)*1024): ascending 1-based indices.  Each thread owns
// 4 consecutive words: block-wide exclusive scan of their popcounts, then every thread writes the
// set bits of its own words (shape masks are sparse: a few bits per word).
// With `andnot_words` the block also clears the mask's bits there (invalidate_indexes! as enabled &= ~mask)
// and leaves the popcount of its updated words in andnot_sums[block]: the next select directory starts
// from those instead of a pass of its own.
__global__ void __launch_bounds__(256)
expand_mask_kernel(const uint64_t *__restrict__ mask, int64_t nwords, const int32_t *__restrict__ block_prefix,
                   int64_t *__restrict__ idx_out, int64_t cap, int32_t *__restrict__ word_prefix_out,
                   uint64_t *__restrict__ andnot_words, int32_t *__restrict__ andnot_sums,
                   int32_t *__restrict__ raw_total = nullptr)
{
    __shared__ int32_t wsum[4];
    __shared__ int32_t esum[4];
    __shared__ int32_t psum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * RH_WORDS_PER_BLOCK;
    // raw_total: block_prefix holds the blocks' popcounts, not their scan -- the block sums up its predecessors itself
    // (a few hundred values) and the last block leaves the grand total: one launch less than scanning them first
    int before = 0;
    if (raw_total != nullptr) {
        for (int k = threadIdx.x; k < (int)blockIdx.x; k += 256) before += block_prefix[k];
        for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
        if (lane == 0) psum[wave] = before;
    }
    uint64_t m[4];
    int pc[4], tsum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t w = base + threadIdx.x * 4 + k;
        m[k] = w < nwords ? mask[w] : 0ULL;
        pc[k] = __popcll(m[k]);
        tsum += pc[k];
    }
    if (andnot_words != nullptr) {
        int left = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t w = base + threadIdx.x * 4 + k;
            if (w < nwords) {
                const uint64_t e = andnot_words[w] & ~m[k];
                if (m[k] != 0) andnot_words[w] = e;
                left += __popcll(e);
            }
        }
        for (int off = 32; off > 0; off >>= 1) left += __shfl_down(left, off);
        if (lane == 0) esum[wave] = left;
    }
    int inc = tsum;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (andnot_words != nullptr && threadIdx.x == 0) andnot_sums[blockIdx.x] = esum[0] + esum[1] + esum[2] + esum[3];
    int woff = 0;
    for (int k = 0; k < wave; k++) woff += wsum[k];
    const int32_t bpre = raw_total != nullptr ? (psum[0] + psum[1] + psum[2] + psum[3]) : block_prefix[blockIdx.x];
    if (raw_total != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *raw_total = bpre + woff + inc;
    int64_t run = (int64_t)bpre + woff + inc - tsum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t w = base + threadIdx.x * 4 + k;
        if (word_prefix_out != nullptr && w < nwords) word_prefix_out[w] = (int32_t)run;
        if (idx_out != nullptr) {
            uint64_t bits = m[k];
            while (bits) {
                const int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                if (run < cap) idx_out[run] = (w << 6) + b + 1;
                run++;
            }
        } else {
            run += pc[k];
        }
    }
}

// ------------------------------------------------- enabled maintenance ----
__global__ void invalidate_idx_kernel(const int64_t *__restrict__ idx, int64_t n, int64_t npoints,
                                      uint64_t *__restrict__ enabled, const int32_t *__restrict__ pos,
                                      uint64_t *__restrict__ men)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t i0 = idx[k] - 1;
    if (i0 < 0 || i0 >= npoints) return;
    atomicAnd((unsigned long long *)&enabled[i0 >> 6], ~(1ULL << (i0 & 63)));
    if (men != nullptr) {
        const int32_t mp = pos[i0];
        atomicAnd((unsigned long long *)&men[mp >> 6], ~(1ULL << (mp & 63)));
    }
}

// One launch per change of the enabled bits: sub_enabled bit j = enabled[sub_idx0[j]], and the subset points
// whose bit went 1 -> 0 (with `reset`: every disabled one) are appended to `dis`, the list the liveness
// pass scores.  A wave regathers RH_SUBUPD_WPW consecutive words; a block reserves the range of its points
// with ONE atomicAdd on the list length, so the ranges of different blocks follow each other in arbitrary
// order (the liveness counts are sums over the points: only the culling boxes see the order), while the
// points of a block stay in internal (k-d leaf) order, i.e. spatially compact.
constexpr int RH_SUBUPD_WPW = 8;

__global__ void __launch_bounds__(256)
update_sub_enabled_kernel(const uint64_t *__restrict__ enabled, const int32_t *__restrict__ sub_idx0, int64_t s,
                          int64_t swords, uint64_t *__restrict__ sub_enabled, int reset,
                          const double *__restrict__ sub, int64_t sub_stride, double *__restrict__ dis,
                          int64_t dis_stride, int32_t *__restrict__ ndis)
{
    __shared__ int32_t wtot[4];
    __shared__ int32_t base_s;
    __shared__ uint16_t lst[4][RH_SUBUPD_WPW * 64];   // per wave: positions (relative to its first word) of the gone points
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t w0 = ((int64_t)blockIdx.x * 4 + wave) * RH_SUBUPD_WPW;
    // all the loads of the wave's words first (the two gathers are dependent; the words are not)
    int32_t i0[RH_SUBUPD_WPW];
    uint64_t ew[RH_SUBUPD_WPW];
#pragma unroll
    for (int k = 0; k < RH_SUBUPD_WPW; k++) {
        const int64_t j = ((w0 + k) << 6) + lane;
        i0[k] = j < s ? sub_idx0[j] : -1;
    }
#pragma unroll
    for (int k = 0; k < RH_SUBUPD_WPW; k++) ew[k] = i0[k] >= 0 ? enabled[i0[k] >> 6] : 0ULL;
    const bool own = lane < RH_SUBUPD_WPW && w0 + lane < swords;   // lane k keeps word w0 + k
    const uint64_t oldv = (own && !reset) ? sub_enabled[w0 + lane] : 0ULL;
    uint64_t newv = 0;
    int tot = 0;
#pragma unroll
    for (int k = 0; k < RH_SUBUPD_WPW; k++) {
        const bool bit = i0[k] >= 0 && ((ew[k] >> (i0[k] & 63)) & 1ULL);
        const uint64_t neww = __builtin_amdgcn_ballot_w64(bit);
        const uint64_t valid = w0 + k < swords ? valid_mask((w0 + k) << 6, s) : 0ULL;
        const uint64_t oldw = reset ? valid : (uint64_t)__shfl((unsigned long long)oldv, k);
        const uint64_t g = oldw & ~neww & valid;
        if (lane == k) newv = neww;
        if ((g >> lane) & 1ULL) lst[wave][tot + __popcll(g & ((1ULL << lane) - 1ULL))] = (uint16_t)(k * 64 + lane);
        tot += __popcll(g);
    }
    if (own) sub_enabled[w0 + lane] = newv;
    if (lane == 0) wtot[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        base_s = t > 0 ? atomicAdd(ndis, t) : 0;
    }
    __syncthreads();
    int64_t run = base_s;
    for (int k = 0; k < wave; k++) run += wtot[k];
    for (int e = lane; e < tot; e += 64) {
        const int64_t j = (w0 << 6) + lst[wave][e];
        const int64_t dst = run + e;
#pragma unroll
        for (int q = 0; q < 6; q++) dis[q * dis_stride + dst] = sub[q * sub_stride + j];
    }
}

// ---- liveness of a SMALL candidate store after an extraction: removeinvalidshapes! (fitting.jl:209-221)
// recomputed as "does the candidate contain one of the points that were just disabled".  All kinds in one
// launch; the end of the disabled list is read from device memory (the host has not seen it yet), the grid
// is sized for the longest list possible and the blocks beyond the end leave at once.  A wave keeps
// RH_SC_WAVE_PTS points in registers and walks every stored candidate; flags[base[kind] + slot] = 1 marks
// a dead one (the flags are all zero on entry, pack_live_kernel re-zeroes them).  grid.y splits the candidates
// of every kind into rows of RH_LIVE_CH: the walk is a chain of dependent scalar loads, so it is kept short.
constexpr int RH_LIVE_CH = 4;   // candidates of each kind per block row

template <int KIND>
__device__ __forceinline__ void live_kind(const rh_live_args &A, const double (&px)[RH_SC_PPT], const double (&py)[RH_SC_PPT],
                                          const double (&pz)[RH_SC_PPT], const double (&qx)[RH_SC_PPT],
                                          const double (&qy)[RH_SC_PPT], const double (&qz)[RH_SC_PPT], int64_t wbase,
                                          int64_t nd, int lane, int32_t *__restrict__ flags)
{
    const int nk = A.nk[KIND];
    if (nk == 0) return;
    uint64_t vm[RH_SC_PPT];
    bool any = false;
#pragma unroll
    for (int p = 0; p < RH_SC_PPT; p++) {
        const int64_t gb = wbase + p * 64;   // bits of the points in [first, nd)
        const int64_t skip = (int64_t)A.first[KIND] - gb;
        vm[p] = valid_mask(gb, nd) & (skip <= 0 ? ~0ULL : (skip >= 64 ? 0ULL : ~((1ULL << skip) - 1ULL)));
        any = any || vm[p] != 0;
    }
    if (!any) return;
    const rh_prep *__restrict__ prep = A.prep[KIND];
    const double eps = A.eps[KIND], cosa = A.cosa[KIND];
    const int c_lo = blockIdx.y * RH_LIVE_CH, c_hi = min(nk, c_lo + RH_LIVE_CH);   // this block row's candidates
    for (int c = c_lo; c < c_hi; c++) {
        const rh_prep P = prep[c];
        uint64_t hit = 0;
        if (A.f32) {   // Float32 cloud: the binary32 test on the float record the stored one implies
            const rh_prepf Pf = prepf_of<KIND>(P);
#pragma unroll
            for (int p = 0; p < RH_SC_PPT; p++)
                hit |= test_point32<KIND>(Pf, (float)px[p], (float)py[p], (float)pz[p], (float)qx[p], (float)qy[p], (float)qz[p], eps, cosa) & vm[p];
        } else {
#pragma unroll
            for (int p = 0; p < RH_SC_PPT; p++)
                hit |= test_point<KIND>(P, px[p], py[p], pz[p], qx[p], qy[p], qz[p], eps, cosa) & vm[p];
        }
        if (hit != 0 && lane == 0) flags[A.base[KIND] + c] = 1;
    }
}

__global__ void __launch_bounds__(RH_SC_THREADS)
liveness_small_kernel(const double *__restrict__ dis, int64_t stride, const int32_t *__restrict__ ndis_ptr, int64_t lo,
                      const rh_live_args A, int32_t *__restrict__ flags)
{
    const int64_t nd = *ndis_ptr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wbase = lo + (int64_t)blockIdx.x * RH_SC_TILE + (int64_t)wave * RH_SC_WAVE_PTS;
    if (wbase >= nd) return;
    double px[RH_SC_PPT], py[RH_SC_PPT], pz[RH_SC_PPT], qx[RH_SC_PPT], qy[RH_SC_PPT], qz[RH_SC_PPT];
#pragma unroll
    for (int p = 0; p < RH_SC_PPT; p++) {   // (the list has a tile of slack behind its capacity)
        const int64_t i = wbase + p * 64 + lane;
        px[p] = dis[i]; py[p] = dis[stride + i]; pz[p] = dis[2 * stride + i];
        qx[p] = dis[3 * stride + i]; qy[p] = dis[4 * stride + i]; qz[p] = dis[5 * stride + i];
    }
    live_kind<RH_PLANE>(A, px, py, pz, qx, qy, qz, wbase, nd, lane, flags);
    live_kind<RH_SPHERE>(A, px, py, pz, qx, qy, qz, wbase, nd, lane, flags);
    live_kind<RH_CYLINDER>(A, px, py, pz, qx, qy, qz, wbase, nd, lane, flags);
    live_kind<RH_CONE>(A, px, py, pz, qx, qy, qz, wbase, nd, lane, flags);
}

// the extraction's read-back in one small kernel: liveness flags (re-zeroed behind the copy) and the two list
// lengths, straight into pinned host memory
__global__ void __launch_bounds__(256)
pack_live_kernel(int32_t *__restrict__ flags, int32_t n_flags, int32_t *__restrict__ h_flags,
                 const int32_t *__restrict__ d_total, const int32_t *__restrict__ d_ndis, int32_t *__restrict__ h_scalars)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_flags) {
        h_flags[i] = flags[i];
        flags[i] = 0;
    }
    if (i == 0) {
        h_scalars[0] = *d_total;
        h_scalars[1] = *d_ndis;
    }
}

// (ranks / out may be pinned host memory: a call with a few ranks is then this one launch, nothing else)
__global__ void select_kernel(const uint64_t *__restrict__ enabled, const int32_t *__restrict__ word_prefix,
                              int64_t nwords, const int32_t *__restrict__ total_ptr, const int64_t *__restrict__ ranks,
                              int32_t k, int64_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const int32_t total = *total_ptr;   // the select directory's count of enabled points
    const int64_t r = ranks[i];
    if (r < 1 || r > total) { out[i] = 0; return; }
    // last word w with word_prefix[w] < r
    int64_t lo = 0, hi = nwords;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (word_prefix[mid] < r) lo = mid; else hi = mid;
    }
    uint64_t m = enabled[lo];
    int rem = (int)(r - word_prefix[lo]);
    for (int t = 1; t < rem; t++) m &= m - 1;
    out[i] = (lo << 6) + __ffsll((unsigned long long)m);
}

// samplepointcloud4! (fitting.jl:383-430) on the root cell for EVERY start position of a stream of raw 64-bit draws: thread
// p plays the call that would begin at raw[p] -- first point by rejection on rand(1:n) (:388-395), the others as the
// rand(1:count)-th enabled point with one redraw when it repeats the first (:414-423), all-different test (:425-428) -- and
// leaves the points, whether the set is usable and how many draws the call consumed.  The host then follows the chain
// p -> p + consumed[p]: k calls in a row cost one launch, and the result is what k sequential calls give, draw for draw.
// rec[p]: drawN points (1-based int64; 0 = none), then consumed (0 = the draws ran out before the call finished) and ok.
__global__ void __launch_bounds__(256)
sample_sets_seq_kernel(const uint64_t *__restrict__ enabled, const int32_t *__restrict__ word_prefix, int64_t nwords, int64_t n,
                       const int32_t *__restrict__ total_ptr, const uint64_t *__restrict__ raw, int32_t L, int32_t drawN,
                       int64_t *__restrict__ rec)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= L) return;
    const int64_t count = *total_ptr;
    int64_t *o = rec + (int64_t)p * (drawN + 2);
    for (int q = 0; q < drawN + 2; q++) o[q] = 0;
    auto range = [](uint64_t u, int64_t m) { return 1 + (int64_t)__umul64hi(u, (uint64_t)m); };   // rand(1:m), rh_rng_range
    auto select = [&](int64_t r) -> int64_t {
        int64_t lo = 0, hi = nwords;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (word_prefix[mid] < r) lo = mid; else hi = mid;
        }
        uint64_t m = enabled[lo];
        const int rem = (int)(r - word_prefix[lo]);
        for (int t = 1; t < rem; t++) m &= m - 1;
        return (lo << 6) + __ffsll((unsigned long long)m);
    };
    int q = p;
    int64_t sd[16];
    if (count <= 0) return;
    int64_t r1;
    for (;;) {
        if (q >= L) return;                                  // ran out: consumed stays 0
        r1 = range(raw[q++], n);
        if ((enabled[(r1 - 1) >> 6] >> ((r1 - 1) & 63)) & 1ULL) break;
    }
    if (count < drawN) { o[drawN] = q - p; return; }          // (false, ..): fitting.jl:409-411
    sd[0] = r1;
    for (int i = 1; i < drawN; i++) {
        if (q >= L) return;
        int64_t pick = select(range(raw[q++], count));
        if (pick == sd[0]) {
            if (q >= L) return;
            pick = select(range(raw[q++], count));
        }
        sd[i] = pick;
    }
    bool distinct = true;
    for (int a = 1; a < drawN; a++)
        for (int b = 0; b < a; b++) distinct = distinct && sd[a] != sd[b];
    for (int i = 0; i < drawN; i++) o[i] = sd[i];
    o[drawN] = q - p;
    o[drawN + 1] = distinct ? 1 : 0;
}

// AoS (Vector{SVector{3,Float64}}) -> SoA planes, optionally gathering through an index list
__global__ void transpose_kernel(const double *__restrict__ xyz, const double *__restrict__ nrm,
                                 const int32_t *__restrict__ gather, int64_t count, double *__restrict__ dst,
                                 int64_t stride)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const int64_t i = gather ? gather[j] : j;
    dst[j] = xyz[3 * i];
    dst[stride + j] = xyz[3 * i + 1];
    dst[2 * stride + j] = xyz[3 * i + 2];
    dst[3 * stride + j] = nrm[3 * i];
    dst[4 * stride + j] = nrm[3 * i + 1];
    dst[5 * stride + j] = nrm[3 * i + 2];
}

// 64-byte point records for the sampler's random gathers (one HBM line per point instead of six)
__global__ void pack_records_kernel(const double *__restrict__ xyz, const double *__restrict__ nrm, int64_t n,
                                    rh_f64x2 *__restrict__ rec)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    rh_f64x2 a, b, c, d;
    a.x = xyz[3 * j]; a.y = xyz[3 * j + 1];
    b.x = xyz[3 * j + 2]; b.y = nrm[3 * j];
    c.x = nrm[3 * j + 1]; c.y = nrm[3 * j + 2];
    d.x = 0.0; d.y = 0.0;
    rec[4 * j] = a; rec[4 * j + 1] = b; rec[4 * j + 2] = c; rec[4 * j + 3] = d;
}

// men[pos[i]] = enabled[i] for every point: one thread per Morton position
__global__ void oct_gather_enabled_kernel(const uint64_t *__restrict__ enabled, const int32_t *__restrict__ perm,
                                          int64_t n, uint64_t *__restrict__ men)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bit = false;
    if (j < n) {
        const int32_t i0 = perm[j];
        bit = (enabled[i0 >> 6] >> (i0 & 63)) & 1ULL;
    }
    const uint64_t w = __builtin_amdgcn_ballot_w64(bit);
    if ((threadIdx.x & 63) == 0 && (j >> 6) < (n + 63) / 64) men[j >> 6] = w;
}

__global__ void iota_kernel(int32_t *d, int32_t n, int32_t base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = base + i;
}

__global__ void gather_prep_kernel(const rh_prep *__restrict__ src, const int32_t *__restrict__ idx, int32_t n,
                                   rh_prep *__restrict__ dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

template <int KIND>
int launch_score(rh_cloud *c, const double *pts, int64_t stride, int64_t s, const uint64_t *en,
                 const rh_prep *prep, const int32_t *orig, const int32_t *nk, int32_t nk_bound, double eps,
                 double cosa, int32_t *counts, uint64_t *masks, int64_t mask_stride)
{
    const int ctiles = cdiv(nk_bound, RH_SC_CT);
    const int64_t ntiles = (s + RH_SC_TILE - 1) / RH_SC_TILE;
    if (ctiles == 0 || ntiles == 0) return RH_OK;
    if (ctiles > 65535) {   // grid.y limit: 4M candidates of one kind per launch
        rh_set_error("batch of %d candidates is too large for one launch (max %d per kind)", nk_bound, 65535 * RH_SC_CT);
        return RH_E_INVALID;
    }
    // Blocks that
    // share a candidate tile differ in blockIdx.x, so consecutive ids (dealt round-robin over the
    // XCDs) stream different point tiles against the same SGPR-resident candidates.
    int64_t splits = 65536 / ctiles;   // one tile per block unless the grid gets huge: best balance (measured)
    if (splits < 1) splits = 1;
    if (splits > ntiles) splits = ntiles;
    dim3 grid((unsigned)splits, (unsigned)ctiles);
    if (c->f32) {
        if (masks)
            hipLaunchKernelGGL((score_kernel<KIND, true, true>), grid, dim3(RH_SC_THREADS), 0, c->stream, pts, stride, s, en,
                               prep, orig, nk, eps, cosa, counts, masks, mask_stride);
        else
            hipLaunchKernelGGL((score_kernel<KIND, false, true>), grid, dim3(RH_SC_THREADS), 0, c->stream, pts, stride, s, en,
                               prep, orig, nk, eps, cosa, counts, masks, mask_stride);
    } else if (masks)
        hipLaunchKernelGGL((score_kernel<KIND, true>), grid, dim3(RH_SC_THREADS), 0, c->stream, pts, stride, s, en,
                           prep, orig, nk, eps, cosa, counts, masks, mask_stride);
    else
        hipLaunchKernelGGL((score_kernel<KIND, false>), grid, dim3(RH_SC_THREADS), 0, c->stream, pts, stride, s, en,
                           prep, orig, nk, eps, cosa, counts, masks, mask_stride);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

}  // namespace

int rhk_compact_generic(hipStream_t stream, const uint64_t *mask, int64_t nwords, int32_t *ws_block_sums,
                        int64_t *idx_out, int64_t cap, int32_t *d_total);

// host twin of prep_one (used when the candidate is passed by value as a kernel argument)
void rh_prep_host(const rh_shape &s, rh_prep *o)
{
    for (int i = 0; i < 12; i++) o->f[i] = 0.0;
    const double sgn = s.outwards ? 1.0 : -1.0;
    switch (s.kind) {
    case RH_PLANE: {
        for (int i = 0; i < 6; i++) o->f[i] = s.v[i];
        const double a = s.v[3], b = s.v[4], c = s.v[5];
        const double inv = 1.0 / __builtin_sqrt((a * a + b * b) + c * c);
        o->f[6] = inv * a; o->f[7] = inv * b; o->f[8] = inv * c;
        break;
    }
    case RH_SPHERE:
        for (int i = 0; i < 4; i++) o->f[i] = s.v[i];
        o->f[4] = sgn;
        break;
    case RH_CYLINDER:
        for (int i = 0; i < 7; i++) o->f[i] = s.v[i];
        o->f[7] = sgn;
        break;
    default:
        for (int i = 0; i < 6; i++) o->f[i] = s.v[i];
        o->f[6] = s.v[7];
        o->f[7] = s.v[8];
        o->f[8] = sgn;
        break;
    }
    prep_derived(*o, s.kind);
}

int rhk_transpose_aos(rh_cloud *c, const double *d_xyz, const double *d_nrm, int64_t n, const int32_t *d_gather,
                      int64_t count, double *dst, int64_t dst_stride)
{
    (void)n;
    if (count == 0) return RH_OK;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(count, 256)), dim3(256), 0, c->stream, d_xyz, d_nrm, d_gather,
                       count, dst, dst_stride);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// two int32 from device memory into (pinned) host memory, in stream order, without a copy-engine transfer
__global__ void fetch_i32_kernel(const int32_t *__restrict__ src0, const int32_t *__restrict__ src1,
                                 int32_t *__restrict__ h_dst)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        h_dst[0] = *src0;
        h_dst[1] = *src1;
    }
}

int rhk_fetch2_i32(rh_cloud *c, const int32_t *d_src0, const int32_t *d_src1, int32_t *h_pinned_dst)
{
    hipLaunchKernelGGL(fetch_i32_kernel, dim3(1), dim3(64), 0, c->stream, d_src0, d_src1, h_pinned_dst);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// `span`: an upper bound of the list entries behind `lo` (the true end is read on the device)
int rhk_liveness_small(rh_cloud *c, int64_t lo, int64_t span, const rh_live_args &A, int32_t *d_flags)
{
    if (span <= 0) return RH_OK;
    int nk_max = 0;
    for (int q = 0; q < 4; q++) nk_max = std::max(nk_max, (int)A.nk[q]);
    if (nk_max == 0) return RH_OK;
    hipLaunchKernelGGL(liveness_small_kernel, dim3(cdiv(span, RH_SC_TILE), cdiv(nk_max, RH_LIVE_CH)), dim3(RH_SC_THREADS), 0,
                       c->stream, c->dis, c->dis_stride, c->d_ndis, lo, A, d_flags);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_pack_live(rh_cloud *c, int32_t *d_flags, int32_t n_flags, int32_t *h_flags, int32_t *h_scalars)
{
    hipLaunchKernelGGL(pack_live_kernel, dim3(std::max(1, cdiv(n_flags, 256))), dim3(256), 0, c->stream, d_flags, n_flags,
                       h_flags, c->d_total, c->d_ndis, h_scalars);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_pack_records(rh_cloud *c, const double *d_xyz, const double *d_nrm, int64_t n, double *d_rec)
{
    if (n == 0) return RH_OK;
    hipLaunchKernelGGL(pack_records_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, d_xyz, d_nrm, n, (rh_f64x2 *)d_rec);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// The culled score path (score4.hip) is there for this cloud: a subset of RH_G2_MIN_POINTS points or more, in k-d leaf order
// with group boxes.  Smaller subsets are scored by the brute-force kernel (score_kernel).
bool rh_score_v4_enabled(const rh_cloud *c)
{
    return c->f32 ? c->f32_groups : c->use_groups;
}

// eps + cosa given (and the bins are the cloud's own): the prep kernel also leaves the classifier and culling records of
// the bins in c->d_qpre / c->d_box; c->qpre_v4 says so to the score dispatch
static PreArgs pre_args(rh_cloud *c, const double *eps, const double *cosa)
{
    PreArgs QA;
    for (int k = 0; k < 4; k++) { QA.eps[k] = eps ? eps[k] : 0.0; QA.cosa[k] = cosa ? cosa[k] : 0.0; }
    QA.coord_mag = c->coord_mag;
    QA.nrm_mag = c->nrm_mag;
    QA.f32 = c->f32 ? 1 : 0;
    QA.box = c->d_box;
    QA.bstride = 4 * c->batch_cap;
    c->qpre_v4 = eps != nullptr && cosa != nullptr && rh_score_v4_enabled(c) && c->d_box != nullptr && c->d_qpre != nullptr;
    return QA;
}

int rhk_prep_sorted(rh_cloud *c, const rh_shape *d_shapes_sorted, int32_t b, rh_prep *d_prep, int32_t *d_counts_to_zero,
                    const double *eps, const double *cosa)
{
    const bool own = d_prep == c->d_prep && eps != nullptr && cosa != nullptr;
    const PreArgs QA = pre_args(c, own ? eps : nullptr, own ? cosa : nullptr);
    if (b == 0) return RH_OK;
    hipLaunchKernelGGL(prep_sorted_kernel, dim3(cdiv(b, 256)), dim3(256), 0, c->stream, d_shapes_sorted, b, d_prep,
                       d_counts_to_zero, c->qpre_v4 ? (rh4::rh_cls *)c->d_qpre : (rh4::rh_cls *)nullptr, QA);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// prep + bin the first min(*d_count, cap_entries) entries into c->d_prep / c->d_orig / c->d_nk (batch_cap >= cap_entries)
int rhk_prep_entries(rh_cloud *c, const rh_cand_entry *d_entries, const int32_t *d_count, int32_t cap_entries,
                     int32_t launch_bound, int32_t *d_counts, int nk_is_zero, const double *eps, const double *cosa,
                     const rh_oct_state *ost)
{
    if (!nk_is_zero) RH_HIP(hipMemsetAsync(c->d_nk, 0, 4 * sizeof(int32_t), c->stream));
    const PreArgs QA = pre_args(c, eps, cosa);
    if (launch_bound <= 0) return RH_OK;
    hipLaunchKernelGGL(prep_entries_kernel, dim3(cdiv(launch_bound, 256)), dim3(256), 0, c->stream, d_entries, d_count,
                       cap_entries, c->d_prep, c->d_orig, c->d_nk, c->batch_cap, d_counts,
                       c->qpre_v4 ? (rh4::rh_cls *)c->d_qpre : (rh4::rh_cls *)nullptr, QA, ost);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// a multiplier near 0.618 b that is coprime to b: t -> (t * m) mod b is a permutation that sends neighbours far apart
int32_t rh_spread_multiplier(int32_t b)
{
    if (b < 4) return 1;
    auto gcd = [](int64_t x, int64_t y) { while (y) { const int64_t r = x % y; x = y; y = r; } return x; };
    int64_t m = (int64_t)(0.6180339887498949 * b) | 1;
    while (gcd(m, b) != 1) m += 2;
    return (int32_t)(m % b);
}

int rhk_prep_binned(rh_cloud *c, const rh_shape *d_shapes, int32_t b, rh_prep *d_prep, int32_t *d_orig,
                    int32_t *d_nk, int64_t cap, int32_t *d_counts_to_zero, int32_t *d_nk_other, int nk_is_zero,
                    const double *eps, const double *cosa)
{
    if (!nk_is_zero) RH_HIP(hipMemsetAsync(d_nk, 0, 4 * sizeof(int32_t), c->stream));
    const bool own = d_prep == c->d_prep && cap == c->batch_cap;
    PreArgs QA = pre_args(c, own ? eps : nullptr, own ? cosa : nullptr);
    if (b == 0) return RH_OK;
    const int no_spread = rh_opt_on(c, RH_OPT_NO_SPREAD) ? 1 : 0;
    // (consumed by this launch: set by rh_score_batch_allreduce_dev around its score call)
    int32_t *zx = c->zero_extra;
    const int32_t zxn = c->zero_extra_n;
    c->zero_extra = nullptr; c->zero_extra_n = 0;
#ifndef RH_PREP_BLOCK
#define RH_PREP_BLOCK 64   // (one wave per block: the launch is a latency chain -- shape, bin reservation, divisions, records -- and 64 blocks spread it over 64 CUs: cfg3 step 0.0860 -> 0.0846 ms, cfg2 0.0431 -> 0.0421; 128: the same; 256: round 4)
#endif
    hipLaunchKernelGGL(prep_binned_kernel, dim3(cdiv(b, RH_PREP_BLOCK)), dim3(RH_PREP_BLOCK), 0, c->stream, d_shapes, b, d_prep, d_orig,
                       d_nk, cap, d_counts_to_zero, d_nk_other, no_spread ? 1 : rh_spread_multiplier(b),
                       c->qpre_v4 ? (rh4::rh_cls *)c->d_qpre : (rh4::rh_cls *)nullptr, QA, zx, zxn);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_score_kind(rh_cloud *c, int kind, const double *pts, int64_t stride, int64_t s, const uint64_t *en,
                   const rh_prep *d_prep, const int32_t *d_orig, const int32_t *d_nk, int32_t nk_bound, double eps,
                   double cosa, int32_t *d_counts, uint64_t *d_masks, int64_t mask_stride)
{
    switch (kind) {
    case RH_PLANE: return launch_score<RH_PLANE>(c, pts, stride, s, en, d_prep, d_orig, d_nk, nk_bound, eps, cosa, d_counts, d_masks, mask_stride);
    case RH_SPHERE: return launch_score<RH_SPHERE>(c, pts, stride, s, en, d_prep, d_orig, d_nk, nk_bound, eps, cosa, d_counts, d_masks, mask_stride);
    case RH_CYLINDER: return launch_score<RH_CYLINDER>(c, pts, stride, s, en, d_prep, d_orig, d_nk, nk_bound, eps, cosa, d_counts, d_masks, mask_stride);
    case RH_CONE: return launch_score<RH_CONE>(c, pts, stride, s, en, d_prep, d_orig, d_nk, nk_bound, eps, cosa, d_counts, d_masks, mask_stride);
    }
    rh_set_error("unknown shape kind %d", kind);
    return RH_E_INVALID;
}

int rhk_refit_mask(rh_cloud *c, const rh_prep &P, int kind, double eps, double cosa, bool apply)
{
    c->k_applied = false;
    c->k_sums_ready = false;
    if (c->nwords == 0) return RH_OK;
    if (rhk_refit_is_culled(c)) {
        RH_TRY(rhk_refitk_mask(c, P, kind, eps, cosa, apply));
        c->k_applied = apply;
        c->k_sums_ready = apply;
        return RH_OK;
    }
    const int env_blocks = (int)rh_opt_int(c, RH_OPT_REFIT_BLOCKS, 0);
    int64_t blocks = cdiv(c->nwords, 4 * RH_RF_WPW);
    // one group of words per wave up to 32768 blocks, a grid-stride loop beyond (measured with 4-word groups, plane
    // scan: 10M points 0.0771 ms at 2048 blocks, 0.0743 at 4096-9766; 50M points 0.364 ms at 2048, 0.350 at 32768+)
    const int64_t cap = env_blocks > 0 ? env_blocks : 32768;
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), blk(256);
#define RH_LAUNCH_REFIT(K)                                                                                        \
    hipLaunchKernelGGL((refit_mask_kernel<K>), grid, blk, 0, c->stream, c->full, c->n_pad, c->n, c->nwords,       \
                       c->enabled, P, eps, cosa, c->refit_mask)
    switch (kind) {
    case RH_PLANE: RH_LAUNCH_REFIT(RH_PLANE); break;
    case RH_SPHERE: RH_LAUNCH_REFIT(RH_SPHERE); break;
    case RH_CYLINDER: RH_LAUNCH_REFIT(RH_CYLINDER); break;
    case RH_CONE: RH_LAUNCH_REFIT(RH_CONE); break;
    default: rh_set_error("unknown shape kind %d", kind); return RH_E_INVALID;
    }
#undef RH_LAUNCH_REFIT
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_compact_mask(rh_cloud *c, const uint64_t *mask, int64_t nwords, int64_t *idx_out, int64_t cap,
                     int32_t *d_total)
{
    return rhk_compact_generic(c->stream, mask, nwords, c->block_sums, idx_out, cap, d_total);
}

// refit list + invalidate_indexes! in one compaction: idx_out = the set bits of refit_mask (ascending,
// 1-based), enabled &= ~refit_mask, and the per-block popcounts of the updated enabled words are left for
// the next select directory
int rhk_compact_refit_apply(rh_cloud *c)
{
    c->select_valid = false;
    c->en_sums_valid = false;
    if (!c->k_applied) c->k_men_valid = false;   // (a culled scan with `apply` has cleared the Morton-order bits itself)
    c->k_applied = false;
    // (the culled scan's last pass has left the per-block popcounts of refit_mask in block_sums already)
    if (c->nblocks > 0 && !c->k_sums_ready)
        hipLaunchKernelGGL(block_popc_kernel, dim3((unsigned)c->nblocks), dim3(256), 0, c->stream, c->refit_mask, c->nwords,
                           c->block_sums);
    c->k_sums_ready = false;
    const bool raw = c->nblocks > 0 && c->nblocks <= 2048;   // the expansion sums the blocks before it itself
    if (!raw) hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, c->stream, c->block_sums, c->nblocks, c->d_total);
    if (c->nblocks > 0)
        hipLaunchKernelGGL(expand_mask_kernel, dim3((unsigned)c->nblocks), dim3(256), 0, c->stream, c->refit_mask, c->nwords,
                           c->block_sums, c->idx_out, c->n, (int32_t *)nullptr, c->enabled, c->en_block_sums,
                           raw ? c->d_total : (int32_t *)nullptr);
    RH_HIP(hipGetLastError());
    c->en_sums_valid = c->nblocks > 0;
    return RH_OK;
}

int rhk_invalidate_idx(rh_cloud *c, const int64_t *d_idx, int64_t n)
{
    if (n == 0) return RH_OK;
    c->en_sums_valid = false;
    const bool both = c->k_built && c->k_men_valid;   // keep the Morton-order bits in step (else they are regathered later)
    hipLaunchKernelGGL(invalidate_idx_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, d_idx, n, c->n, c->enabled,
                       both ? c->oct_pos : (const int32_t *)nullptr, both ? c->oct_men : (uint64_t *)nullptr);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// subset bits + the list of disabled subset points, after any change of the enabled bits (one launch;
// `reset_list` rebuilds the list from every disabled point)
int rhk_rebuild_sub_enabled(rh_cloud *c, bool reset_list)
{
    if (reset_list) RH_HIP(hipMemsetAsync(c->d_ndis, 0, sizeof(int32_t), c->stream));
    if (c->s == 0) return RH_OK;
    hipLaunchKernelGGL(update_sub_enabled_kernel, dim3(cdiv(c->swords, 4 * RH_SUBUPD_WPW)), dim3(256), 0, c->stream,
                       c->enabled, c->sub_idx0, c->s, c->swords, c->sub_enabled, reset_list ? 1 : 0, c->sub, c->s_pad,
                       c->dis, c->dis_stride, c->d_ndis);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// one thread per bit: the set bits of word w land at word_prefix[w] + (rank inside the word);
// a wave owns one word, so the writes of a dense mask are coalesced
static __global__ void __launch_bounds__(256)
expand_dense_kernel(const uint64_t *__restrict__ mask, int64_t nwords, const int32_t *__restrict__ word_prefix,
                    int32_t *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t w = g >> 6;
    if (w >= nwords) return;
    const int lane = threadIdx.x & 63;
    const uint64_t m = mask[w];
    if ((m >> lane) & 1ULL) out[word_prefix[w] + __popcll(m & ((1ULL << lane) - 1ULL))] = (int32_t)g;
}

// select directory of the enabled bits: word_prefix + d_total.  The per-block popcounts come from the
// extraction that changed the bits when it left them (rhk_compact_refit_apply), else from a pass here.
int rhk_build_select(rh_cloud *c)
{
    c->sel_valid = false;
    c->crec_valid = false;   // the rank order changes with the bits
    c->very_long_windows = 0;
    if (c->nwords == 0) { c->select_valid = true; return RH_OK; }
    if (!c->en_sums_valid)
        hipLaunchKernelGGL(block_popc_kernel, dim3((unsigned)c->nblocks), dim3(256), 0, c->stream, c->enabled, c->nwords,
                           c->en_block_sums);
    c->en_sums_valid = false;   // scanned in place
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, c->stream, c->en_block_sums, c->nblocks, c->d_total);
    hipLaunchKernelGGL(expand_mask_kernel, dim3((unsigned)c->nblocks), dim3(256), 0, c->stream, c->enabled, c->nwords,
                       c->en_block_sums, (int64_t *)nullptr, (int64_t)0, c->word_prefix, (uint64_t *)nullptr,
                       (int32_t *)nullptr);
    RH_HIP(hipGetLastError());
    c->select_valid = true;
    return RH_OK;
}

// flat select list (sel_list[r] = index of the (r + 1)-th enabled point): what long sampling windows read
// instead of searching the directory; 4 bytes per enabled point, so it is only built on demand
int rhk_build_sel_list(rh_cloud *c)
{
    if (!c->select_valid) RH_TRY(rhk_build_select(c));
    if (c->sel_valid) return RH_OK;
    if (c->nwords > 0) {
        hipLaunchKernelGGL(expand_dense_kernel, dim3((unsigned)cdiv(c->nwords * 64, 256)), dim3(256), 0, c->stream, c->enabled,
                           c->nwords, c->word_prefix, c->sel_list);
        RH_HIP(hipGetLastError());
    }
    c->sel_valid = true;
    return RH_OK;
}

int rhk_select(rh_cloud *c, const int64_t *d_ranks, int32_t k, int64_t *d_out)
{
    if (k == 0) return RH_OK;
    hipLaunchKernelGGL(select_kernel, dim3(cdiv(k, 256)), dim3(256), 0, c->stream, c->enabled, c->word_prefix,
                       c->nwords, c->d_total, d_ranks, k, d_out);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_sample_sets_seq(rh_cloud *c, const uint64_t *d_raw, int32_t L, int32_t drawN, int64_t *d_rec)
{
    if (L == 0) return RH_OK;
    hipLaunchKernelGGL(sample_sets_seq_kernel, dim3(cdiv(L, 256)), dim3(256), 0, c->stream, c->enabled, c->word_prefix, c->nwords,
                       c->n, c->d_total, d_raw, L, drawN, d_rec);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_count_enabled(rh_cloud *c, int64_t *out)
{
    *out = 0;
    if (c->nwords == 0) return RH_OK;
    hipLaunchKernelGGL(block_popc_kernel, dim3((unsigned)c->nblocks), dim3(256), 0, c->stream, c->enabled, c->nwords,
                       c->block_sums);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, c->stream, c->block_sums, c->nblocks, c->d_total);
    RH_HIP(hipGetLastError());
    int32_t total = 0;
    RH_HIP(hipMemcpyAsync(&total, c->d_total, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    *out = total;
    c->select_valid = false;   // block_sums was clobbered
    return RH_OK;
}

int rhk_iota(rh_cloud *c, int32_t *d, int32_t n, int32_t base)
{
    if (n == 0) return RH_OK;
    hipLaunchKernelGGL(iota_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, d, n, base);
    RH_HIP(hipGetLastError());
    return RH_OK;
}


// ---- device-managed candidate store: compaction after an extraction (rh_internal.h, rh_store_plan) ----------------------
namespace {
__device__ __forceinline__ bool store_alive(const rh_store_plan &P, int64_t g, int &q_out, int32_t &slot_out, bool &valid)
{
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; k++) q += g >= P.pbase[k] ? 1 : 0;
    const int32_t slot = (int32_t)(g - P.pbase[q]);
    q_out = q; slot_out = slot;
    valid = slot < P.n[q];
    return valid && P.counts[g] == 0 && P.id[q][slot] != P.extracted_id;
}

__global__ void __launch_bounds__(RH_STORE_PAD)
store_count_kernel(const rh_store_plan P, int32_t *__restrict__ blk_cnt)
{
    __shared__ int32_t wc[RH_STORE_PAD / 64];
    const int64_t g = (int64_t)blockIdx.x * RH_STORE_PAD + threadIdx.x;
    int q; int32_t slot; bool valid;
    const bool alive = store_alive(P, g, q, slot, valid);
    const uint64_t bm = __builtin_amdgcn_ballot_w64(alive);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bm);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t a = 0;
        for (int w = 0; w < RH_STORE_PAD / 64; w++) a += wc[w];
        blk_cnt[blockIdx.x] = a;
    }
}

// one block: exclusive scan of the block counts; the kinds' offsets and new lengths; the dead-list counter back to zero
__global__ void __launch_bounds__(1024)
store_scan_kernel(const rh_store_plan P, const int32_t *__restrict__ blk_cnt, int32_t nblocks, int32_t *__restrict__ blk_off,
                  int32_t *__restrict__ kind_off, int32_t *__restrict__ dead_counter, int32_t *__restrict__ h_out)
{
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x;
    const int32_t per = (nblocks + 1023) / 1024;
    const int32_t b0 = min(nblocks, tid * per), b1 = min(nblocks, b0 + per);
    int32_t a = 0;
    for (int32_t b = b0; b < b1; b++) a += blk_cnt[b];
    // exclusive scan of the 1024 partial sums: shuffles inside the waves, then over the 16 wave totals
    const int lane = tid & 63, wv = tid >> 6;
    int32_t incl = a;
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    if (tid == 0) {
        int32_t run = 0;
        for (int w = 0; w < 16; w++) { const int32_t v = part[w]; part[w] = run; run += v; }
    }
    __syncthreads();
    int32_t run = part[wv] + incl - a;
    for (int32_t b = b0; b < b1; b++) { blk_off[b] = run; run += blk_cnt[b]; }
    if (tid == 1023) blk_off[nblocks] = run;   // (the last thread's range ends the array: the total)
    __syncthreads();
    if (tid == 0) {
        int32_t total_old = 0, total_new = 0;
        for (int q = 0; q < 4; q++) {
            const int32_t lo = blk_off[P.pbase[q] / RH_STORE_PAD], hi = blk_off[P.pbase[q + 1] / RH_STORE_PAD];
            kind_off[q] = lo;
            h_out[q] = hi - lo;
            total_old += P.n[q];
            total_new += hi - lo;
        }
        h_out[4] = total_old - total_new;
        *dead_counter = total_old - total_new;   // (the length of the dead list: store_move_kernel places its entries by rank, no atomics)
    }
}

__global__ void __launch_bounds__(RH_STORE_PAD)
store_move_kernel(const rh_store_plan P, const int32_t *__restrict__ blk_off, const int32_t *__restrict__ kind_off,
                  int32_t *__restrict__ dead_counter, int32_t *__restrict__ h_dead, rh_store_best *__restrict__ h_best)
{
    __shared__ int32_t wc[RH_STORE_PAD / 64];
    __shared__ double wE[RH_STORE_PAD / 64];
    __shared__ long long wI[RH_STORE_PAD / 64];
    const int64_t g = (int64_t)blockIdx.x * RH_STORE_PAD + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int q; int32_t slot; bool valid;
    const bool alive = store_alive(P, g, q, slot, valid);
    const uint64_t bm = __builtin_amdgcn_ballot_w64(alive);
    if (lane == 0) wc[wv] = __popcll(bm);
    __syncthreads();
    // alive entries before this one, over the whole index space
    int32_t apos = blk_off[blockIdx.x] + __popcll(bm & ((1ULL << lane) - 1ULL));
    for (int w = 0; w < wv; w++) apos += wc[w];
    if (alive) {
        const int32_t pos = apos - kind_off[q];
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *src = (const u32x4 *)(P.prep[q] + slot);
        u32x4 *dst = (u32x4 *)(P.spare[q] + pos);
        u32x4 v[sizeof(rh_prep) / 16];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(rh_prep) / 16); i++) v[i] = src[i];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(rh_prep) / 16); i++) dst[i] = v[i];
        P.spare_id[q][pos] = P.id[q][slot];
        P.spare_E[q][pos] = P.E[q][slot];
    } else if (valid) {
        // dead: its place in the dead list = valid entries before it - alive entries before it (returning atomics on one
        // counter, one per wave, serialised in L2: 30 of the kernel's 40 us on a 100 000-entry store)
        int32_t vbefore = slot;
        for (int k = 0; k < q; k++) vbefore += P.n[k];
        h_dead[vbefore - apos] = P.id[q][slot];
    }
    // the block's best survivor: greatest score, smallest id among equals
    double bE = 0.0;
    long long bI = -1;
    if (alive) { bE = P.E[q][slot]; bI = P.id[q][slot]; }
    auto better = [](double e1, long long i1, double e2, long long i2) { return i1 >= 0 && (i2 < 0 || e1 > e2 || (e1 == e2 && i1 < i2)); };
    for (int off = 32; off > 0; off >>= 1) {
        const double oE = __shfl_xor(bE, off);
        const long long oI = __shfl_xor(bI, off);
        if (better(oE, oI, bE, bI)) { bE = oE; bI = oI; }
    }
    if (lane == 0) { wE[wv] = bE; wI[wv] = bI; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RH_STORE_PAD / 64; w++)
            if (better(wE[w], wI[w], bE, bI)) { bE = wE[w]; bI = wI[w]; }
        h_best[blockIdx.x].E = bE;
        h_best[blockIdx.x].id = bI;
    }
}
}  // namespace

namespace {
// the dead list and the blocks' best entries -> pinned host memory, as coalesced stores (scattered 4-byte stores from
// store_move_kernel's waves crossed PCIe one by one: 40 us per compaction of a 100 000-entry store)
__global__ void __launch_bounds__(1024)
store_flush_kernel(const int32_t *__restrict__ d_dead, const int32_t *__restrict__ dead_counter, int32_t cap, const rh_store_best *__restrict__ d_best,
                   int32_t nblocks, int32_t *__restrict__ h_dead, rh_store_best *__restrict__ h_best)
{
    const int32_t n = min(*dead_counter, cap);
    for (int32_t i = threadIdx.x; i < n; i += 1024) h_dead[i] = d_dead[i];
    for (int32_t i = threadIdx.x; i < nblocks; i += 1024) h_best[i] = d_best[i];
}
}  // namespace

int rhk_store_compact(rh_cloud *c, const rh_store_plan &P, int32_t *d_work, int32_t *h_out, int32_t *h_dead, rh_store_best *h_best,
                      int32_t *d_dead, rh_store_best *d_best)
{
    const int32_t nblocks = P.pbase[4] / RH_STORE_PAD;
    if (nblocks <= 0) { for (int i = 0; i < 5; i++) h_out[i] = 0; return RH_OK; }
    int32_t *blk_cnt = d_work, *blk_off = d_work + nblocks, *kind_off = d_work + 2 * nblocks + 1, *counter = d_work + 2 * nblocks + 8;
    hipLaunchKernelGGL(store_count_kernel, dim3((unsigned)nblocks), dim3(RH_STORE_PAD), 0, c->stream, P, blk_cnt);
    hipLaunchKernelGGL(store_scan_kernel, dim3(1), dim3(1024), 0, c->stream, P, blk_cnt, nblocks, blk_off, kind_off, counter, h_out);
    hipLaunchKernelGGL(store_move_kernel, dim3((unsigned)nblocks), dim3(RH_STORE_PAD), 0, c->stream, P, blk_off, kind_off, counter, d_dead, d_best);
    int32_t total = 0;
    for (int q = 0; q < 4; q++) total += P.n[q];
    hipLaunchKernelGGL(store_flush_kernel, dim3(1), dim3(1024), 0, c->stream, d_dead, counter, total, d_best, nblocks, h_dead, h_best);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_gather_prep(rh_cloud *c, const rh_prep *src, const int32_t *d_idx, int32_t n, rh_prep *dst)
{
    if (n == 0) return RH_OK;
    hipLaunchKernelGGL(gather_prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, src, d_idx, n, dst);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// mask (nwords 64-bit words) -> ascending 1-based set-bit positions; ws_block_sums needs nb+2 ints
int rhk_compact_generic(hipStream_t stream, const uint64_t *mask, int64_t nwords, int32_t *ws_block_sums,
                        int64_t *idx_out, int64_t cap, int32_t *d_total)
{
    const int64_t nb = (nwords + RH_WORDS_PER_BLOCK - 1) / RH_WORDS_PER_BLOCK;
    if (nb > 0) hipLaunchKernelGGL(block_popc_kernel, dim3((unsigned)nb), dim3(256), 0, stream, mask, nwords, ws_block_sums);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, stream, ws_block_sums, nb, d_total);
    if (nb > 0)
        hipLaunchKernelGGL(expand_mask_kernel, dim3((unsigned)nb), dim3(256), 0, stream, mask, nwords, ws_block_sums,
                           idx_out, cap, (int32_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// all kinds against subset 1 in one launch of the culled kernel (score4.hip); cls / box: the bins' classifier and culling
// records (prep kernels); nk_total_bound >= the number of candidates over all kinds
int rhk_score_all_groups(rh_cloud *c, const uint64_t *const en[4], const rh_prep *const prep[4],
                         const int32_t *const orig[4], const int32_t *const nk[4], int32_t nk_total_bound,
                         const double eps[4], const double cosa[4], int32_t *d_counts, uint64_t *d_masks_int,
                         const void *const cls[4], const float *const box[4], int64_t bstride)
{
    if (c->ngroups == 0 || nk_total_bound <= 0) return RH_OK;
    if (cls == nullptr || box == nullptr || !rh_score_v4_enabled(c) || c->gb32 == nullptr || (d_masks_int != nullptr && !c->masks4)) {
        rh_set_error("internal: the culled score kernel without its records (classifier, culling, group boxes, mask lists)");
        return RH_E_INTERNAL;
    }
    return rhk_score4_all(c, en, prep, cls, box, bstride, orig, nk, nk_total_bound, eps, cosa, d_counts, d_masks_int,
                          d_masks_int ? c->d_occ : nullptr, c->mstride4);
}

// liveness pass of a cloud without the culled path: candidates against dis[first, first + cnt) (counts only), brute force
int rhk_score_kind_dis(rh_cloud *c, int kind, int64_t first, int64_t cnt, const rh_prep *d_prep, const int32_t *d_orig,
                       const int32_t *d_nk, int32_t nk_bound, double eps, double cosa, int32_t *d_counts)
{
    if (cnt <= 0 || nk_bound <= 0) return RH_OK;
    return rhk_score_kind(c, kind, c->dis + first, c->dis_stride, cnt, nullptr, d_prep, d_orig, d_nk, nk_bound, eps,
                          cosa, d_counts, nullptr, 0);
}

int rhk_group_bounds_of(rh_cloud *c, const double *pts, int64_t stride, int64_t count, int64_t ngroups, double *gb, int64_t gstride)
{
    if (ngroups == 0) return RH_OK;
    hipLaunchKernelGGL(group_bounds_kernel, dim3(cdiv(ngroups, 4)), dim3(256), 0, c->stream, pts, stride, count, ngroups, gb, gstride);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_group_bounds(rh_cloud *c)
{
    if (c->ngroups == 0) return RH_OK;
    hipLaunchKernelGGL(group_bounds_kernel, dim3(cdiv(c->ngroups, 4)), dim3(256), 0, c->stream, c->sub, c->s_pad, c->s,
                       c->ngroups, c->gb, c->ng_pad);
    RH_HIP(hipGetLastError());
    return rhk_gb32_build(c);   // the binary32 twins (v4 score kernel)
}

int rhk_unpermute_masks(rh_cloud *c, const uint64_t *d_in, int32_t b, uint64_t *d_out)
{
    if (b == 0 || c->swords == 0) return RH_OK;
    if (c->swords > RH_UNPERM_SEG_WORDS) {
        const int64_t total = (int64_t)b * c->swords;
        RH_HIP(hipMemsetAsync(d_out, 0, sizeof(uint64_t) * (size_t)total, c->stream));
        hipLaunchKernelGGL(unpermute_masks_atomic_kernel, dim3(cdiv(total, 256)), dim3(256), 0, c->stream, d_in, c->sub_perm,
                           c->swords, total, d_out);
        RH_HIP(hipGetLastError());
        return RH_OK;
    }
    const int nseg = cdiv(c->swords, RH_UNPERM_SEG_WORDS);
    const size_t lds = sizeof(uint64_t) * (size_t)std::min<int64_t>(c->swords, RH_UNPERM_SEG_WORDS);
    static bool attr_set = false;
    if (!attr_set) {   // more than the default 64 KB of dynamic LDS
        RH_HIP(hipFuncSetAttribute((const void *)unpermute_masks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(uint64_t) * RH_UNPERM_SEG_WORDS)));
        attr_set = true;
    }
    for (int32_t r0 = 0; r0 < b; r0 += 65535 * 32) {   // grid.x limit (2^31 - 1) is far away; keep launches bounded anyway
        const int32_t rows = std::min<int32_t>(b - r0, 65535 * 32);
        hipLaunchKernelGGL(unpermute_masks_kernel, dim3((unsigned)rows, (unsigned)nseg), dim3(1024), lds, c->stream,
                           d_in + (int64_t)r0 * c->swords, c->sub_perm, c->swords, d_out + (int64_t)r0 * c->swords);
    }
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// exclusive popcount prefix per word (+ total at [nwords]); clobbers c->block_sums / c->d_total
int rhk_word_prefix(rh_cloud *c, const uint64_t *words, int64_t nwords, int32_t *prefix_out)
{
    if (nwords == 0) return RH_OK;
    const int64_t nb = (nwords + RH_WORDS_PER_BLOCK - 1) / RH_WORDS_PER_BLOCK;
    hipLaunchKernelGGL(block_popc_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, words, nwords, c->block_sums);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, c->stream, c->block_sums, nb, c->d_total);
    hipLaunchKernelGGL(expand_mask_kernel, dim3((unsigned)nb), dim3(256), 0, c->stream, words, nwords, c->block_sums,
                       (int64_t *)nullptr, (int64_t)0, prefix_out, (uint64_t *)nullptr, (int32_t *)nullptr);
    RH_HIP(hipMemcpyAsync(prefix_out + nwords, c->d_total, sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    RH_HIP(hipGetLastError());
    c->select_valid = false;
    return RH_OK;
}

// tab[k] = first Morton position whose code is >= k << shift (k = entries - 1: n): the cell directory of the octree
// sampler (fit_shared.h, OctView::cell_bounds)
__global__ void __launch_bounds__(256)
oct_tab_kernel(const uint64_t *__restrict__ code, int64_t n, int shift, int64_t entries, int32_t *__restrict__ tab)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= entries) return;
    int64_t lo = 0, hi = n;
    if (k == entries - 1) lo = n;
    else {
        const uint64_t key = (uint64_t)k << shift;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (code[mid] < key) lo = mid + 1; else hi = mid;
        }
    }
    tab[k] = (int32_t)lo;
}

__global__ void __launch_bounds__(256)
oct_code_o_kernel(const uint64_t *__restrict__ code, const int32_t *__restrict__ perm, int64_t n, uint64_t *__restrict__ code_o)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) code_o[perm[q]] = code[q];
}

int rhk_oct_build_tab(rh_cloud *c)
{
    (void)hipFree(c->oct_tab);
    (void)hipFree(c->oct_code_o);
    c->oct_tab = nullptr;
    c->oct_code_o = nullptr;
    c->oct_tab_level = 0;
    if (c->n == 0) return RH_OK;
    RH_HIP(hipMalloc((void **)&c->oct_code_o, sizeof(uint64_t) * (size_t)c->n));
    hipLaunchKernelGGL(oct_code_o_kernel, dim3(cdiv(c->n, 256)), dim3(256), 0, c->stream, c->oct_code, c->oct_perm, c->n, c->oct_code_o);
    const int level = c->oct_depth < 8 ? c->oct_depth : 8;   // 8^7 + 1 entries at most (8 MB)
    const int64_t entries = ((int64_t)1 << (3 * (level - 1))) + 1;
    RH_HIP(hipMalloc((void **)&c->oct_tab, sizeof(int32_t) * (size_t)entries));
    hipLaunchKernelGGL(oct_tab_kernel, dim3(cdiv(entries, 256)), dim3(256), 0, c->stream, c->oct_code, c->n, 3 * (21 - (level - 1)),
                       entries, c->oct_tab);
    RH_HIP(hipGetLastError());
    c->oct_tab_level = level;
    return RH_OK;
}

int rhk_oct_gather_enabled(rh_cloud *c)
{
    if (c->n == 0) return RH_OK;
    hipLaunchKernelGGL(oct_gather_enabled_kernel, dim3(cdiv(c->nwords * 64, 256)), dim3(256), 0, c->stream, c->enabled,
                       c->oct_perm, c->n, c->oct_men);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_oct_sync_enabled(rh_cloud *c)
{
    if (c->n == 0) return RH_OK;
    RH_TRY(rhk_korder_sync_enabled(c));
    return rhk_word_prefix(c, c->oct_men, c->nwords, c->oct_prefix);
}

// after an extraction (`enabled` already updated in stream order): the Morton-order bits follow -- the culled refit
// scan has cleared them on its way (k_men_valid still set), after the plain scan they are regathered -- and the prefix
// is rebuilt
int rhk_oct_clear_mask(rh_cloud *c, const uint64_t *mask)
{
    (void)mask;
    return rhk_oct_sync_enabled(c);
}
