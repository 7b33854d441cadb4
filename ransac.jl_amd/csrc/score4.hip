// score4.hip -- the v4 batched score kernel: scorecandidates! / scorecandidate (/root/reference/src/fitting.jl:181-190,
// shapes/plane.jl:61-71, sphere.jl:118-134, cylinder.jl:172-183, cone.jl:155-167) for a whole batch in one launch.
//
// Subset 1 in k-d leaf order, 64-point groups with boxes (kernels.hip), a block per tile of RH_G2_TG groups.
//
//  * The scalar unit is the scarce resource (measured, tools/ubench/valu_rates.hip: a scalar instruction costs a SIMD
//    ~4.2 cycles -- one scalar ALU per CU -- as much as a binary64 vector instruction; a binary32 one costs ~2.3).  The
//    older kernel walks (candidate, group) pairs with lane = point: per visit a dozen scalar instructions of loop
//    control, ballots and popcounts around 15 vector ones.  Here a LANE OWNS A PAIR.  Stage 1 (lane = candidate,
//    binary32 box tests against the tile's boxes held in scalar registers) appends the surviving (candidate, group)
//    pairs to the wave's ring in LDS; whenever 64 are queued, stage 2 takes them one per lane and every lane loops
//    over the 64 points of ITS group (LDS reads at per-lane addresses: at most RH_G2_TG distinct rows per
//    instruction, padded apart in the banks) with its candidate's record and its counters in vector registers: no
//    scalar instruction in the loop but its control, no ballot, no reduction, no divergence.
//  * The per-point work is the two-sided binary32 classifier of score4_device.h on a binary32 tile (24 B per point):
//    t = min(a, b) per point with sure <=> t > 0 and maybe <=> t > -1 (plane, sphere, cylinder).  A pair whose "sure"
//    and "maybe" counts differ -- some point lies within the rounding margin of a threshold -- is redone as a whole by
//    the reference's binary64 test (score_device.h, unchanged; lane = point, the points from global memory).  Cones
//    have a binary32 band prefilter only: its survivors go through a per-wave ring to the exact test one by one.
//  * A block's waves never wait for each other between staging and the end of a kind: each walks its own chunks
//    (culling records prefetched one chunk ahead); only the partial last batches of the waves are merged.
//
// Bit-exactness does not depend on the classifier's margins being tight, only on their being upper bounds (proof
// obligations and the audit kernel: score4_device.h).
#include <stdio.h>
#include <stdlib.h>

#include "rh_internal.h"
#include "score_device.h"
#include "score4_device.h"

namespace {

using namespace rhdev;
using namespace rh4;

typedef float rh_f32x4 __attribute__((ext_vector_type(4)));
typedef float rh_f32x2 __attribute__((ext_vector_type(2)));

constexpr int S4_TG = RH_G2_TG;          // groups per tile
constexpr int S4_ROW = 65;               // padded row length of the tile arrays (bank spread of the per-lane rows)
constexpr int S4_W = 4;                  // waves per block
constexpr int S4_RING = 512;             // per-wave pair ring (entries): a batch of 64 and a chunk's 256 survivors fit with room to spare
static_assert(S4_TG == 4, "entry encoding: 2 bits of group");

struct S4Shared {
    rh_f32x4 pa[S4_TG][S4_ROW];          // (x, y, z, nx); zeros for a disabled / out-of-range point
    rh_f32x2 pb[S4_TG][S4_ROW];          // (ny, nz)
    uint64_t len[S4_TG];                 // enabled & valid bits of the groups
    uint32_t ring[S4_W][S4_RING];        // per-wave ring of surviving pairs: candidate slot << 2 | group
    uint32_t left[S4_W * 64];            // the waves' last, partial batches of a kind, merged
    int32_t cntb[S4_W][64];              // cone: per-wave inlier counts of the batch's pairs
    uint16_t qb[S4_W][128];              // cone: per-wave ring (slot-in-batch << 6 | point-in-group) for the exact test
    int weirdw[S4_TG];                   // per staging wave: an enabled point with a non-finite value
    int anyoff[S4_TG];                   // per staging wave: its group has a valid point that is disabled
    int nleft, next_chunk;
};

static __device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static __device__ __forceinline__ bool is_nan_bits(float v)
{
    return (__builtin_bit_cast(uint32_t, v) & 0x7fffffffu) > 0x7f800000u;
}

struct S4KindArgs {
    const rh_cls *cls;      // classifier records of the bin
    const float *box;       // culling records of the bin: field f of slot i at box[f * bstride + i]
    const rh_prep *prep;    // binary64 records of the bin
    const int32_t *orig, *nk;
    const uint64_t *en;
    double eps, cosa;
};
struct S4AllArgs {
    S4KindArgs k[4];
    int64_t ntiles, bstride, ngroups;
    const float *gb32;      // binary32 boxes of the groups, 8 floats each
    unsigned long long *trace;   // debug (RH_S4_TRACE): 16 shader-clock stamps per wave, or null
};

template <int KIND> struct S4Fields { static constexpr int NBOX = KIND == RH_PLANE || KIND == RH_SPHERE ? 5 : (KIND == RH_CYLINDER ? 9 : 10); };

#define RH4_CONST_AS __attribute__((address_space(4)))

// segmented sum over runs of equal keys in adjacent lanes (runs of up to 4: the groups of one candidate): the LAST lane
// of a run gets the run's total, the others 0
static __device__ __forceinline__ int run_total(int v, int key, int lane)
{
    int u = __shfl_up(v, 1), k1 = __shfl_up(key, 1);
    if (lane >= 1 && k1 == key) v += u;
    u = __shfl_up(v, 2); k1 = __shfl_up(key, 2);
    if (lane >= 2 && k1 == key) v += u;
    const int kn = __shfl_down(key, 1);
    return (lane == 63 || kn != key) ? v : 0;
}

// one batch: the 64 pairs at the head of the wave's ring (n of them valid), one per lane
template <int KIND>
static __device__ __forceinline__ void
score4_batch(S4Shared &sh, const int wv, const int lane, const int head, const int n, const double *__restrict__ pts,
             int64_t stride, const int64_t p0, const rh_prep *__restrict__ prep, const rh_cls *__restrict__ cls,
             const int32_t *__restrict__ orig, double eps, double cosa, int32_t *__restrict__ counts, const bool weird)
{
    const bool act = lane < n;
    const uint32_t e = sh.ring[wv][(head + (act ? lane : 0)) & (S4_RING - 1)];
    const int g = (int)(e & 3u), ci = (int)(e >> 2);
    const rh_f32x4 *__restrict__ rowa = &sh.pa[g][0];
    const rh_f32x2 *__restrict__ rowb = &sh.pb[g][0];
    const rh_cls *__restrict__ rec = &cls[ci];
    int total = 0;
    if (KIND != RH_CONE) {
        constexpr int NF = KIND == RH_PLANE ? 9 : (KIND == RH_SPHERE ? 8 : 11);
        rh_cls C;
#pragma unroll
        for (int f = 0; f < NF; f++) C.f[f] = rec->f[f];
        const bool exact_only = weird || is_nan_bits(rec->f[RH_CLS_FLAG]);
        int cs = 0, cm = 0;
#pragma unroll 8
        for (int j = 0; j < 64; j++) {
            const rh_f32x4 a = rowa[j];
            const rh_f32x2 b = rowb[j];
            const float t = KIND == RH_PLANE ? cls_plane_t(C, a.x, a.y, a.z, a.w, b.x, b.y)
                                             : cls_round_t<KIND == RH_PLANE ? RH_SPHERE : KIND>(C, a.x, a.y, a.z, a.w, b.x, b.y);
            cs += t > 0.0f ? 1 : 0;
            cm += t > -1.0f ? 1 : 0;
        }
        const bool amb = act && (cs != cm || exact_only);
        total = (act && !amb) ? cs : 0;
        // pairs the classifier could not decide: the exact test on the whole group, lane = point
        uint64_t redo = WB(amb);
        while (redo != 0) {
            const int k = __builtin_ctzll(redo);
            redo &= redo - 1;
            const uint32_t ek = __builtin_amdgcn_readlane(e, k);
            const int g2 = (int)(ek & 3u), ci2 = (int)(ek >> 2);
            const rh_prep P = rh_ld_prep_const(&prep[ci2]);
            const int64_t gi = p0 + g2 * 64 + lane;
            const uint64_t mres = test_point<KIND>(P, pts[gi], pts[stride + gi], pts[2 * stride + gi], pts[3 * stride + gi],
                                                   pts[4 * stride + gi], pts[5 * stride + gi], eps, cosa) & sh.len[g2];
            if (lane == k) total = __popcll(mres);
        }
    } else {
        rh_cls C;
#pragma unroll
        for (int f = 0; f < 10; f++) C.f[f] = rec->f[f];
        const bool exact_only = weird || is_nan_bits(rec->f[RH_CLS_FLAG]);
        uint32_t mlo = 0, mhi = 0;
#pragma unroll 8
        for (int j = 0; j < 32; j++) {
            const rh_f32x4 a = rowa[j];
            mlo |= cls_pre_cone(C, a.x, a.y, a.z) ? (1u << j) : 0u;
        }
#pragma unroll 8
        for (int j = 0; j < 32; j++) {
            const rh_f32x4 a = rowa[32 + j];
            mhi |= cls_pre_cone(C, a.x, a.y, a.z) ? (1u << j) : 0u;
        }
        const uint64_t lg = sh.len[g];
        uint64_t mask = exact_only ? lg : ((((uint64_t)mhi << 32) | mlo) & lg);
        if (!act) mask = 0;
        sh.cntb[wv][lane] = 0;
        int qbh = 0, qbn = 0;   // ring head / fill (wave-uniform)
        // lane = (pair, point) of the ring: the reference's binary64 test; the point comes from global memory
        auto drain_b = [&](int k) {
            wave_lds_sync();
            const bool on = lane < k;
            const unsigned e2 = on ? sh.qb[wv][(qbh + lane) & 127] : 0u;
            const int slot = (int)(e2 >> 6);
            const uint32_t pe = sh.ring[wv][(head + slot) & (S4_RING - 1)];
            const int64_t gi = p0 + (int)(pe & 3u) * 64 + (int)(e2 & 63u);
            const rh_prep Pv = prep[pe >> 2];
            const uint64_t r = test_point<KIND>(Pv, pts[gi], pts[stride + gi], pts[2 * stride + gi], pts[3 * stride + gi],
                                                pts[4 * stride + gi], pts[5 * stride + gi], eps, cosa);
            if (on && ((r >> lane) & 1ULL)) atomicAdd(&sh.cntb[wv][slot], 1);
            qbh = (qbh + k) & 127;
            qbn -= k;
        };
        // the set bits of the 64 masks, one per lane and round, compacted onto the ring
        while (WB(mask != 0) != 0) {
            const bool has = mask != 0;
            const int j = has ? __builtin_ctzll(mask) : 0;
            mask &= mask - 1;
            const uint64_t mm = WB(has);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
            if (has) sh.qb[wv][(qbh + qbn + rank) & 127] = (uint16_t)((lane << 6) | j);
            qbn += __popcll(mm);
            if (qbn >= 64) drain_b(64);
        }
        if (qbn > 0) drain_b(qbn);
        wave_lds_sync();
        total = sh.cntb[wv][lane];
    }
    // one global atomic per (candidate, tile) with inliers: the pairs of a candidate sit in adjacent lanes
    const int v = run_total(act ? total : 0, act ? ci : -1 - lane, lane);
    if (v != 0) atomicAdd(&counts[orig[ci]], v);
}

// one kind on this block: every wave walks its chunks (box tests -> pair ring -> batches as the ring fills); the
// waves' last, partial batches are merged and shared out again
template <int KIND>
static __device__ __forceinline__ void
score4_kind(S4Shared &sh, const S4KindArgs &K, const rh_box32 (&G)[S4_TG], const int64_t bstride, const double *__restrict__ pts,
            int64_t stride, const int64_t g0, const unsigned live, const bool weird, int32_t *__restrict__ counts, int dbg,
            unsigned long long *tr, int &tslot)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nk = *K.nk;
    const int nch = (nk + 63) >> 6;
    constexpr int NB = S4Fields<KIND>::NBOX;
    int head = 0, fill = 0;   // wave-uniform
    float B[RH_BOX_FIELDS], Bn[RH_BOX_FIELDS];
    // the block's waves take this block's chunks (every gridDim.y-th of the kind) one at a time from a common counter;
    // the NEXT chunk is taken -- and its culling records requested -- before the current one is worked on
    const int rows = (int)gridDim.y, row = (int)blockIdx.y;
    const int nmine = nch > row ? (nch - row + rows - 1) / rows : 0;   // chunks row, row + rows, ... of the kind
    auto grab = [&]() {
        int t = 0;
        if (lane == 0) t = atomicAdd(&sh.next_chunk, 1);
        return __builtin_amdgcn_readfirstlane(t);
    };
    int t = grab();
    if (t < nmine) {
        const int ci = ((row + t * rows) << 6) + lane;
#pragma unroll
        for (int f = 0; f < NB; f++) B[f] = ci < nk ? K.box[(int64_t)f * bstride + ci] : 0.0f;
    }
    while (t < nmine) {
        const int c = row + t * rows;
        const int ci = (c << 6) + lane;
        const int tn = grab();
        if (tn < nmine) {
            const int cin = ((row + tn * rows) << 6) + lane;
#pragma unroll
            for (int f = 0; f < NB; f++) Bn[f] = cin < nk ? K.box[(int64_t)f * bstride + cin] : 0.0f;
        }
        unsigned surv = 0;
#pragma unroll
        for (int g = 0; g < S4_TG; g++) surv |= box_skip32<KIND>(B, G[g]) ? 0u : (1u << g);
        if (dbg == 2) surv = 15u;
        surv &= live;
        if (ci >= nk || dbg == 1) surv = 0;
        // positions candidate-major: all lanes before me, then my own lower groups
        const int k = __popc(surv);
        const uint64_t b0 = WB(k & 1), b1 = WB(k & 2), b2 = WB(k & 4);
        const int tot = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
        if (tot != 0) {
            auto mb = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)); };
            int pos = head + fill + mb(b0) + 2 * mb(b1) + 4 * mb(b2);
#pragma unroll
            for (int g = 0; g < S4_TG; g++) {
                if ((surv >> g) & 1u) {
                    sh.ring[wv][pos & (S4_RING - 1)] = ((uint32_t)ci << 2) | (uint32_t)g;
                    pos++;
                }
            }
            fill += tot;
            if (fill >= 64) wave_lds_sync();
            while (fill >= 64) {
                score4_batch<KIND>(sh, wv, lane, head, 64, pts, stride, g0 * 64, K.prep, K.cls, K.orig, K.eps, K.cosa, counts, weird);
                head = (head + 64) & (S4_RING - 1);
                fill -= 64;
            }
        }
#pragma unroll
        for (int f = 0; f < NB; f++) B[f] = Bn[f];
        t = tn;
    }
    if (tr != nullptr && lane == 0 && tslot < 16) tr[tslot] = __builtin_amdgcn_s_memtime();   // end of the wave's own chunks
    tslot++;
    // the partial batches of the block's waves, merged
    if (fill > 0) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&sh.nleft, fill);
        base = __builtin_amdgcn_readfirstlane(base);
        wave_lds_sync();
        if (lane < fill) sh.left[base + lane] = sh.ring[wv][(head + lane) & (S4_RING - 1)];
    }
    __syncthreads();
    const int nleft = sh.nleft;
    if (wv * 64 < nleft) {
        const int n = min(64, nleft - wv * 64);
        if (lane < n) sh.ring[wv][lane] = sh.left[wv * 64 + lane];
        wave_lds_sync();
        score4_batch<KIND>(sh, wv, lane, 0, n, pts, stride, g0 * 64, K.prep, K.cls, K.orig, K.eps, K.cosa, counts, weird);
    }
    __syncthreads();
    if (threadIdx.x == 0) { sh.nleft = 0; sh.next_chunk = 0; }
}

// the tile as binary32 with the enabled words of one kind applied (the first S4_TG waves: one point per thread)
static __device__ __forceinline__ void s4_stage(S4Shared &sh, const double *__restrict__ pts, int64_t stride, int64_t s,
                                                const uint64_t *__restrict__ enabled_words, const int64_t p0)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wv < S4_TG) {
        const int64_t gi = p0 + tid;
        // (the six loads go out before the enabled word is looked at: one round trip)
        const double x = pts[gi], y = pts[stride + gi], z = pts[2 * stride + gi];
        const double nx = pts[3 * stride + gi], ny = pts[4 * stride + gi], nz = pts[5 * stride + gi];
        const uint64_t vall = valid_mask((gi >> 6) << 6, s);
        uint64_t v = vall;
        if (enabled_words != nullptr && v != 0) v &= enabled_words[gi >> 6];
        const bool on = (v >> (gi & 63)) & 1ULL;
        rh_f32x4 a = { 0.f, 0.f, 0.f, 0.f };
        rh_f32x2 b = { 0.f, 0.f };
        bool bad = false;
        if (lane == 0) sh.anyoff[wv] = v != vall ? 1 : 0;
        if (on) {
            a.x = (float)x; a.y = (float)y; a.z = (float)z; a.w = (float)nx; b.x = (float)ny; b.y = (float)nz;
            const float sum = (fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w)) + (fabsf(b.x) + fabsf(b.y));
            bad = !(sum < __builtin_inff());   // an infinite or NaN value (binary32 overflow included)
        }
        sh.pa[wv][lane] = a;
        sh.pb[wv][lane] = b;
        const uint64_t wb = WB(bad);
        if (lane == 0) { sh.weirdw[wv] = wb != 0 ? 1 : 0; sh.len[wv] = v; }   // wave w stages group w: v is its word
    }
    if (tid == 0) { sh.nleft = 0; sh.next_chunk = 0; }
}

// grid: (tiles padded to a multiple of 8, rows).  The block stages its tile (again where a kind's enabled words
// differ: the reference's sphere scorer ignores isenabled, sphere.jl:121,131 -- spheres run last for that) and runs the
// kinds one after the other, the expensive ones first.
template <int WAVES>
__global__ void __launch_bounds__(64 * S4_W, WAVES == 8 ? 8 : 1)
score4_kernel(const double *__restrict__ pts, int64_t stride, int64_t s, const S4AllArgs A, int32_t *__restrict__ counts, int dbg)
{
    __shared__ S4Shared sh;
    const int64_t tile = blockIdx.x;
    if (tile >= A.ntiles) return;
    const int64_t g0 = tile * S4_TG;
    // the boxes of the tile's groups: wave-uniform, in scalar registers for the whole block
    rh_box32 G[S4_TG];
    {
        const RH4_CONST_AS float *gq = (const RH4_CONST_AS float *)(uintptr_t)A.gb32;
#pragma unroll
        for (int g = 0; g < S4_TG; g++) {
            const int64_t gg = g0 + g < A.ngroups ? g0 + g : A.ngroups - 1;
            const RH4_CONST_AS float *q = gq + gg * 8;
            G[g].cx = q[0]; G[g].cy = q[1]; G[g].cz = q[2]; G[g].hx = q[3]; G[g].hy = q[4]; G[g].hz = q[5]; G[g].hr = q[6];
        }
    }
    const uint64_t *cur_en = nullptr;
    bool staged = false, weird = false, anyoff = true;
    unsigned live = 0;
    int tslot = 0;
    unsigned long long *tr = A.trace ? A.trace + ((int64_t)((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * S4_W + (threadIdx.x >> 6)) * 16 : nullptr;
#define RH_S4_STAMP() do { if (tr != nullptr && (threadIdx.x & 63) == 0 && tslot < 16) tr[tslot] = __builtin_amdgcn_s_memtime(); tslot++; } while (0)
    RH_S4_STAMP();
#define RH_S4_KIND(K)                                                                                                  \
    if (*A.k[K].nk > 0) {                                                                                              \
        /* (a tile without a disabled point looks the same under every kind's enabled words) */                      \
        if (!staged || (A.k[K].en != cur_en && (anyoff || cur_en == nullptr))) {                                      \
            if (staged) __syncthreads();                                                                               \
            s4_stage(sh, pts, stride, s, A.k[K].en, g0 * 64);                                                          \
            cur_en = A.k[K].en;                                                                                        \
            staged = true;                                                                                             \
            __syncthreads();                                                                                           \
            live = 0;                                                                                                  \
            for (int g = 0; g < S4_TG; g++) live |= sh.len[g] != 0 ? (1u << g) : 0u;                                   \
            live = __builtin_amdgcn_readfirstlane(live);                                                               \
            weird = __builtin_amdgcn_readfirstlane(sh.weirdw[0] | sh.weirdw[1] | sh.weirdw[2] | sh.weirdw[3]) != 0;    \
            anyoff = __builtin_amdgcn_readfirstlane(sh.anyoff[0] | sh.anyoff[1] | sh.anyoff[2] | sh.anyoff[3]) != 0;    \
        }                                                                                                              \
        RH_S4_STAMP();                                                                                                 \
        if (live != 0) score4_kind<K>(sh, A.k[K], G, A.bstride, pts, stride, g0, live, weird, counts, dbg, tr, tslot);  \
        RH_S4_STAMP();                                                                                                 \
    }
    RH_S4_KIND(RH_CONE)
    RH_S4_KIND(RH_CYLINDER)
    RH_S4_KIND(RH_PLANE)
    RH_S4_KIND(RH_SPHERE)
#undef RH_S4_KIND
}

// binary32 boxes of the groups from the binary64 ones (7 planes of gstride doubles): 8 floats per group
__global__ void gb32_kernel(const double *__restrict__ gb, int64_t gstride, int64_t ngroups, float *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const double c[3] = { gb[g], gb[gstride + g], gb[2 * gstride + g] };
    const double h[3] = { gb[3 * gstride + g], gb[4 * gstride + g], gb[5 * gstride + g] };
    float o[8];
    box_to_f32(c, h, o);
#pragma unroll
    for (int k = 0; k < 8; k++) out[g * 8 + k] = o[k];
}

inline int cdiv4(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace

int rhk_gb32_build(rh_cloud *c)
{
    if (c->ngroups == 0 || c->gb32 == nullptr) return RH_OK;
    hipLaunchKernelGGL(gb32_kernel, dim3(cdiv4(c->ngroups, 256)), dim3(256), 0, c->stream, c->gb, c->ng_pad, c->ngroups, c->gb32);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// the v4 launch: cls[k] / box[k] = the classifier / culling records of bin prep[k], slot for slot, made for eps / cosa.
int rhk_score4_all(rh_cloud *c, const uint64_t *const en[4], const rh_prep *const prep[4], const void *const cls[4],
                   const float *const box[4], int64_t bstride, const int32_t *const orig[4], const int32_t *const nk[4],
                   int32_t nk_total_bound, const double eps[4], const double cosa[4], int32_t *d_counts)
{
    const int64_t ntiles = (c->ngroups + S4_TG - 1) / S4_TG;
    const int nchunks = cdiv4(nk_total_bound, 64) + 3;   // every bin may end in a partial chunk
    if (ntiles == 0 || nk_total_bound <= 0) return RH_OK;
    static int dbg = -1, env_swz = -1, env_rows = -1;
    if (dbg < 0) { const char *e = getenv("RH_G2_DBG"); dbg = e ? atoi(e) : 0; }
    if (env_swz < 0) { const char *e = getenv("RH_G2_XCD"); env_swz = e ? atoi(e) : 1; }
    if (env_rows < 0) { const char *e = getenv("RH_S4_ROWS"); env_rows = e ? atoi(e) : 0; }
    S4AllArgs A;
    for (int k = 0; k < 4; k++)
        A.k[k] = { (const rh_cls *)cls[k], box[k], prep[k], orig[k], nk[k], en[k], eps[k], cosa[k] };
    A.ntiles = ntiles;
    A.bstride = bstride;
    A.ngroups = c->ngroups;
    A.gb32 = c->gb32;
    A.trace = nullptr;
    static int env_trace = -1;
    if (env_trace < 0) { const char *e = getenv("RH_S4_TRACE"); env_trace = e ? atoi(e) : 0; }
    // one block per tile while the batch is small; rows of ~16 chunks per wave beyond
    int64_t rows = env_rows > 0 ? env_rows : std::max<int64_t>(3, nchunks / (S4_W * 8));
    if (rows > 65535) rows = 65535;
    const bool pad8 = env_swz && ntiles >= 1024;   // XCD-aware grid (kernels.hip)
    dim3 grid((unsigned)(pad8 ? ((ntiles + 7) / 8) * 8 : ntiles), (unsigned)rows);
    static int trace_calls = 0;
    size_t trace_n = 0;
    if (env_trace && ++trace_calls == env_trace) {   // debug: the env_trace-th launch leaves its waves' time stamps in /tmp/rh_s4_trace.bin
        trace_n = (size_t)grid.x * grid.y * S4_W * 16;
        RH_HIP(hipMalloc((void **)&A.trace, trace_n * 8));
        RH_HIP(hipMemsetAsync(A.trace, 0, trace_n * 8, c->stream));
    }
    hipLaunchKernelGGL((score4_kernel<8>), grid, dim3(64 * S4_W), 0, c->stream, c->sub, c->s_pad, c->s, A, d_counts, dbg);
    if (A.trace != nullptr) {
        std::vector<unsigned long long> h(trace_n);
        RH_HIP(hipMemcpyAsync(h.data(), A.trace, trace_n * 8, hipMemcpyDeviceToHost, c->stream));
        RH_HIP(hipStreamSynchronize(c->stream));
        const char *path = getenv("RH_S4_TRACE_FILE");
        FILE *f = fopen(path ? path : "/tmp/rh_s4_trace.bin", "wb");
        if (f) { fwrite(h.data(), 8, trace_n, f); fclose(f); }
        (void)hipFree(A.trace);
    }
    RH_HIP(hipGetLastError());
    return RH_OK;
}
