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
//    are classified the same way (the closed form of project2cone's frame, score4_device.h); their undecided POINTS go
//    through a per-wave ring to the exact test one by one (a redo of the whole group costs 247 binary64 instructions).
//  * A block's waves never wait for each other between staging and the end of a kind: each walks its own chunks
//    (culling records prefetched one chunk ahead); only the partial last batches of the waves are merged.
//
// Bit-exactness does not depend on the classifier's margins being tight, only on their being upper bounds (proof
// obligations and the audit kernel: score4_device.h).
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "rh_internal.h"
#ifdef RH_DIAG
#include "ransac_hip_diag.h"
#endif
#include "score_device.h"
#include "score4_device.h"
#include "score_device32.h"

namespace {

using namespace rhdev;
using namespace rh4;
using namespace rhdev32;

typedef float rh_f32x4 __attribute__((ext_vector_type(4)));
typedef float rh_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long rh_u64x2 __attribute__((ext_vector_type(2)));

constexpr int S4_TG = 4;                 // groups per tile (8, measured in round 4 -- the per-chunk work of stage 1 paid once per 8 box tests,
                                         // fuller batches: cfg3 0.0906 -> 0.0959 ms, cfg2 0.078 -> 0.104, cfg5 0.384 -> 0.375; R = 4 with it: worse still)
constexpr int S4_GB = 2;                 // bits of the group in a pair-list entry
constexpr unsigned S4_GM = (1u << S4_GB) - 1u;
constexpr int S4_ROW = 65;               // padded row length of the tile arrays (bank spread of the per-lane rows)
constexpr int S4_W = 4;                  // waves per block
constexpr int S4_STG = 16;               // groups per super-tile (st_cull_kernel): 4 tiles = 1024 points of the k-d leaf order
// R: 64-candidate chunks per block row -- 16 / 12 / 4 / 2 by the size of the grid (rhk_score4_all holds the rule and the
// measurements behind it)
template <int R, bool MASK = false, bool LIST = false>
struct S4Shared {
    static_assert(S4_TG == (1 << S4_GB) && R * 64 * S4_TG <= 65536 && (R % S4_W == 0 || R < S4_W), "entry encoding: S4_GB bits of group, the rest of 16 for the candidate");
    rh_f32x4 pa[S4_TG][S4_ROW];          // (x, y, z, nx); zeros for a disabled / out-of-range point
    rh_f32x2 pb[S4_TG][S4_ROW];          // (ny, nz)
    uint64_t len[S4_TG];                 // enabled & valid bits of the groups
    rh_f32x4 gbox[S4_TG][2];             // the groups' binary32 boxes (cx cy cz hx | hy hz hr 0): stage 1 reads them as broadcasts
    uint16_t plist[R * 64 * S4_TG + 2];  // the block's surviving pairs: candidate of the row << S4_GB | group (+ a dump slot)
    uint16_t ctab[LIST ? R * 64 : 2];    // LIST: the bin slot of every candidate of the row (the row walks its super-tile's list)
    int32_t cntb[S4_W][64];              // cone: per-wave inlier counts of the batch's pairs
    unsigned long long maskb[S4_W][64];  // cone, masks wanted: per-wave inlier words of the batch's pairs
    uint16_t qb[S4_W][128];              // cone: per-wave ring (slot-in-batch << 6 | point-in-group) for the exact test
    int weirdw[S4_TG];                   // per group: an enabled point with a non-finite value
    int npairs, next_batch;
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
    const int32_t *stop;   // chained octree windows: non-zero = the window has ended, nothing to score (else null)
    int row0;              // TAIL launch: first row that the sized launch did not cover
    int rows;              // sized launch: > 0 = a one-dimensional grid of (tiles padded to 8) x rows blocks in XCD order (score4_kernel)
    S4KindArgs k[4];
    int64_t ntiles, bstride, ngroups;
    const float *gb32;      // binary32 boxes of the groups, 8 floats each
    // masks wanted: per candidate (as the caller numbers it) a LIST of its non-zero inlier words in internal (k-d leaf)
    // order -- entries of 16 bytes (word number, word), mstride of them per row at most; occ = one int32 cursor per row
    // (zero on entry)
    uint64_t *masks;
    uint8_t *occ;
    int64_t mstride;
    unsigned long long *stats;   // diag build: event counters of the launch (rh_dbg_s4_stats; layout at S4_STAT), else null
    // LIST launches: per super-tile (S4_STG groups) and kind the bin slots of the candidates whose culling record does not
    // rule the super-tile's box out, ascending (st_cull_kernel), stcap apart, and their numbers
    const uint16_t *stlist;
    const int32_t *stcount;
    int64_t stcap;
};

// ---- event counters (diag build only; tools/isa_account.py multiplies them with the static instruction histogram of the
// kernel's regions): one add per WAVE-level event.  Per kind k at 24 k + ...: 0 segments entered, 1..4 chunk visits of the
// wave's 1st..4th chunk, 5 candidates box-tested, 6 chunk visits without a survivor, 7 surviving pairs, 8 batches, 9 pairs
// taken by the second (lane = point) pass, 10 undecided points queued for the exact test, 11 full ring drains, 12 final drains,
// 13 points the exact test accepted, 14 pairs with a count > 0, 15 cone: compaction rounds.  Global at 96 + ...: 0 waves of
// blocks with a tile, 1 stagings (per wave), 2 re-used stagings, 3 waves whose tile has no enabled point.
#ifdef RH_DIAG
#define S4_STAT(stats, idx, val) do { const unsigned long long v_ = (unsigned long long)(val); if ((stats) != nullptr && (threadIdx.x & 63) == 0) atomicAdd((stats) + (idx), v_); } while (0)
#else
#define S4_STAT(stats, idx, val) do { } while (0)
#endif

template <int KIND> struct S4Fields { static constexpr int NBOX = KIND == RH_PLANE || KIND == RH_SPHERE ? 5 : (KIND == RH_CYLINDER ? 9 : 11); };

#define RH4_CONST_AS __attribute__((address_space(4)))

// lane i <- lane i - 1 (lane 0 <- 0) / lane i <- lane i + 1 (lane 63 <- 0): DPP moves across the whole wave, no trip through LDS
static __device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
static __device__ __forceinline__ int wave_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }

// segmented sum over runs of equal keys in adjacent lanes (runs of up to S4_TG = 4: the groups of one candidate): the
// LAST lane of a run gets the run's total, the others 0
static __device__ __forceinline__ int run_total(int v, int key, int lane)
{
    static_assert(S4_TG <= 4, "two steps cover runs of four");
    // (every move with all lanes active: a DPP move reads nothing from a lane that is switched off)
    const int k1 = wave_shr1(key), u1 = wave_shr1(v);
    v += (lane >= 1 && k1 == key) ? u1 : 0;
    const int k2 = wave_shr1(k1), u2 = wave_shr1(wave_shr1(v));
    v += (lane >= 2 && k2 == key) ? u2 : 0;
    const int kn = wave_shl1(key);
    return (lane == 63 || kn != key) ? v : 0;
}

// one batch: the 64 pairs of the block's list from `head` on (n of them valid), one per lane
template <int KIND, int R, bool MASK, bool F32, bool LIST>
static __device__ __forceinline__ void
score4_batch(S4Shared<R, MASK, LIST> &sh, const int wv, const int lane, const int head, const int n, const int cbase, const double *__restrict__ pts,
             int64_t stride, const int64_t p0, const rh_prep *__restrict__ prep, const rh_cls *__restrict__ cls,
             const int32_t *__restrict__ orig, double eps, double cosa, int32_t *__restrict__ counts, const bool weird,
             uint64_t *__restrict__ masks, uint8_t *__restrict__ occ, const int64_t mstride, const int64_t g0, unsigned long long *__restrict__ stats)
{
    (void)stats;
    S4_STAT(stats, 24 * KIND + 8, 1);
    const bool act = lane < n;
    const uint32_t e = sh.plist[head + (act ? lane : 0)];
    const int g = (int)(e & S4_GM), ci = LIST ? (int)sh.ctab[e >> S4_GB] : cbase + (int)(e >> S4_GB);
    const rh_f32x4 *__restrict__ rowa = &sh.pa[g][0];
    const rh_f32x2 *__restrict__ rowb = &sh.pb[g][0];
    const rh_cls *__restrict__ rec = &cls[ci];
    int total = 0;
    uint64_t word = 0;   // MASK: the inlier word of the lane's pair
    // Undecided POINTS go to the exact test through a per-wave ring of (pair of the batch << 6 | point of its group), 64 at a
    // time, lane = ring entry: the reference's test of the cloud's element type on the entry's own record; what it accepts is
    // added to the pair's count (and word) in LDS.  (Float32 cloud: the staged floats are the points; Float64: global memory.)
    int qbh = 0, qbn = 0;   // ring head / fill (wave-uniform)
    auto drain_b = [&](int k) {
        wave_lds_sync();
        const bool on = lane < k;
        const unsigned e2 = on ? sh.qb[wv][(qbh + lane) & 127] : 0u;
        const int slot = (int)(e2 >> 6);
        const uint32_t pe = sh.plist[head + slot];
        const int64_t gi = p0 + (int)(pe & S4_GM) * 64 + (int)(e2 & 63u);
        uint64_t r;
        if (F32) {
            const rh_prepf Pv = prepf_of<KIND>(prep[LIST ? (int)sh.ctab[pe >> S4_GB] : cbase + (int)(pe >> S4_GB)]);
            const rh_f32x4 a = sh.pa[pe & S4_GM][e2 & 63u];
            const rh_f32x2 b = sh.pb[pe & S4_GM][e2 & 63u];
            r = test_point32<KIND>(Pv, a.x, a.y, a.z, a.w, b.x, b.y, eps, cosa);
        } else {
            const rh_prep Pv = prep[LIST ? (int)sh.ctab[pe >> S4_GB] : cbase + (int)(pe >> S4_GB)];
            r = test_point<KIND>(Pv, pts[gi], pts[stride + gi], pts[2 * stride + gi], pts[3 * stride + gi], pts[4 * stride + gi],
                                 pts[5 * stride + gi], eps, cosa);
        }
        S4_STAT(stats, 24 * KIND + (k == 64 ? 11 : 12), 1);
        S4_STAT(stats, 24 * KIND + 13, __popcll(r & (k >= 64 ? ~0ULL : ((1ULL << k) - 1ULL))));
        if (on && ((r >> lane) & 1ULL)) {
            atomicAdd(&sh.cntb[wv][slot], 1);
            if (MASK) atomicOr(&sh.maskb[wv][slot], 1ULL << (e2 & 63u));
        }
        qbh = (qbh + k) & 127;
        qbn -= k;
    };
    if (KIND != RH_CONE) {
        constexpr int NF = KIND == RH_PLANE ? 9 : (KIND == RH_SPHERE ? 8 : 11);
        rh_cls C;
#pragma unroll
        for (int f = 0; f < NF; f++) C.f[f] = rec->f[f];
        const bool exact_only = weird || is_nan_bits(rec->f[RH_CLS_FLAG]);
        // The books of the pair over its 64 points (score4_device.h, cls_plane_u): the sign bit of a point's u -- surely an
        // inlier -- is shifted into the pair's word from the right (un-reversed afterwards), and the unsigned minimum of the
        // u's bit patterns ends <= bits(1.0f) iff some point was undecided.  No compare, no carry chain: v_alignbit_b32 +
        // half a v_min3_u32 per point.
        uint32_t wlo = 0, whi = 0, umin = 0xffffffffu;
        auto point_u = [&](int j) {
            const rh_f32x4 a = rowa[j];
            const rh_f32x2 b = rowb[j];
            const float u = KIND == RH_PLANE ? cls_plane_u(C, a.x, a.y, a.z, a.w, b.x, b.y)
                                             : cls_round_u<KIND == RH_PLANE ? RH_SPHERE : KIND>(C, a.x, a.y, a.z, a.w, b.x, b.y);
            const uint32_t ub = __builtin_bit_cast(uint32_t, u);
            umin = umin < ub ? umin : ub;
            return ub;
        };
#pragma unroll 8
        for (int j = 0; j < 32; j++) wlo = __builtin_amdgcn_alignbit(wlo, point_u(j), 31);
#pragma unroll 8
        for (int j = 32; j < 64; j++) whi = __builtin_amdgcn_alignbit(whi, point_u(j), 31);
        if (MASK) { wlo = __brev(wlo); whi = __brev(whi); }
        const int cs = __popc(wlo) + __popc(whi);
        // (u = 1 exactly -- surely outside by the margins -- counts as undecided too: the redo is exact either way; a NaN u
        // can only come from a non-finite record or point, and those never get here: exact_only, weird)
        const bool amb = act && (umin <= 0x3f800000u || exact_only);
        total = (act && !amb) ? cs : 0;
        word = (act && !amb) ? (((uint64_t)whi << 32) | wlo) : 0ULL;
        // Pairs with an undecided point (a percent of them): the classifier once more with lane = POINT (the candidate's record
        // through scalar loads, the points from LDS), which says WHICH points are undecided -- only those take the exact test,
        // 64 at a time through the ring, whatever pair they belong to.  (Round 3 ran the exact test on the whole group of every
        // such pair: 40 binary64 instructions per wave and pair, a tenth of the launch's issue cycles.)
        uint64_t redo = WB(amb);
        S4_STAT(stats, 24 * KIND + 9, __popcll(redo));
        if (redo != 0) {
            sh.cntb[wv][lane] = 0;
            if (MASK) sh.maskb[wv][lane] = 0ULL;
            while (redo != 0) {
                const int k = __builtin_ctzll(redo);
                redo &= redo - 1;
                const uint32_t ek = __builtin_amdgcn_readlane(e, k);
                const int g2 = (int)(ek & S4_GM);
                const int ci2 = LIST ? __builtin_amdgcn_readfirstlane((int)sh.ctab[ek >> S4_GB]) : cbase + (int)(ek >> S4_GB);
                const RH4_CONST_AS rh_cls *rk = (const RH4_CONST_AS rh_cls *)(uintptr_t)&cls[ci2];   // (wave-uniform: scalar loads)
                rh_cls Ck;
#pragma unroll
                for (int f = 0; f < NF; f++) Ck.f[f] = rk->f[f];
                const bool xo = weird || is_nan_bits(rk->f[RH_CLS_FLAG]);
                const rh_f32x4 a = sh.pa[g2][lane];
                const rh_f32x2 b = sh.pb[g2][lane];
                const float u = KIND == RH_PLANE ? cls_plane_u(Ck, a.x, a.y, a.z, a.w, b.x, b.y)
                                                 : cls_round_u<KIND == RH_PLANE ? RH_SPHERE : KIND>(Ck, a.x, a.y, a.z, a.w, b.x, b.y);
                const uint64_t lg = sh.len[g2];
                const uint64_t sure = xo ? 0ULL : (WB(cls_sure(u)) & lg);
                const uint64_t und = xo ? lg : (WB(cls_undecided(u)) & lg);   // (exact-only: every enabled point)
                if (lane == k) { total = __popcll(sure); word = sure; }
                const bool has = (und >> lane) & 1ULL;
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(und >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)und, 0));
                if (has) sh.qb[wv][(qbh + qbn + rank) & 127] = (uint16_t)((k << 6) | lane);
                qbn += __popcll(und);
                S4_STAT(stats, 24 * KIND + 10, __popcll(und));
                if (qbn >= 64) drain_b(64);
            }
            if (qbn > 0) drain_b(qbn);
            wave_lds_sync();
            total += sh.cntb[wv][lane];
            if (MASK) word |= sh.maskb[wv][lane];
        }
    } else {
        rh_cls C;
#pragma unroll
        for (int f = 0; f < 13; f++) C.f[f] = rec->f[f];
        const bool exact_only = weird || is_nan_bits(rec->f[RH_CLS_FLAG]);
        // two words per pair, sign bits shifted in from the right and un-reversed afterwards: s = surely an inlier (u < 0),
        // m = not surely out (u < 1, the sign of u - 1); undecided = m & ~s
        uint32_t slo = 0, shi = 0, mlo = 0, mhi = 0;
#pragma unroll 4
        for (int j = 0; j < 32; j++) {
            const rh_f32x4 a = rowa[j];
            const rh_f32x2 b = rowb[j];
            const float u = cls_cone_u(C, a.x, a.y, a.z, a.w, b.x, b.y);
            slo = __builtin_amdgcn_alignbit(slo, __builtin_bit_cast(uint32_t, u), 31);
            mlo = __builtin_amdgcn_alignbit(mlo, __builtin_bit_cast(uint32_t, u - 1.0f), 31);
        }
#pragma unroll 4
        for (int j = 0; j < 32; j++) {
            const rh_f32x4 a = rowa[32 + j];
            const rh_f32x2 b = rowb[32 + j];
            const float u = cls_cone_u(C, a.x, a.y, a.z, a.w, b.x, b.y);
            shi = __builtin_amdgcn_alignbit(shi, __builtin_bit_cast(uint32_t, u), 31);
            mhi = __builtin_amdgcn_alignbit(mhi, __builtin_bit_cast(uint32_t, u - 1.0f), 31);
        }
        mlo &= ~slo; mhi &= ~shi;
        slo = __brev(slo); shi = __brev(shi); mlo = __brev(mlo); mhi = __brev(mhi);
        const uint64_t lg = sh.len[g];
        // (a disabled point is staged as zeros: whatever the classifier makes of it, its bit is masked out here)
        uint64_t sure = exact_only ? 0ULL : ((((uint64_t)shi << 32) | slo) & lg);
        uint64_t mask = exact_only ? lg : ((((uint64_t)mhi << 32) | mlo) & lg);
        if (!act) { mask = 0; sure = 0; }
        sh.cntb[wv][lane] = 0;
        if (MASK) sh.maskb[wv][lane] = 0ULL;
        // the set bits of the 64 masks, one per lane and round, compacted onto the ring
        while (WB(mask != 0) != 0) {
            const bool has = mask != 0;
            const int j = has ? __builtin_ctzll(mask) : 0;
            mask &= mask - 1;
            const uint64_t mm = WB(has);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
            if (has) sh.qb[wv][(qbh + qbn + rank) & 127] = (uint16_t)((lane << 6) | j);
            qbn += __popcll(mm);
            S4_STAT(stats, 24 * KIND + 15, 1);
            S4_STAT(stats, 24 * KIND + 10, __popcll(mm));
            if (qbn >= 64) drain_b(64);
        }
        if (qbn > 0) drain_b(qbn);
        wave_lds_sync();
        total = sh.cntb[wv][lane] + __popcll(sure);
        if (MASK) word = sh.maskb[wv][lane] | sure;
    }
    if (MASK) {
        // the pairs' inlier words join their candidates' LISTS (any order): entry = (internal word number, word), 16 bytes.  The
        // pairs of a candidate sit in adjacent lanes: the first lane of such a run reserves the run's places with ONE add on
        // the row's cursor.  (Measured, cfg3 launch, against round 3's two plain stores into sparse rows, 94 us: an add per
        // entry at the end of the batch 116; places for every pair reserved at the start of the batch, the round trip behind
        // the point loop, 287 -- the cursors' adds serialise; words parked in LDS and flushed once per segment 128.)
        const bool nz = act && word != 0;
        const uint64_t nzm = WB(nz);
        if (nzm != 0) {
            const int keyp = wave_shr1(act ? ci : -1 - lane);   // (wave-uniform branch: all lanes active)
            const uint64_t starts = WB(lane == 0 || keyp != (act ? ci : -1 - lane));
            const uint64_t below = lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL);          // lanes 0 .. lane
            const int rs = 63 - __builtin_clzll(starts & below);                            // first lane of my run
            const uint64_t after = lane == 63 ? 0ULL : (starts >> (lane + 1));
            const int re = after != 0 ? lane + __builtin_ctzll(after) : 63;               // last lane of my run
            const uint64_t runm = (re == 63 ? ~0ULL : ((2ULL << re) - 1ULL)) & ~((1ULL << rs) - 1ULL);
            const int tot = __popcll(nzm & runm);
            int base = 0;
            int64_t mrow = 0;
            if (nz || (lane == rs && tot > 0)) mrow = orig[ci];
            if (lane == rs && tot > 0) base = atomicAdd((int32_t *)occ + mrow, tot);
            base = __shfl(base, rs);
            if (nz) {
                const int rank = __popcll(nzm & runm & ((1ULL << lane) - 1ULL));
                rh_u64x2 ent;
                ent.x = (uint64_t)(g0 + g);
                ent.y = word;
                ((rh_u64x2 *)masks)[mrow * mstride + base + rank] = ent;
            }
        }
    }
    S4_STAT(stats, 24 * KIND + 14, __popcll(WB(act && total > 0)));
    // one global atomic per (candidate, tile) with inliers: the pairs of a candidate sit in adjacent lanes
    const int v = run_total(act ? total : 0, act ? ci : -1 - lane, lane);
    if (v != 0) atomicAdd(&counts[orig[ci]], v);
}

// one kind segment of the block's row: chunks [lo, hi) of the kind (at most S4_R).  Stage 1: the waves share the
// chunks out, lane = candidate, box tests, survivors -> the block's pair list.  Stage 2: the waves take batches of 64
// pairs from the list until it is empty.
template <int KIND, int R, bool MASK, bool F32, bool LIST>
static __device__ __forceinline__ void
score4_segment(S4Shared<R, MASK, LIST> &sh, const S4KindArgs &K, const int32_t *__restrict__ nkp, const uint16_t *__restrict__ list, const int64_t bstride, const int lo, const int hi,
               const double *__restrict__ pts, int64_t stride, const int64_t g0, const unsigned live, const bool weird,
               int32_t *__restrict__ counts, int dbg, uint64_t *__restrict__ masks, uint8_t *__restrict__ occ, const int64_t mstride,
               unsigned long long *__restrict__ stats)
{
    (void)stats;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nk = *nkp;
    S4_STAT(stats, 24 * KIND + 0, 1);
    constexpr int NB = S4Fields<KIND>::NBOX;
    // (the culling records of the wave's chunks are requested together, before the first box test)
    constexpr int CPW = (R + S4_W - 1) / S4_W;   // (rows of 1 or 2 chunks: the waves without a chunk of their own go straight to the batches)
    float B[2][RH_BOX_FIELDS];
    // (unconditional loads from a clamped slot: a lane without a candidate reads slot 0 and is masked out below -- a
    // guarded load per field costs a scalar exec-mask save / branch / restore each, ~20 scalar instructions per chunk)
    // LIST: the row's candidates come from the super-tile's list -- the bin slots of ALL the wave's chunks are requested
    // together (one round trip in front of the record loads instead of one per chunk) and kept in LDS for stage 2
    int slot[LIST ? CPW : 1];
    auto clamped = [&](int h) { const int ci = ((lo + wv + h * S4_W) << 6) + lane; return (lo + wv + h * S4_W < hi && ci < nk) ? ci : 0; };
    if (LIST) {
#pragma unroll
        for (int h = 0; h < CPW; h++) slot[h] = (int)list[clamped(h)];
#pragma unroll
        for (int h = 0; h < CPW; h++) {
            const int ci = ((lo + wv + h * S4_W) << 6) + lane;
            if (lo + wv + h * S4_W < hi && ci < nk) sh.ctab[ci - (lo << 6)] = (uint16_t)slot[h];
        }
    }
    {
        const int cil = LIST ? slot[0] : clamped(0);
#pragma unroll
        for (int f = 0; f < NB; f++) B[0][f] = K.box[(int64_t)f * bstride + cil];
    }
#pragma unroll
    for (int h = 0; h < CPW; h++) {
        const int c = lo + wv + h * S4_W;
        if (c >= hi) break;
        const int ci = (c << 6) + lane;
        if (h + 1 < CPW) {   // the next chunk's records: in flight during this chunk's tests
            const int cinl = LIST ? slot[h + 1 < CPW ? h + 1 : 0] : clamped(h + 1);
#pragma unroll
            for (int f = 0; f < NB; f++) B[(h + 1) & 1][f] = K.box[(int64_t)f * bstride + cinl];
        }
        unsigned surv = 0;
#pragma unroll
        for (int g = 0; g < S4_TG; g++) {
            // (the boxes come from LDS, wave-uniform addresses: kept in scalar registers across the four per-kind bodies
            // of the kernel they were spilled lane by lane into vector registers -- 30 spilled registers, 217 lane moves)
            const rh_f32x4 g0v = sh.gbox[g][0], g1v = sh.gbox[g][1];
            rh_box32 G;
            G.cx = g0v.x; G.cy = g0v.y; G.cz = g0v.z; G.hx = g0v.w; G.hy = g1v.x; G.hz = g1v.y; G.hr = g1v.z;
            surv |= box_skip32<KIND>(B[h & 1], G) ? 0u : (1u << g);
        }
        if (dbg == 2) surv = (1u << S4_TG) - 1u;
        surv &= live;
        if (ci >= nk || dbg == 1) surv = 0;
        // positions candidate-major: all lanes before me, then my own lower groups
        const int k = __popc(surv);
        const uint64_t b0 = WB(k & 1), b1 = WB(k & 2), b2 = WB(k & 4), b3 = WB(k & 8);
        const int tot = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3);
        S4_STAT(stats, 24 * KIND + 1 + (h < 3 ? h : 3), 1);
        S4_STAT(stats, 24 * KIND + 5, __popcll(WB(ci < nk)));
        S4_STAT(stats, 24 * KIND + 7, tot);
        if (tot == 0) { S4_STAT(stats, 24 * KIND + 6, 1); continue; }
        int base = 0;
        if (lane == 0) base = atomicAdd(&sh.npairs, tot);
        base = __builtin_amdgcn_readfirstlane(base);
        auto mb = [&](uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)); };
        int pos = base + mb(b0) + 2 * mb(b1) + 4 * mb(b2) + 8 * mb(b3);
        // (branch-free: a group that did not survive writes to the dump slot behind the list -- a predicated store costs a
        // scalar exec-mask save / branch / restore per group)
#pragma unroll
        for (int g = 0; g < S4_TG; g++) {
            const bool on = (surv >> g) & 1u;
            sh.plist[on ? pos : R * 64 * S4_TG] = (uint16_t)(((ci - (lo << 6)) << S4_GB) | g);
            pos += on ? 1 : 0;
        }
    }
    __syncthreads();
    const int npairs = sh.npairs;
    for (;;) {
        int bt = 0;
        if (lane == 0) bt = atomicAdd(&sh.next_batch, 1);
        bt = __builtin_amdgcn_readfirstlane(bt);
        if (bt * 64 >= npairs) break;
        score4_batch<KIND, R, MASK, F32, LIST>(sh, wv, lane, bt * 64, min(64, npairs - bt * 64), lo << 6, pts, stride, g0 * 64, K.prep, K.cls, K.orig,
                                         K.eps, K.cosa, counts, weird, masks, occ, mstride, g0, stats);
    }
}

// the tile as binary32 with the enabled words of one kind applied (one point per thread)
template <class SH>
static __device__ __forceinline__ void s4_stage(SH &sh, const double *__restrict__ pts, int64_t stride, int64_t s,
                                                const uint64_t *__restrict__ enabled_words, const int64_t p0, const float *__restrict__ gb32,
                                                const int64_t ngroups)
{
    static_assert(S4_TG % S4_W == 0, "every wave stages S4_TG / S4_W groups");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int GPW = S4_TG / S4_W;
    // (the loads of all the wave's groups go out before any enabled word is looked at: one round trip)
    double x[GPW], y[GPW], z[GPW], nx[GPW], ny[GPW], nz[GPW];
#pragma unroll
    for (int q = 0; q < GPW; q++) {
        const int64_t gi = p0 + (int64_t)(wv + q * S4_W) * 64 + lane;
        x[q] = pts[gi]; y[q] = pts[stride + gi]; z[q] = pts[2 * stride + gi];
        nx[q] = pts[3 * stride + gi]; ny[q] = pts[4 * stride + gi]; nz[q] = pts[5 * stride + gi];
    }
#pragma unroll
    for (int q = 0; q < GPW; q++) {
        const int g = wv + q * S4_W;
        const int64_t gi = p0 + (int64_t)g * 64 + lane;
        uint64_t v = valid_mask((gi >> 6) << 6, s);
        if (enabled_words != nullptr && v != 0) v &= enabled_words[gi >> 6];
        const bool on = (v >> (gi & 63)) & 1ULL;
        rh_f32x4 a = { 0.f, 0.f, 0.f, 0.f };
        rh_f32x2 b = { 0.f, 0.f };
        bool bad = false;
        if (on) {
            a.x = (float)x[q]; a.y = (float)y[q]; a.z = (float)z[q]; a.w = (float)nx[q]; b.x = (float)ny[q]; b.y = (float)nz[q];
            const float sum = (fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w)) + (fabsf(b.x) + fabsf(b.y));
            bad = !(sum < __builtin_inff());   // an infinite or NaN value (binary32 overflow included)
        }
        sh.pa[g][lane] = a;
        sh.pb[g][lane] = b;
        const uint64_t wb = WB(bad);
        if (lane == 0) { sh.weirdw[g] = wb != 0 ? 1 : 0; sh.len[g] = v; }   // v is the group's word
    }
    if (tid == 0) { sh.npairs = 0; sh.next_batch = 0; }
    if (tid < 2 * S4_TG) {   // the boxes, 32 bytes each
        const int64_t g = p0 / 64 + (tid >> 1);
        sh.gbox[tid >> 1][tid & 1] = ((const rh_f32x4 *)gb32)[(g < ngroups ? g : ngroups - 1) * 2 + (tid & 1)];
    }
}

// grid: (tiles padded to a multiple of 8, rows).  The 64-candidate chunks of the four kind bins are laid end to end,
// the expensive kinds first (cone, cylinder, sphere, plane), and cut into rows of R; a block runs the per-kind
// segment(s) of its row (almost always one) on its tile.
// TAIL: the launch that picks up what a sized-before-the-count-was-known launch leaves over (rhk_score4_all, `open`):
// its blocks walk the rows from A.row0 on grid-stride.  A separate instantiation -- the loop around the four per-kind
// bodies costs the register allocation dearly (96 scalar + 40 vector registers spilled), the one-row form none.
template <int R, bool MASK, bool F32, bool TAIL = false, bool LIST = false>
// 7 blocks of four waves per CU = 7 waves per SIMD = 72 vector registers (measured against 8 / 64 registers: cfg3 0.0816 ->
// 0.0807 ms, cfg5 0.3006 -> 0.2973; 6 / 80 registers is slower: profiles/r4/experiments.txt)
#ifndef RH_S4_MINBLK
#define RH_S4_MINBLK 7
#endif
__global__ void __launch_bounds__(64 * S4_W, (F32 && !(LIST && R == 16)) ? 8 : RH_S4_MINBLK)   // (the Float32 instantiations: 8 / 64 registers, 0.0916 -> see experiments.txt; LIST rows of 16: 20.8 KB of LDS, seven blocks)
score4_kernel(const double *__restrict__ pts, int64_t stride, int64_t s, const S4AllArgs A, int32_t *__restrict__ counts, int dbg_arg)
{
#ifdef RH_DIAG
    const int dbg = dbg_arg;   // diag build: 1 = no pair survives stage 1 (the skeleton alone), 2 = every pair does
#else
    constexpr int dbg = 0;     // the product kernel has no such switch
    (void)dbg_arg;
#endif
    __shared__ S4Shared<R, MASK, LIST> sh;
    // Blocks go to the 8 XCDs round-robin by their linear id, each XCD with an L2 of its own.  With rows > 0 the id is
    // read as ((tile / 8) x rows + row) x 8 + tile % 8: the rows of one tile follow each other on ONE XCD, so the tile is
    // fetched from HBM once and staged from L2 by the other rows (cfg5, 9 rows over a 75-MB subset: the (tile, row) grid
    // fetched every tile once per row -- 724 MB per launch, profiles/r3).
    int64_t tile = blockIdx.x;
    int rowy = (int)blockIdx.y;
    if (!TAIL && A.rows > 0) {
        const int64_t k = blockIdx.x >> 3;
        rowy = (int)(k % A.rows);
        tile = ((k / A.rows) << 3) + (blockIdx.x & 7);
    }
    if (tile >= A.ntiles) return;
    if (A.stop != nullptr && *A.stop != 0) return;
    const int64_t g0 = tile * S4_TG;
    S4_STAT(A.stats, 96 + 0, 1);
    static_assert(!(TAIL && LIST), "the row-walking launch takes its candidates in bin order");
    int nch[4], total = 0;
    const int64_t st4 = LIST ? (tile / (S4_STG / S4_TG)) * 4 : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { nch[k] = ((LIST ? A.stcount[st4 + k] : *A.k[k].nk) + 63) >> 6; total += nch[k]; }
    bool ran = false, weird = false;
    unsigned live = 0;
    const uint64_t *staged_en = nullptr;
    (void)staged_en;
#define RH_S4_BODY(K)                                                                                                  \
    {                                                                                                                  \
        const int slo = max(lo, base) - base, shi = min(hi, base + nch[K]) - base;                                     \
        if (slo < shi) {                                                                                               \
            if (ran) __syncthreads();   /* the previous segment's waves are done with the tile and the list */        \
            if (!ran || A.k[K].en != staged_en) {                                                                      \
                s4_stage(sh, pts, stride, s, A.k[K].en, g0 * 64, A.gb32, A.ngroups);                                   \
                staged_en = A.k[K].en;                                                                                 \
                S4_STAT(A.stats, 96 + 1, 1);                                                                           \
            } else { S4_STAT(A.stats, 96 + 2, 1); if (threadIdx.x == 0) { sh.npairs = 0; sh.next_batch = 0; } }   /* same tile, same enabled words */   \
            __syncthreads();                                                                                           \
            live = 0;                                                                                                  \
            for (int g = 0; g < S4_TG; g++) live |= sh.len[g] != 0 ? (1u << g) : 0u;                                   \
            live = __builtin_amdgcn_readfirstlane(live);                                                               \
            { int ww = 0; for (int g = 0; g < S4_TG; g++) ww |= sh.weirdw[g]; weird = __builtin_amdgcn_readfirstlane(ww) != 0; }  \
            if (live != 0) score4_segment<K, R, MASK, F32, LIST>(sh, A.k[K], LIST ? A.stcount + (st4 + K) : A.k[K].nk, LIST ? A.stlist + (st4 + K) * A.stcap : nullptr, A.bstride, slo, shi, pts, stride, g0, live, weird, counts, dbg, A.masks, A.occ, A.mstride, A.stats); \
            else S4_STAT(A.stats, 96 + 3, 1);                                                                         \
            ran = true;                                                                                                \
        }                                                                                                              \
        base += nch[K];                                                                                                \
    }
    if (!TAIL) {
        const int lo = rowy * R, hi = min(total, lo + R);
        if (lo >= hi) return;
        int base = 0;
        RH_S4_BODY(RH_CONE)
        RH_S4_BODY(RH_CYLINDER)
        RH_S4_BODY(RH_SPHERE)
        RH_S4_BODY(RH_PLANE)
    } else {
        for (int lo = (A.row0 + (int)blockIdx.y) * R; lo < total; lo += (int)gridDim.y * R) {
            const int hi = min(total, lo + R);
            int base = 0;
            RH_S4_BODY(RH_CONE)
            RH_S4_BODY(RH_CYLINDER)
            RH_S4_BODY(RH_SPHERE)
            RH_S4_BODY(RH_PLANE)
        }
    }
#undef RH_S4_BODY
}

#ifdef RH_DIAG
// ---- audit of the classifier's margins (tests): every (candidate, point) of a batch against subset 1.
// a32 / b32 are what the score kernel computes (the very same functions: ua, ub of score4_device.h); a64 / b64 the same scaled
// quantities from the reference's binary64 arithmetic.  |x32 - x64| has to stay below 1/2 for the classification to be sound; by
// construction (RH_CLS_SAFETY) it should stay below ~1/8.  out[kind * 2 + {0, 1}] = max over the batch of |a32 - a64|,
// |b32 - b64| (sphere / cylinder: only where the distance half is not far outside, ua64 < 2 -- the norm's error is relative
// to the norm); out[8 + kind] = pairs looked at.
struct S4AuditCand { rh_prep P; rh_cls C; double cNhi, wN, eDlo, wD, cosa; int kind, usable; };

__global__ void __launch_bounds__(256)
cls_audit_kernel(const double *__restrict__ pts, int64_t stride, int64_t s, const S4AuditCand *__restrict__ cands, int ncand,
                 unsigned long long *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (c >= ncand) return;
    const S4AuditCand &Q = cands[c];
    double ea = 0.0, eb = 0.0;
    bool counted = false;
    if (i < s && Q.usable) {
        const double x = pts[i], y = pts[stride + i], z = pts[2 * stride + i];
        const double nx = pts[3 * stride + i], ny = pts[4 * stride + i], nz = pts[5 * stride + i];
        const rh_prep &P = Q.P;
        float a32, b32;
        double a64, b64;
        if (Q.kind == RH_PLANE) {
            cls_plane_ab(Q.C, (float)x, (float)y, (float)z, (float)nx, (float)ny, (float)nz, a32, b32);
            const double dn = (P.f[3] * nx + P.f[4] * ny) + P.f[5] * nz;
            const double d = (P.f[6] * (x - P.f[0]) + P.f[7] * (y - P.f[1])) + P.f[8] * (z - P.f[2]);
            a64 = (Q.cNhi - dn) / Q.wN;
            b64 = (fabs(d) - Q.eDlo) / Q.wD;
            ea = fabs((double)a32 - a64);
            eb = fabs((double)b32 - b64);
            counted = true;
        } else if (Q.kind == RH_SPHERE || Q.kind == RH_CYLINDER) {
            double qx, qy, qz, R, sgn;
            if (Q.kind == RH_SPHERE) {
                cls_round_ab<RH_SPHERE>(Q.C, (float)x, (float)y, (float)z, (float)nx, (float)ny, (float)nz, a32, b32);
                qx = x - P.f[0]; qy = y - P.f[1]; qz = z - P.f[2]; R = P.f[3]; sgn = P.f[4];
            } else {
                cls_round_ab<RH_CYLINDER>(Q.C, (float)x, (float)y, (float)z, (float)nx, (float)ny, (float)nz, a32, b32);
                const double tx = x - P.f[3], ty = y - P.f[4], tz = z - P.f[5];
                const double sd = (P.f[0] * tx + P.f[1] * ty) + P.f[2] * tz;
                qx = (x - P.f[0] * sd) - P.f[3]; qy = (y - P.f[1] * sd) - P.f[4]; qz = (z - P.f[2] * sd) - P.f[5];
                R = P.f[6]; sgn = P.f[7];
            }
            const double nr = sqrt((qx * qx + qy * qy) + qz * qz);
            const double inv = 1.0 / nr;
            const double dt = ((inv * qx) * nx + (inv * qy) * ny) + (inv * qz) * nz;
            a64 = (fabs(nr - R) - Q.eDlo) / Q.wD;
            b64 = (Q.cNhi - sgn * dt) / Q.wN;
            // the norm's error is relative (4.5 u nr): far outside the band it exceeds any fixed margin and cannot matter
            // (ua is a few thousand widths above 1 there); what has to hold is the bound NEAR the band
            if (a64 < 2.0) { ea = fabs((double)a32 - a64); eb = fabs((double)b32 - b64); }
            counted = true;
        } else if (Q.kind == RH_CONE) {
            // a64 / b64 from the reference's own frame (cone_frame: its dist and its normal cosine), the closed form only
            // supplies rho for the multiplied-through angle test; points the classifier hands to the exact test because
            // they lie next to the axis are not looked at
            bool near_axis;
            cls_cone_ab(Q.C, (float)x, (float)y, (float)z, (float)nx, (float)ny, (float)nz, a32, b32, near_axis);
            if (!near_axis) {
                double dist, dt;
                cone_frame(P, x, y, z, nx, ny, nz, dist, dt);
                const double an = sqrt((P.f[3] * P.f[3] + P.f[4] * P.f[4]) + P.f[5] * P.f[5]);
                const double wx = x - P.f[0], wy = y - P.f[1], wz = z - P.f[2];
                const double h = ((P.f[3] * wx + P.f[4] * wy) + P.f[5] * wz) / an;
                const double qx = wx - h * (P.f[3] / an), qy = wy - h * (P.f[4] / an), qz = wz - h * (P.f[5] / an);
                const double rho = sqrt((qx * qx + qy * qy) + qz * qz);
                a64 = (fabs(dist) - Q.eDlo) / Q.wD;
                b64 = 0.5 - rho * (P.f[8] * dt - Q.cosa) / Q.wN;
                ea = fabs((double)a32 - a64);
                eb = fabs((double)b32 - b64);
                counted = true;
            }
        }
        if (!(ea == ea)) ea = 1e30;   // a NaN on one side only is a failure
        if (!(eb == eb)) eb = 1e30;
    }
    // block maximum, then one atomic per block (non-negative doubles order like their bit patterns)
    __shared__ unsigned long long sa[4], sb[4];
    __shared__ int sc[4];
    unsigned long long ua = __builtin_bit_cast(unsigned long long, ea), ub = __builtin_bit_cast(unsigned long long, eb);
    int cnt = counted ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long va = __shfl_down(ua, off), vb = __shfl_down(ub, off);
        ua = va > ua ? va : ua;
        ub = vb > ub ? vb : ub;
        cnt += __shfl_down(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = ua; sb[threadIdx.x >> 6] = ub; sc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0 && Q.kind >= 0 && Q.kind < 4) {
        unsigned long long ma = 0, mb = 0;
        int n = 0;
        for (int w = 0; w < 4; w++) { ma = sa[w] > ma ? sa[w] : ma; mb = sb[w] > mb ? sb[w] : mb; n += sc[w]; }
        atomicMax(&out[Q.kind * 2], ma);
        atomicMax(&out[Q.kind * 2 + 1], mb);
        atomicAdd(&out[8 + Q.kind], (unsigned long long)n);
    }
}

// ---- audit of the DECISIONS (tests, tools/fuzz_score.py): what the margin audit above argues, counted.  One wave per
// (candidate, 64-point group of subset 1), lane = point, with the very records (cls_make) and functions (box_skip32,
// cls_*_t) the score kernel uses, against the exact test of the cloud's element type:
//   out[kind * 10 + 0] pairs, 1 pairs the box test skips, 2 VIOLATION: skipped pairs with an exact inlier,
//   3 points, 4 classified surely-in, 5 surely-out, 6 VIOLATION: surely-in but the exact test rejects,
//   7 VIOLATION: surely-out but the exact test accepts, 8 exact inliers, 9 VIOLATION: the all-zero point a disabled point
//   is staged as comes out surely-in (plane / sphere / cylinder count what the classifier says without looking at the
//   enabled bits; the cone masks them).
struct S4SoundCand { rh_prep P; rh_prepf Pf; rh_cls C; float box[RH_BOX_FIELDS]; int kind; int pad; };
struct S4SoundArgs { double eps[4], cosa[4]; int f32; };

template <int KIND>
static __device__ __forceinline__ void sound_one(const S4SoundCand &Q, const S4SoundArgs &A, double x, double y, double z, double nx, double ny,
                                                 double nz, const uint64_t valid, const rh_box32 &G, const rh_box32 &GS, const int lane, unsigned long long *__restrict__ out)
{
    const double eps = A.eps[KIND], cosa = A.cosa[KIND];
    uint64_t ex;
    if (A.f32) ex = test_point32<KIND>(Q.Pf, (float)x, (float)y, (float)z, (float)nx, (float)ny, (float)nz, eps, cosa);
    else ex = test_point<KIND>(Q.P, x, y, z, nx, ny, nz, eps, cosa);
    ex &= valid;
    const bool skip = box_skip32<KIND>(Q.box, G);
    const bool stskip = box_skip32<KIND>(Q.box, GS);   // the super-tile's box (st_cull_kernel): out[48 + k] such pairs, out[52 + k] VIOLATION: with an exact inlier
    const bool exact_only = is_nan_bits(Q.C.f[RH_CLS_FLAG]);
    const float fx = (float)x, fy = (float)y, fz = (float)z, fnx = (float)nx, fny = (float)ny, fnz = (float)nz;
    float t, t0;   // (the u of score4_device.h: sign bit = surely in, 0 <= u <= 1 undecided)
    if (KIND == RH_PLANE) { t = cls_plane_u(Q.C, fx, fy, fz, fnx, fny, fnz); t0 = cls_plane_u(Q.C, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f); }
    else if (KIND == RH_CONE) { t = cls_cone_u(Q.C, fx, fy, fz, fnx, fny, fnz); t0 = 0.5f; }
    else { t = cls_round_u<KIND == RH_SPHERE ? RH_SPHERE : RH_CYLINDER>(Q.C, fx, fy, fz, fnx, fny, fnz);
           t0 = cls_round_u<KIND == RH_SPHERE ? RH_SPHERE : RH_CYLINDER>(Q.C, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f); }
    const float sum = (fabsf(fx) + fabsf(fy)) + (fabsf(fz) + fabsf(fnx)) + (fabsf(fny) + fabsf(fnz));
    const bool weird = WB(!(sum < __builtin_inff()) && ((valid >> lane) & 1ULL)) != 0;   // the kernel: every candidate exact-only on this tile
    const bool decided = !exact_only && !weird;
    // (the cone's per-point undecided bit is the sign of u - 1, not-surely-in: u = 1 exactly counts as surely out there)
    const uint64_t sin_ = WB(decided && cls_sure(t)) & valid;
    const uint64_t amb = (WB(!decided || (KIND == RH_CONE ? (!cls_sure(t) && cls_sure(t - 1.0f)) : cls_undecided(t)))) & valid;
    const uint64_t sout = valid & ~sin_ & ~amb;
    // a "band point": its DISTANCE half alone is not surely failed (the normal half aside) -- a pair whose group holds one
    // cannot be decided by any test on the group's position box: the floor of the necessary (candidate, group) work
    float du, dn;
    bool near_axis = false;
    if (KIND == RH_PLANE) cls_plane_ab(Q.C, fx, fy, fz, fnx, fny, fnz, dn, du);
    else if (KIND == RH_CONE) cls_cone_ab(Q.C, fx, fy, fz, fnx, fny, fnz, du, dn, near_axis);
    else cls_round_ab<KIND == RH_SPHERE ? RH_SPHERE : RH_CYLINDER>(Q.C, fx, fy, fz, fnx, fny, fnz, du, dn);
    const uint64_t band = WB(!decided || near_axis || !(du > 1.0f)) & valid;
    if (lane == 0) {
        if (band != 0) atomicAdd(&out[40 + KIND], 1ULL);
        if (ex != 0) atomicAdd(&out[44 + KIND], 1ULL);
        if (stskip) atomicAdd(&out[48 + KIND], 1ULL);
        if (stskip && ex != 0) atomicAdd(&out[52 + KIND], 1ULL);
        unsigned long long *o = out + KIND * 10;
        atomicAdd(&o[0], 1ULL);
        if (skip) atomicAdd(&o[1], 1ULL);
        if (skip && ex != 0) atomicAdd(&o[2], 1ULL);
        atomicAdd(&o[3], (unsigned long long)__popcll(valid));
        if (sin_) atomicAdd(&o[4], (unsigned long long)__popcll(sin_));
        if (sout) atomicAdd(&o[5], (unsigned long long)__popcll(sout));
        if (sin_ & ~ex) atomicAdd(&o[6], (unsigned long long)__popcll(sin_ & ~ex));
        if (sout & ex) atomicAdd(&o[7], (unsigned long long)__popcll(sout & ex));
        if (ex) atomicAdd(&o[8], (unsigned long long)__popcll(ex));
        if (blockIdx.x == 0 && !exact_only && cls_sure(t0)) atomicAdd(&o[9], 1ULL);
    }
}

__global__ void __launch_bounds__(64)
cls_sound_kernel(const double *__restrict__ pts, int64_t stride, int64_t s, const float *__restrict__ gb32, const float *__restrict__ st32, const S4SoundCand *__restrict__ cands,
                 int ncand, const S4SoundArgs A, unsigned long long *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    const int64_t i = g * 64 + lane;
    const uint64_t valid = valid_mask(g * 64, s);
    const bool on = (valid >> lane) & 1ULL;
    const double x = on ? pts[i] : 0.0, y = on ? pts[stride + i] : 0.0, z = on ? pts[2 * stride + i] : 0.0;
    const double nx = on ? pts[3 * stride + i] : 0.0, ny = on ? pts[4 * stride + i] : 0.0, nz = on ? pts[5 * stride + i] : 0.0;
    rh_box32 G;
    G.cx = gb32[g * 8 + 0]; G.cy = gb32[g * 8 + 1]; G.cz = gb32[g * 8 + 2]; G.hx = gb32[g * 8 + 3];
    G.hy = gb32[g * 8 + 4]; G.hz = gb32[g * 8 + 5]; G.hr = gb32[g * 8 + 6];
    rh_box32 GS;
    {
        const float *b = st32 + (g / S4_STG) * 8;
        GS.cx = b[0]; GS.cy = b[1]; GS.cz = b[2]; GS.hx = b[3]; GS.hy = b[4]; GS.hz = b[5]; GS.hr = b[6];
    }
    for (int c = blockIdx.y; c < ncand; c += gridDim.y) {
        const S4SoundCand &Q = cands[c];
        switch (Q.kind) {
        case RH_PLANE: sound_one<RH_PLANE>(Q, A, x, y, z, nx, ny, nz, valid, G, GS, lane, out); break;
        case RH_SPHERE: sound_one<RH_SPHERE>(Q, A, x, y, z, nx, ny, nz, valid, G, GS, lane, out); break;
        case RH_CYLINDER: sound_one<RH_CYLINDER>(Q, A, x, y, z, nx, ny, nz, valid, G, GS, lane, out); break;
        case RH_CONE: sound_one<RH_CONE>(Q, A, x, y, z, nx, ny, nz, valid, G, GS, lane, out); break;
        default: break;
        }
    }
}

#endif   // RH_DIAG

// ---- masks -> subset order, from the entry lists the score kernel left (round 4; round 3 wrote the words into sparse
// internal-order rows with an occupancy byte each, and the un-permutation searched them: 1.0 ms at cfg5, a chain of
// dependent loads per block -- occupancy -> list -> words -> positions -- at two blocks per CU).  One block per (candidate
// row, output segment): the segment of the output row is assembled in LDS and written out once, coalesced.  The block
// streams the row's entries (coalesced 16-byte loads, several per thread in flight), keeps of each word the bits that
// land in ITS segment (segmask[segment][word], made once per cloud; without it -- more than 8 segments -- a range test
// after the look-up), and its waves look the surviving bits' positions up 64 at a time through a per-wave ring.
// a row of up to 8192 words (524 288 subset points) is ONE segment (64 KB of LDS); longer rows are cut into segments of 3072
// words (cfg5, 24 415 words per row, masks step: 8192 -> 0.883 ms, 6144 0.851, 4096 0.848, 3072 0.828, 2560 0.839, 2048 0.847,
// 1536 0.891: smaller segments mean more blocks per CU for a chain of dependent loads, and more passes over the entries)
#ifndef RH_UNP_BLOCK
#define RH_UNP_BLOCK 512      // threads per block of the multi-segment form
#endif
#ifndef RH_UNP_MULTI
#define RH_UNP_MULTI 3072     // words per output segment of rows that need several
#endif
constexpr int S4_UNP_WORDS = 8192, S4_UNP_WORDS_MULTI = RH_UNP_MULTI;
constexpr int S6_UNROLL = 4;

// WAVEWORD (rows of one segment): a wave per entry, a lane per bit -- the word's 64 positions are one coalesced load and
// every set bit lands in the block's segment; four entries in flight per wave (round 3's inner loop, fed from the list)
template <bool WAVEWORD, int BLK = 512>
__global__ void __launch_bounds__(BLK)
unpermute6_kernel(const rh_u64x2 *__restrict__ ent, const int32_t *__restrict__ cursor, int64_t mstride, const int32_t *__restrict__ perm,
                  int64_t swords, int64_t seg_words, const uint64_t *__restrict__ segmask, int64_t smstride, uint64_t *__restrict__ out)
{
    extern __shared__ unsigned long long seg[];
    __shared__ uint32_t ring[BLK / 64][128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = blockIdx.x;
    const int64_t w0 = (int64_t)blockIdx.y * seg_words;
    const int nw = (int)(swords - w0 < seg_words ? swords - w0 : seg_words);
    for (int t = threadIdx.x; t < nw; t += BLK) seg[t] = 0ULL;
    const int n = min(cursor[row], (int32_t)mstride);
    __syncthreads();
    const int32_t lo = (int32_t)(w0 << 6), span = nw << 6;
    const rh_u64x2 *__restrict__ src = ent + row * mstride;
    const uint64_t *__restrict__ sm = segmask != nullptr ? segmask + (int64_t)blockIdx.y * smstride : nullptr;
    if (WAVEWORD) {
        for (int i0 = wv * 4; i0 < n; i0 += (BLK / 64) * 4) {
            rh_u64x2 e[4];
            int32_t j[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                e[k] = src[min(i0 + k, n - 1)];
                j[k] = perm[((int64_t)e[k].x << 6) + lane] - lo;
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i0 + k < n && ((e[k].y >> lane) & 1ULL) && j[k] >= 0 && j[k] < span) atomicOr(&seg[j[k] >> 6], 1ULL << (j[k] & 63));
        }
        __syncthreads();
        uint64_t *__restrict__ dstw = out + row * swords + w0;
        for (int t = threadIdx.x; t < nw; t += BLK) dstw[t] = seg[t];
        return;
    }
    int head = 0, fill = 0;   // the wave's ring (wave-uniform)
    auto drain = [&](int k) {
        wave_lds_sync();
        if (lane < k) {
            const uint32_t e = ring[wv][(head + lane) & 127];
            const int32_t j = perm[e] - lo;
            if (j >= 0 && j < span) atomicOr(&seg[j >> 6], 1ULL << (j & 63));
        }
        head = (head + k) & 127;
        fill -= k;
    };
    for (int i0 = threadIdx.x; i0 - lane < n; i0 += BLK * S6_UNROLL) {   // (whole waves stay in the loop: ballots)
        rh_u64x2 e[S6_UNROLL];
#pragma unroll
        for (int k = 0; k < S6_UNROLL; k++) {
            const int i = i0 + k * BLK;
            e[k].x = 0; e[k].y = 0;
            if (i < n) e[k] = src[i];
        }
        uint64_t m[S6_UNROLL];
#pragma unroll
        for (int k = 0; k < S6_UNROLL; k++) m[k] = (sm != nullptr && e[k].y != 0) ? (e[k].y & sm[e[k].x]) : e[k].y;
#pragma unroll
        for (int k = 0; k < S6_UNROLL; k++) {
            const uint32_t gbase = (uint32_t)e[k].x << 6;
            uint64_t mm = m[k];
            for (;;) {
                const bool has = mm != 0;
                const uint64_t hm = WB(has);
                if (hm == 0) break;
                const int bit = has ? __builtin_ctzll(mm) : 0;
                mm &= mm - 1;
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0));
                if (has) ring[wv][(head + fill + rank) & 127] = gbase + (uint32_t)bit;
                fill += __popcll(hm);
                if (fill >= 64) drain(64);
            }
        }
    }
    if (fill > 0) drain(fill);
    __syncthreads();
    uint64_t *__restrict__ dst = out + row * swords + w0;
    for (int t = threadIdx.x; t < nw; t += BLK) dst[t] = seg[t];
}

__global__ void clear_cursors_kernel(int32_t *__restrict__ cursor, int32_t b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b) cursor[i] = 0;
}

// segmask[seg * smstride + g]: bit b set <=> point b of internal word g has its subset position in output segment seg
__global__ void segmask_kernel(const int32_t *__restrict__ perm, int64_t s, int64_t ngroups, int64_t seg_bits, int nseg, int64_t smstride,
                               uint64_t *__restrict__ segmask)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    for (int sgi = 0; sgi < nseg; sgi++) {
        uint64_t m = 0;
        for (int b = 0; b < 64; b++) {
            const int64_t i = (g << 6) + b;
            if (i < s && (int64_t)perm[i] / seg_bits == sgi) m |= 1ULL << b;
        }
        segmask[(int64_t)sgi * smstride + g] = m;
    }
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

// boxes of the super-tiles: the union of S4_STG consecutive groups' boxes, widened by what the unions' own arithmetic can
// lose, then made binary32 like a group's.  A group box that is not finite makes the super-tile's NaN: never skipped.
__global__ void st32_kernel(const double *__restrict__ gb, int64_t gstride, int64_t ngroups, int64_t nst, float *__restrict__ out)
{
    const int64_t st = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (st >= nst) return;
    double lo[3] = { __builtin_inf(), __builtin_inf(), __builtin_inf() }, hi[3] = { -__builtin_inf(), -__builtin_inf(), -__builtin_inf() };
    bool bad = false;
    for (int64_t g = st * S4_STG; g < (st + 1) * S4_STG && g < ngroups; g++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double cc = gb[k * gstride + g], hh = gb[(3 + k) * gstride + g];
            bad |= !(fabs(cc) < __builtin_inf()) || !(fabs(hh) < __builtin_inf());
            lo[k] = fmin(lo[k], cc - hh);
            hi[k] = fmax(hi[k], cc + hh);
        }
    double c3[3], h3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        c3[k] = 0.5 * lo[k] + 0.5 * hi[k];
        // (cc -+ hh above and the two differences below each round once: a few ulps of the LARGEST magnitude involved, not of h)
        h3[k] = fmax(hi[k] - c3[k], c3[k] - lo[k]) * 1.000000000001 + (fabs(lo[k]) + fabs(hi[k])) * 1e-15;
        bad |= !(fabs(c3[k]) < __builtin_inf()) || !(h3[k] < __builtin_inf());
    }
    float o[8];
    box_to_f32(c3, h3, o);
#pragma unroll
    for (int k = 0; k < 8; k++) out[st * 8 + k] = bad ? __builtin_nanf("") : o[k];
}

// ---- the super-tile lists of a batch.  One block per (super-tile, kind): the kind's candidates in bin order, 256 at a time,
// lane = candidate, the culling record against the super-tile's box with the very test of stage 1 (box_skip32: a box that
// holds all the points of its groups may be skipped only when none of them can pass) -- the survivors' bin slots, in order.
struct StCullArgs {
    const float *box[4];
    const int32_t *nk[4];
    int64_t bstride, stcap;
    const float *st32;
    uint16_t *stlist;
    int32_t *stcount;
};

template <int KIND>
static __device__ __forceinline__ void st_cull_kind(const StCullArgs &A, const int64_t st)
{
    constexpr int Q = 4;                       // candidates per thread and round: 1024 per round, their records requested together
    __shared__ int32_t wsum[2][Q * 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nk = *A.nk[KIND];
    int32_t *cnt = A.stcount + st * 4 + KIND;
    if (nk <= 0) { if (tid == 0) *cnt = 0; return; }
    rh_box32 G;
    {
        const float *b = A.st32 + st * 8;
        G.cx = b[0]; G.cy = b[1]; G.cz = b[2]; G.hx = b[3]; G.hy = b[4]; G.hz = b[5]; G.hr = b[6];
    }
    uint16_t *__restrict__ list = A.stlist + (st * 4 + KIND) * A.stcap;
    constexpr int NB = S4Fields<KIND>::NBOX;
    const float *__restrict__ box = A.box[KIND];
    int base = 0, it = 0;
    for (int j0 = 0; j0 < nk; j0 += Q * 256, it++) {
        float B[Q][RH_BOX_FIELDS];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int cand = j0 + q * 256 + tid, cl = cand < nk ? cand : 0;
#pragma unroll
            for (int f = 0; f < NB; f++) B[q][f] = box[(int64_t)f * A.bstride + cl];
        }
        uint64_t m[Q];
        bool keep[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            keep[q] = j0 + q * 256 + tid < nk && !box_skip32<KIND>(B[q], G);
            m[q] = WB(keep[q]);
            if (lane == 0) wsum[it & 1][q * 4 + wv] = __popcll(m[q]);
        }
        __syncthreads();   // (the sums alternate between two rows: one barrier per round)
        int run = base;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            int off = run;
#pragma unroll
            for (int w = 0; w < 4; w++) { const int v = wsum[it & 1][q * 4 + w]; off += w < wv ? v : 0; run += v; }
            if (keep[q]) list[off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[q], 0))] = (uint16_t)(j0 + q * 256 + tid);
        }
        base = run;
    }
    if (tid == 0) *cnt = base;
}

__global__ void __launch_bounds__(256)
st_cull_kernel(const StCullArgs A)
{
    const int64_t st = blockIdx.x;
    switch (blockIdx.y) {
    case RH_PLANE: st_cull_kind<RH_PLANE>(A, st); break;
    case RH_SPHERE: st_cull_kind<RH_SPHERE>(A, st); break;
    case RH_CYLINDER: st_cull_kind<RH_CYLINDER>(A, st); break;
    default: st_cull_kind<RH_CONE>(A, st); break;
    }
}

inline int cdiv4(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace

int rhk_gb32_build(rh_cloud *c)
{
    if (c->ngroups == 0 || c->gb32 == nullptr) return RH_OK;
    hipLaunchKernelGGL(gb32_kernel, dim3(cdiv4(c->ngroups, 256)), dim3(256), 0, c->stream, c->gb, c->ng_pad, c->ngroups, c->gb32);
    if (c->st32 != nullptr && c->nst > 0)
        hipLaunchKernelGGL(st32_kernel, dim3(cdiv4(c->nst, 64)), dim3(64), 0, c->stream, c->gb, c->ng_pad, c->ngroups, c->nst, c->st32);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// the super-tile lists' workspace of the batch slot in place: [nst][4][cap] bin slots + [nst][4] counts
static int ensure_stlists(rh_cloud *c, int64_t cap)
{
    if (c->d_stlist != nullptr && cap <= c->stlist_cap && c->stlist_nst == c->nst) return RH_OK;
    RH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_stlist); (void)hipFree(c->d_stcount);
    c->d_stlist = nullptr; c->d_stcount = nullptr; c->stlist_cap = 0;
    const int64_t ncap = std::max<int64_t>(cap, 1024);
    RH_HIP(hipMalloc((void **)&c->d_stlist, sizeof(uint16_t) * (size_t)c->nst * 4 * (size_t)ncap));
    RH_HIP(hipMalloc((void **)&c->d_stcount, sizeof(int32_t) * (size_t)c->nst * 4));
    c->stlist_cap = ncap;
    c->stlist_nst = c->nst;
    return RH_OK;
}

// the v4 launch: cls[k] / box[k] = the classifier / culling records of bin prep[k], slot for slot, made for eps / cosa.
int rhk_score4_all(rh_cloud *c, const uint64_t *const en[4], const rh_prep *const prep[4], const void *const cls[4],
                   const float *const box[4], int64_t bstride, const int32_t *const orig[4], const int32_t *const nk[4],
                   int32_t nk_total_bound, const double eps[4], const double cosa[4], int32_t *d_counts, uint64_t *d_masks_int,
                   uint8_t *d_occ, int64_t mstride)
{
    const bool open_count = c->s4_open_count;
    const bool f32cloud = c->f32;   // the exact tests in binary32, on float records derived from `prep`
    // the points: subset 1 in internal order, or the set rhk_score4_dis put in place (a segment of the disabled list)
    rh_s4_points PS = { c->sub, c->s_pad, c->s, c->ngroups, c->gb32 };
    if (c->s4_points != nullptr) PS = *c->s4_points;
    const int64_t ntiles = (PS.ngroups + S4_TG - 1) / S4_TG;
    const int nchunks = cdiv4(nk_total_bound, 64) + 3;   // every bin may end in a partial chunk
    if (ntiles == 0 || nk_total_bound <= 0) return RH_OK;
    const int dbg = (int)rh_opt_int(c, RH_OPT_G2_DBG, 0);   // (diag build only -- 1: no pair survives stage 1, the skeleton alone: tools/region_counters.sh)
    S4AllArgs A;
    for (int k = 0; k < 4; k++)
        A.k[k] = { (const rh_cls *)cls[k], box[k], prep[k], orig[k], nk[k], en[k], eps[k], cosa[k] };
    A.stop = c->s4_stop;
    A.ntiles = ntiles;
    A.bstride = bstride;
    A.ngroups = PS.ngroups;
    A.gb32 = PS.gb32;
    A.masks = d_masks_int;
    A.occ = d_occ;
    A.mstride = mstride;
    A.stats = nullptr;
#ifdef RH_DIAG
    A.stats = (unsigned long long *)c->s4_stats;   // (rh_dbg_s4_stats switched the counters on)
#endif
    // Super-tile lists (round 5): a candidate meets only ~13 / 3 / 10 % of the groups (plane / sphere / cylinder at cfg3), yet
    // every block of every tile walks ALL the candidates of its row through stage 1 -- loads of their culling records, four box
    // tests, the list bookkeeping: a third of the launch's issue cycles.  One small launch first tests every candidate against
    // the boxes of the SUPER-TILES (16 groups = 4 tiles: 1/16 of the tests) and leaves per super-tile and kind the list of
    // the candidates that may meet it (42 / 7 / 35 % of them at cfg3, 24 / 6 / 21 % at cfg5); the rows of a tile are then cut
    // from its super-tile's lists.  Same culling rule, so the same counts and masks: a candidate the super-tile's box rules out
    // has no inlier in any of its groups.  For launches whose candidate count the host knows, on subset 1.
    A.stlist = nullptr; A.stcount = nullptr; A.stcap = 0;
    const int64_t st_opt = rh_opt_int(c, RH_OPT_ST_CULL, 0);   // 0 = by size, 1 = whenever possible, 2 = never
    bool use_lists = !open_count && c->s4_points == nullptr && dbg == 0 && c->st32 != nullptr && c->nst >= 2 && nk_total_bound <= 16384 && st_opt != 2;
    // By size (measured, cfg3's mix on 10M / 20M / 30M / cfg5's on 50M points = 1221 / 2442 / 3662 / 6104 tiles, a batch of 4096): the
    // lists cut the launch's instructions by a fifth, but its blocks become three times fewer and heavier, and a small grid then
    // ends in a long tail (and the list launch sits between the prepare and the score launch) -- one batch at a time +17 % / +0 % /
    // -8 % / -13 %; with two batches in flight, where the next batch fills the tail, -8 % / -15 % / -15 % / -24 %; cfg2's 123
    // tiles +50 % either way.
    if (use_lists && st_opt == 0)
        use_lists = nk_total_bound >= 1024 && ntiles >= (rh_opt_int(c, RH_OPT_BATCHES_IN_FLIGHT, 1) > 1 ? 1000 : 3000);
    if (use_lists) {
        RH_TRY(ensure_stlists(c, ((int64_t)nk_total_bound + 63) / 64 * 64));
        StCullArgs SA;
        for (int k = 0; k < 4; k++) { SA.box[k] = box[k]; SA.nk[k] = nk[k]; }
        SA.bstride = bstride; SA.stcap = c->stlist_cap; SA.st32 = c->st32; SA.stlist = c->d_stlist; SA.stcount = c->d_stcount;
        hipLaunchKernelGGL(st_cull_kernel, dim3((unsigned)c->nst, 4), dim3(256), 0, c->stream, SA);
        A.stlist = c->d_stlist; A.stcount = c->d_stcount; A.stcap = c->stlist_cap;
        if (c->time_cull && c->ev_cull != nullptr) RH_HIP(hipEventRecord(c->ev_cull, c->stream));
    }
    const int env_r = (int)rh_opt_int(c, RH_OPT_S4_ROWS, 0);   // rh_set_option(.., "s4_rows", ..), read on every launch: the fuzzers vary it from case to case
    // R = chunks of 64 candidates per block row.  A block's fixed work -- prologue, staging its tile, the four kinds' dispatch --
    // is a third of the launch's issue cycles at cfg3 and nearly half at cfg5 (profiles/r4/region_counters*.txt): longer rows pay
    // it less often, as long as the grid still has a few blocks per slot -- and as long as the HEAVIEST block does not set the
    // launch's time: on few tiles every candidate of a row meets the same dense tiles, and the time grows linearly with R
    // (round 5, cfg2's 123 tiles: R = 1 / 2 / 4 / 8 / 16 -> 0.0428 / 0.0432 / 0.0588 / 0.0898 / 0.158 ms; the same 123 tiles of a
    // sparse 1M-point scene 0.0383 / 0.0323 / 0.0334 / 0.0442; 367 tiles R = 2 / 4 / 8 / 12 -> 0.0589 / 0.0492 / 0.0576 / 0.0616;
    // cfg3's 1221 tiles 4 / 8 / 12 / 16 -> 0.0969 / 0.0866 / 0.0855 / 0.0873; cfg5's 6104 tiles 8 / 12 / 16 / 24 / 32 -> 0.367 / 0.330 /
    // 0.319 / 0.343 / 0.328).  Open-ended windows keep 4 / 8 (their tail launch walks the same rows).
    int R;
    if (env_r == 4 || env_r == 8 || ((env_r == 1 || env_r == 2 || env_r == 12 || env_r == 16) && !open_count)) R = env_r;
    else if (open_count) R = ntiles * ((nchunks + 7) / 8) < 3000 ? 4 : 8;
    else if (ntiles * ((nchunks + 15) / 16) >= 16384) R = 16;
    else if (ntiles * ((nchunks + 11) / 12) >= 3000) R = 12;
    else if (ntiles * ((nchunks + 3) / 4) >= 4000) R = 4;
    else R = 2;
    int64_t rows = (nchunks + R - 1) / R;
    if (!open_count) { c->last_s4[0] = R; c->last_s4[1] = use_lists ? 1 : 0; c->last_s4[2] = (int32_t)rows; c->last_s4[3] = (int32_t)ntiles; }
    if (rows > 65535) { rh_set_error("batch of %d candidates is too large for one launch", nk_total_bound); return RH_E_INVALID; }
    // XCD-aware grid: the hardware deals consecutive block ids round-robin to the 8 XCDs, each with an L2 of its own; with grid.x
    // padded to a multiple of 8 a tile meets the same XCD in every row (only with >= 128 tiles per XCD: cfg2's 15 per XCD unbalance)
    const bool pad8 = ntiles >= 1024;
    dim3 grid((unsigned)(pad8 ? ((ntiles + 7) / 8) * 8 : ntiles), (unsigned)rows);
    const unsigned tiles_x = grid.x;   // the tail / loop launches keep the (tile, row) grid
    A.row0 = 0;
    A.rows = 0;
    // (measured, round 4: cfg2 -- 123 tiles x 17 rows -- 0.0753 -> 0.0663 ms; cfg3 -- 1221 x 9 -- 0.0927 -> 0.1041: with every row
    // in flight at once the record gathers of stage 2 spread over all 4096 candidates instead of a row's 512; cfg5 0.388
    // either way.  So: small launches only.)
#ifndef RH_S4_XROW_MAX
#define RH_S4_XROW_MAX 4096
#endif
    if (rows > 1 && ((ntiles + 7) / 8) * 8 * rows <= RH_S4_XROW_MAX) {
        A.rows = (int)rows;
        grid = dim3((unsigned)(((ntiles + 7) / 8) * 8 * rows), 1);
    }
    // the candidate loop's windows with few candidates (octree sampling: ~1000 local shapes per iteration, few pairs
    // survive the boxes): one block per tile walks ALL the rows -- the tile is staged once instead of once per row
    // (measured on the cfg3 octree leg, 18 chunks: 27 us against 39 + 7 for rows + tail; from ~40 chunks on the rows win:
    // sweep of the threshold 32 / 40 / 48 / 64 -> 0.0491 / 0.0498 / 0.0498 / 0.0499 s for the leg)
    const int loop_max = open_count ? 32 : 0;
    if (loop_max > 0 && d_masks_int == nullptr && nchunks <= loop_max) {
        dim3 gt(tiles_x, 1);
        if (f32cloud) {
            if (R == 4) hipLaunchKernelGGL((score4_kernel<4, false, true, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
            else hipLaunchKernelGGL((score4_kernel<8, false, true, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
        } else {
            if (R == 4) hipLaunchKernelGGL((score4_kernel<4, false, false, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
            else hipLaunchKernelGGL((score4_kernel<8, false, false, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
        }
        RH_HIP(hipGetLastError());
        return RH_OK;
    }
#define RH_S4_LAUNCH(RR, MM, FF)                                                                                                              \
    do {                                                                                                                                      \
        if (use_lists) hipLaunchKernelGGL((score4_kernel<RR, MM, FF, false, true>), grid, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg); \
        else hipLaunchKernelGGL((score4_kernel<RR, MM, FF>), grid, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);  \
    } while (0)
#define RH_S4_LAUNCH_R(MM, FF)                                                                                         \
    do {                                                                                                               \
        if (R == 4) RH_S4_LAUNCH(4, MM, FF);                                                                           \
        else if (R == 1) RH_S4_LAUNCH(1, MM, FF);                                                                      \
        else if (R == 2) RH_S4_LAUNCH(2, MM, FF);                                                                      \
        else if (R == 12) RH_S4_LAUNCH(12, MM, FF);                                                                    \
        else if (R == 16) RH_S4_LAUNCH(16, MM, FF);                                                                    \
        else RH_S4_LAUNCH(8, MM, FF);                                                                                  \
    } while (0)
    if (f32cloud) {   // Float32 cloud: c->sub (and the disabled list) hold the exactly converted values
        if (d_masks_int != nullptr) RH_S4_LAUNCH_R(true, true); else RH_S4_LAUNCH_R(false, true);
    } else if (d_masks_int != nullptr) RH_S4_LAUNCH_R(true, false);
    else RH_S4_LAUNCH_R(false, false);
#undef RH_S4_LAUNCH_R
#undef RH_S4_LAUNCH
    if (open_count) {
        // nk_total_bound was a guess (a window of the candidate loop is queued before its list length is known): whatever
        // lies beyond the rows above is scored by a second, small launch that walks the remaining rows grid-stride --
        // its blocks return at once when there is nothing (the usual case)
        if (d_masks_int != nullptr) { rh_set_error("rhk_score4_all: masks need an exact candidate count"); return RH_E_INTERNAL; }
        A.row0 = (int)rows;
        dim3 gt(tiles_x, 2);
        if (f32cloud) {
            if (R == 4) hipLaunchKernelGGL((score4_kernel<4, false, true, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
            else hipLaunchKernelGGL((score4_kernel<8, false, true, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
        } else {
            if (R == 4) hipLaunchKernelGGL((score4_kernel<4, false, false, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
            else hipLaunchKernelGGL((score4_kernel<8, false, false, true>), gt, dim3(64 * S4_W), 0, c->stream, PS.pts, PS.stride, PS.s, A, d_counts, dbg);
        }
    }
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// the prepared candidates of a device store (driver.hip) -> classifier + culling records for the v4 kernel, made on the
// fly for a liveness pass: index space = the kinds laid end to end from pbase[q] on, fields of the culling records bstride apart
namespace {
__global__ void __launch_bounds__(256)
store_cls_kernel(const rh_prep *p0, const rh_prep *p1, const rh_prep *p2, const rh_prep *p3, int32_t n0, int32_t n1, int32_t n2, int32_t n3,
                 int32_t b1, int32_t b2, int32_t b3, int32_t total, double e0, double e1, double e2, double e3, double c0, double c1, double c2,
                 double c3, double M, double Nm, rh_cls *__restrict__ cls, float *__restrict__ box, int64_t bstride)
{
    const int32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int q = (g >= b1 ? 1 : 0) + (g >= b2 ? 1 : 0) + (g >= b3 ? 1 : 0);
    const int32_t slot = g - (q == 0 ? 0 : (q == 1 ? b1 : (q == 2 ? b2 : b3)));
    const int32_t n = q == 0 ? n0 : (q == 1 ? n1 : (q == 2 ? n2 : n3));
    if (slot >= n) return;
    const rh_prep *pp = q == 0 ? p0 : (q == 1 ? p1 : (q == 2 ? p2 : p3));
    const double eps = q == 0 ? e0 : (q == 1 ? e1 : (q == 2 ? e2 : e3)), cosa = q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
    const rh_prep P = pp[slot];
    cls_make(P, q, eps, cosa, M, Nm, cls[g], box + g, bstride);
}
}  // namespace

int rhk_store_cls(rh_cloud *c, const rh_prep *const prep[4], const int32_t n[4], const int32_t pbase[5], const double eps[4],
                  const double cosa[4], void *d_cls, float *d_box, int64_t bstride)
{
    if (pbase[4] <= 0) return RH_OK;
    hipLaunchKernelGGL(store_cls_kernel, dim3((unsigned)cdiv4(pbase[4], 256)), dim3(256), 0, c->stream, prep[0], prep[1], prep[2], prep[3], n[0], n[1],
                       n[2], n[3], pbase[1], pbase[2], pbase[3], pbase[4], eps[0], eps[1], eps[2], eps[3], cosa[0], cosa[1], cosa[2], cosa[3],
                       c->coord_mag, c->nrm_mag, (rh_cls *)d_cls, d_box, bstride);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// the v4 kernel on cnt points of the disabled list from `first` on (every one counts: no enabled words)
int rhk_score4_dis(rh_cloud *c, int64_t first, int64_t cnt, const rh_prep *const prep[4], const void *const cls[4], const float *const box[4],
                   int64_t bstride, const int32_t *const orig[4], const int32_t *const nk[4], int32_t nk_total_bound, const double eps[4],
                   const double cosa[4], int32_t *d_counts)
{
    if (cnt <= 0 || nk_total_bound <= 0) return RH_OK;
    const int64_t ng = (cnt + 63) / 64;
    if (ng > c->ng_pad) { rh_set_error("rhk_score4_dis: %lld groups", (long long)ng); return RH_E_INTERNAL; }
    RH_TRY(rhk_group_bounds_of(c, c->dis + first, c->dis_stride, cnt, ng, c->dis_gb, c->ng_pad));
    hipLaunchKernelGGL(gb32_kernel, dim3(cdiv4(ng, 256)), dim3(256), 0, c->stream, c->dis_gb, c->ng_pad, ng, c->dis_gb32);
    RH_HIP(hipGetLastError());
    const rh_s4_points PS = { c->dis + first, c->dis_stride, cnt, ng, c->dis_gb32 };
    const uint64_t *en[4] = { nullptr, nullptr, nullptr, nullptr };
    c->s4_points = &PS;
    const int rc = rhk_score4_all(c, en, prep, cls, box, bstride, orig, nk, nk_total_bound, eps, cosa, d_counts, nullptr, nullptr, 0);
    c->s4_points = nullptr;
    return rc;
}

#ifdef RH_DIAG
// diagnostics (tests): the classifier's worst binary32 error on a batch, in units of its margin widths (see
// cls_audit_kernel); out[12]: per kind (plane, sphere, cylinder, -) the two maxima, then the numbers of pairs
extern "C" int rh_dbg_cls_audit(rh_cloud *c, const rh_shape *shapes, int32_t b, const rh_params *p, double *out)
{
    if (c == nullptr || out == nullptr || p == nullptr || b < 0 || (b > 0 && shapes == nullptr)) { rh_set_error("rh_dbg_cls_audit: bad arguments"); return RH_E_INVALID; }
    for (int i = 0; i < 12; i++) out[i] = 0.0;
    if (b == 0 || c->s == 0) return RH_OK;
    RH_HIP(hipSetDevice(c->device));
    std::vector<S4AuditCand> h((size_t)b);
    for (int32_t i = 0; i < b; i++) {
        S4AuditCand &Q = h[(size_t)i];
        Q.kind = shapes[i].kind;
        Q.usable = 0;
        if (Q.kind < 0 || Q.kind > 3) continue;
        rh_prep_host(shapes[i], &Q.P);
        double d4[4] = { 0, 0, 0, 0 };
        cls_make(Q.P, Q.kind, p->eps[Q.kind], p->cos_alpha[Q.kind], c->coord_mag, c->nrm_mag, Q.C, nullptr, 0, d4, c->f32);
        Q.cNhi = d4[0]; Q.wN = d4[1]; Q.eDlo = d4[2]; Q.wD = d4[3]; Q.cosa = p->cos_alpha[Q.kind];
        Q.usable = !(Q.C.f[RH_CLS_FLAG] != Q.C.f[RH_CLS_FLAG]) && Q.wN > 0 && Q.wD > 0;
    }
    S4AuditCand *d_c = nullptr;
    unsigned long long *d_o = nullptr;
    RH_HIP(hipMalloc((void **)&d_c, sizeof(S4AuditCand) * (size_t)b));
    RH_HIP(hipMalloc((void **)&d_o, sizeof(unsigned long long) * 12));
    RH_HIP(hipMemcpyAsync(d_c, h.data(), sizeof(S4AuditCand) * (size_t)b, hipMemcpyHostToDevice, c->stream));
    RH_HIP(hipMemsetAsync(d_o, 0, sizeof(unsigned long long) * 12, c->stream));
    for (int32_t c0 = 0; c0 < b; c0 += 32768) {
        const int32_t nb = std::min<int32_t>(32768, b - c0);
        hipLaunchKernelGGL(cls_audit_kernel, dim3((unsigned)cdiv4(c->s, 256), (unsigned)nb), dim3(256), 0, c->stream, c->sub, c->s_pad,
                           c->s, d_c + c0, nb, d_o);
    }
    unsigned long long ho[12];
    RH_HIP(hipMemcpyAsync(ho, d_o, sizeof ho, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(d_c);
    (void)hipFree(d_o);
    for (int i = 0; i < 8; i++) out[i] = __builtin_bit_cast(double, ho[i]);
    for (int i = 8; i < 12; i++) out[i] = (double)ho[i];
    return RH_OK;
}

// diagnostics (tests, tools/fuzz_score.py): the decisions of the box test and of the classifier against the exact test
// on every (candidate, point) of a batch x subset 1 (all points taken as enabled): out[56], layout at sound_one
extern "C" int rh_dbg_cls_soundness(rh_cloud *c, const rh_shape *shapes, int32_t b, const rh_params *p, uint64_t *out)
{
    if (c == nullptr || out == nullptr || p == nullptr || b < 0 || (b > 0 && shapes == nullptr)) { rh_set_error("rh_dbg_cls_soundness: bad arguments"); return RH_E_INVALID; }
    for (int i = 0; i < 56; i++) out[i] = 0;
    if (b == 0 || c->s == 0 || c->ngroups == 0 || c->gb32 == nullptr || c->st32 == nullptr) return RH_OK;
    RH_HIP(hipSetDevice(c->device));
    std::vector<S4SoundCand> h((size_t)b);
    for (int32_t i = 0; i < b; i++) {
        S4SoundCand &Q = h[(size_t)i];
        Q.kind = shapes[i].kind;
        Q.pad = 0;
        if (Q.kind < 0 || Q.kind > 3) { Q.kind = -1; continue; }
        rh_prep_host(shapes[i], &Q.P);
        prep_one32(shapes[i], Q.Pf);
        cls_make(Q.P, Q.kind, p->eps[Q.kind], p->cos_alpha[Q.kind], c->coord_mag, c->nrm_mag, Q.C, Q.box, 1, nullptr, c->f32);
    }
    S4SoundArgs A;
    for (int k = 0; k < 4; k++) { A.eps[k] = p->eps[k]; A.cosa[k] = p->cos_alpha[k]; }
    A.f32 = c->f32 ? 1 : 0;
    S4SoundCand *d_c = nullptr;
    unsigned long long *d_o = nullptr;
    RH_HIP(hipMalloc((void **)&d_c, sizeof(S4SoundCand) * (size_t)b));
    RH_HIP(hipMalloc((void **)&d_o, sizeof(unsigned long long) * 56));
    RH_HIP(hipMemcpyAsync(d_c, h.data(), sizeof(S4SoundCand) * (size_t)b, hipMemcpyHostToDevice, c->stream));
    RH_HIP(hipMemsetAsync(d_o, 0, sizeof(unsigned long long) * 56, c->stream));
    hipLaunchKernelGGL(cls_sound_kernel, dim3((unsigned)c->ngroups, (unsigned)std::min<int32_t>(b, 64)), dim3(64), 0, c->stream, c->sub, c->s_pad, c->s,
                       c->gb32, c->st32, d_c, b, A, d_o);
    unsigned long long ho[56];
    RH_HIP(hipMemcpyAsync(ho, d_o, sizeof ho, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(d_c);
    (void)hipFree(d_o);
    for (int i = 0; i < 56; i++) out[i] = (uint64_t)ho[i];
    return RH_OK;
}

#endif   // RH_DIAG

#ifdef RH_DIAG
// diagnostics: the event counters of the score launches on this cloud (layout: S4_STAT above).  mode 0: read (out[128], null:
// nothing), 1: switch the counting on and zero the counters, 2: switch it off.  Synchronises the cloud's stream.
extern "C" int rh_dbg_s4_stats(rh_cloud *c, int mode, uint64_t *out)
{
    if (c == nullptr || mode < 0 || mode > 2) { rh_set_error("rh_dbg_s4_stats: bad arguments"); return RH_E_INVALID; }
    RH_HIP(hipSetDevice(c->device));
    RH_HIP(hipStreamSynchronize(c->stream));
    if (mode == 1) {
        if (c->s4_stats == nullptr) RH_HIP(hipMalloc((void **)&c->s4_stats, sizeof(unsigned long long) * 128));
        RH_HIP(hipMemset(c->s4_stats, 0, sizeof(unsigned long long) * 128));
    } else if (mode == 2) {
        if (c->s4_stats != nullptr) (void)hipFree(c->s4_stats);
        c->s4_stats = nullptr;
    } else if (out != nullptr) {
        for (int i = 0; i < 128; i++) out[i] = 0;
        if (c->s4_stats != nullptr) RH_HIP(hipMemcpy(out, c->s4_stats, sizeof(unsigned long long) * 128, hipMemcpyDeviceToHost));
    }
    return RH_OK;
}
#endif   // RH_DIAG

// the entry lists the v4 score kernel left (rows of mstride 16-byte entries, one cursor per row in d_occ) -> dense rows in
// subset order; the cursors are zero again afterwards
int rhk_unpermute_masks4(rh_cloud *c, const uint64_t *d_in, uint8_t *d_occ, int64_t mstride, int32_t b, uint64_t *d_out)
{
    if (b == 0 || c->swords == 0) return RH_OK;
    const int env_words = (int)rh_opt_int(c, RH_OPT_UNP_WORDS, 0);   // rh_set_option(.., "unp_words", n) (tests; read per call): segments of n words, so that a small cloud's rows span several / many
    const int64_t seg_words = std::min<int64_t>(c->swords, env_words > 0 ? std::min(env_words, 16384) : (c->swords <= S4_UNP_WORDS ? S4_UNP_WORDS : S4_UNP_WORDS_MULTI));
    const int nseg = cdiv4(c->swords, seg_words);
    static bool attr_set = false;
    if (!attr_set) {
        RH_HIP(hipFuncSetAttribute((const void *)unpermute6_kernel<false, RH_UNP_BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint64_t) * 16384)));
        RH_HIP(hipFuncSetAttribute((const void *)unpermute6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint64_t) * 16384)));
        attr_set = true;
    }
    const uint64_t *sm = nullptr;
    if (nseg > 1 && nseg <= 16) {   // which bits of a word belong to which segment: made once per cloud and segment width
        if (c->unp_segmask == nullptr || c->unp_seg_words != seg_words) {
            (void)hipFree(c->unp_segmask);
            c->unp_segmask = nullptr;
            RH_HIP(hipMalloc((void **)&c->unp_segmask, sizeof(uint64_t) * (size_t)nseg * (size_t)c->ng_pad));
            hipLaunchKernelGGL(segmask_kernel, dim3((unsigned)cdiv4(c->ngroups, 256)), dim3(256), 0, c->stream, c->sub_perm, c->s, c->ngroups,
                               seg_words * 64, nseg, c->ng_pad, c->unp_segmask);
            RH_HIP(hipGetLastError());
            c->unp_seg_words = seg_words;
        }
        sm = c->unp_segmask;
    }
    const bool waveword = nseg == 1;
    if (waveword)
        hipLaunchKernelGGL(unpermute6_kernel<true>, dim3((unsigned)b, (unsigned)nseg), dim3(512), sizeof(uint64_t) * (size_t)seg_words, c->stream,
                           (const rh_u64x2 *)d_in, (const int32_t *)d_occ, mstride, c->sub_perm, c->swords, seg_words, sm, c->ng_pad, d_out);
    else
        hipLaunchKernelGGL((unpermute6_kernel<false, RH_UNP_BLOCK>), dim3((unsigned)b, (unsigned)nseg), dim3(RH_UNP_BLOCK), sizeof(uint64_t) * (size_t)seg_words, c->stream,
                           (const rh_u64x2 *)d_in, (const int32_t *)d_occ, mstride, c->sub_perm, c->swords, seg_words, sm, c->ng_pad, d_out);
    hipLaunchKernelGGL(clear_cursors_kernel, dim3((unsigned)cdiv4(b, 256)), dim3(256), 0, c->stream, (int32_t *)d_occ, b);
    RH_HIP(hipGetLastError());
    return RH_OK;
}
