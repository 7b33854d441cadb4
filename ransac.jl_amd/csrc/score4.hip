// score4.hip -- the v4 batched score kernel: scorecandidates! / scorecandidate (/root/reference/src/fitting.jl:181-190,
// shapes/plane.jl:61-71, sphere.jl:118-134, cylinder.jl:172-183, cone.jl:155-167) for a whole batch in one launch.
//
// Decomposition as in the culled kernel of kernels.hip: subset 1 in k-d leaf order, 64-point groups with boxes, a block
// per (tile of RH_G2_TG groups, row of the batch's 64-candidate chunks), all kinds in one launch.  What is new:
//
//  * The scalar unit is the scarce resource (measured, tools/ubench/valu_rates.hip: a scalar instruction costs a SIMD
//    ~4.2 cycles -- one scalar ALU per CU -- as much as a binary64 vector instruction; a binary32 one costs ~2.3).  The
//    older kernel walks (candidate, group) pairs with lane = point: per visit a dozen scalar instructions of loop
//    control, ballots and popcounts around 15 vector ones.  Here a LANE OWNS A PAIR: stage 1 (lane = candidate, binary32
//    box tests against boxes held in scalar registers) appends the surviving (candidate, group) pairs to a list in LDS;
//    stage 2 takes 64 pairs at a time, one per lane, and every lane loops over the 64 points of ITS group (LDS reads at
//    per-lane addresses: at most RH_G2_TG distinct rows per instruction, padded apart in the banks), its candidate's
//    record in vector registers, the inlier count in a vector register: no scalar instruction in the loop but its
//    control, no ballot, no reduction.
//  * The per-point work is the two-sided binary32 classifier of score4_device.h on a binary32 tile (24 B per point);
//    only what it cannot decide reaches the reference's binary64 test (score_device.h, unchanged):
//      plane     t = min(a, b) per point; sure <=> t > 0, maybe <=> t > -1; a pair with maybe != sure anywhere is
//                redone as a whole by the exact test (lane = point, points from global memory)
//      sphere /  per pair the 64-bit mask of the points inside the (widened) band; the set bits of 64 pairs are
//      cylinder  compacted on a per-wave ring and get the full classifier with lane = (pair, point): sure -> count,
//                ambiguous -> ring B -> exact test of that one point
//      cone      band prefilter per pair, every survivor -> ring B -> exact test
//
// Bit-exactness does not depend on the classifier's margins being tight, only on their being upper bounds (proof
// obligations and the audit kernel: score4_device.h).
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
constexpr int S4_R = 8;                  // chunks per pass (a row is walked in passes)
constexpr int S4_ROW = 65;               // padded row length of the tile arrays (bank spread of the per-lane rows)
static_assert(S4_TG == 4, "entry encoding: 2 bits of group");

struct S4Shared {
    rh_f32x4 pa[S4_TG][S4_ROW];          // (x, y, z, nx); zeros for a disabled / out-of-range point
    rh_f32x2 pb[S4_TG][S4_ROW];          // (ny, nz)
    uint64_t len[S4_TG];                 // enabled & valid bits of the groups
    uint16_t plist[S4_R * 64 * S4_TG];   // surviving pairs of the pass: g | lane << 2 | chunk-in-pass << 8 | redo << 15
    int32_t cnt[S4_R * 64];              // inlier counts of the pass's candidates on this tile
    uint16_t qa[4][128];                 // per-wave ring A: (slot-in-batch << 6 | point-in-group), band pairs
    uint16_t qb[4][128];                 // per-wave ring B: the same, for the exact test
    int npairs, next_batch;
    int weirdw[4];                       // per wave: it staged an enabled point with a non-finite value
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
    int64_t ntiles, bstride;
    const float *gb32;      // binary32 boxes of the groups, 8 floats each
};

template <int KIND> struct S4Fields { static constexpr int NBOX = KIND == RH_PLANE || KIND == RH_SPHERE ? 5 : (KIND == RH_CYLINDER ? 9 : 10); };

#define RH4_CONST_AS __attribute__((address_space(4)))

template <int KIND>
static __device__ __forceinline__ void
score4_body(S4Shared &sh, const int chunk_lo, const int chunk_hi, const double *__restrict__ pts, int64_t stride, int64_t s,
            const uint64_t *__restrict__ enabled_words, const float *__restrict__ gb32, int64_t ngroups,
            const rh_prep *__restrict__ prep, const rh_cls *__restrict__ cls, const float *__restrict__ box, int64_t bstride,
            const int32_t *__restrict__ orig, const int32_t *__restrict__ nk_ptr, double eps, double cosa,
            int32_t *__restrict__ counts, int dbg, const int64_t tile, const bool ran)
{
    const int nk = *nk_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t g0 = tile * S4_TG;
    const int64_t p0 = g0 * 64;
    if (ran) __syncthreads();   // the previous segment's waves are done with the tile, the list and the counters
    // ---- staging: the tile as binary32, the groups' enabled words, zeroed counters
    {
        static_assert(S4_TG * 64 == 256, "one point per thread");
        const int64_t gi = p0 + tid;
        uint64_t v = valid_mask((gi >> 6) << 6, s);
        if (enabled_words != nullptr && v != 0) v &= enabled_words[gi >> 6];
        const bool on = (v >> (gi & 63)) & 1ULL;
        rh_f32x4 a = { 0.f, 0.f, 0.f, 0.f };
        rh_f32x2 b = { 0.f, 0.f };
        bool bad = false;
        if (on) {
            a.x = (float)pts[gi]; a.y = (float)pts[stride + gi]; a.z = (float)pts[2 * stride + gi];
            a.w = (float)pts[3 * stride + gi]; b.x = (float)pts[4 * stride + gi]; b.y = (float)pts[5 * stride + gi];
            const float sum = (fabsf(a.x) + fabsf(a.y)) + (fabsf(a.z) + fabsf(a.w)) + (fabsf(b.x) + fabsf(b.y));
            bad = !(sum < __builtin_inff());   // an infinite or NaN value (binary32 overflow included)
        }
        sh.pa[tid >> 6][tid & 63] = a;
        sh.pb[tid >> 6][tid & 63] = b;
        const uint64_t wb = WB(bad);
        if (lane == 0) { sh.weirdw[wv] = wb != 0 ? 1 : 0; sh.len[wv] = v; }   // wave w stages group w: v is its word
        sh.cnt[tid] = 0; sh.cnt[tid + 256] = 0;
        static_assert(S4_R * 64 == 512, "two counters per thread");
        if (tid == 0) { sh.npairs = 0; sh.next_batch = 0; }
    }
    // the boxes of the tile's groups: wave-uniform, in scalar registers for the whole block
    rh_box32 G[S4_TG];
    {
        const RH4_CONST_AS float *gq = (const RH4_CONST_AS float *)(uintptr_t)gb32;
#pragma unroll
        for (int g = 0; g < S4_TG; g++) {
            const int64_t gg = g0 + g < ngroups ? g0 + g : ngroups - 1;   // (ngroups >= 1 here)
            const RH4_CONST_AS float *q = gq + gg * 8;
            G[g].cx = q[0]; G[g].cy = q[1]; G[g].cz = q[2]; G[g].hx = q[3]; G[g].hy = q[4]; G[g].hz = q[5]; G[g].hr = q[6];
        }
    }
    __syncthreads();
    unsigned live = 0;
#pragma unroll
    for (int g = 0; g < S4_TG; g++) live |= sh.len[g] != 0 ? (1u << g) : 0u;
    live = __builtin_amdgcn_readfirstlane(live);
    const bool weird = __builtin_amdgcn_readfirstlane(sh.weirdw[0] | sh.weirdw[1] | sh.weirdw[2] | sh.weirdw[3]) != 0;

    for (int pass_lo = chunk_lo; pass_lo < chunk_hi; pass_lo += S4_R) {
        const int pass_n = min(S4_R, chunk_hi - pass_lo);
        const int pbase = pass_lo << 6;   // first candidate slot of the pass
        // ---- stage 1: lane = candidate, box tests, surviving pairs -> list
        for (int cp = wv; cp < pass_n; cp += 4) {
            const int ci = ((pass_lo + cp) << 6) + lane;
            unsigned surv = 0;
            if (ci < nk) {
                float B[RH_BOX_FIELDS];
#pragma unroll
                for (int f = 0; f < S4Fields<KIND>::NBOX; f++) B[f] = box[(int64_t)f * bstride + ci];
#pragma unroll
                for (int g = 0; g < S4_TG; g++) {
                    const bool skip = box_skip32<KIND>(B, G[g]);
                    surv |= skip ? 0u : (1u << g);
                }
                if (dbg == 2) surv = 15u;
                surv &= live;
            }
            if (dbg == 1) surv = 0;
            uint64_t m[S4_TG];
            int tot = 0;
#pragma unroll
            for (int g = 0; g < S4_TG; g++) { m[g] = WB((surv >> g) & 1u); tot += __popcll(m[g]); }
            if (tot == 0) continue;
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh.npairs, tot);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int g = 0; g < S4_TG; g++) {
                if ((surv >> g) & 1u) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[g] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[g], 0));
                    sh.plist[base + rank] = (uint16_t)(g | (lane << 2) | (cp << 8));
                }
                base += __popcll(m[g]);
            }
        }
        __syncthreads();
        const int npairs = sh.npairs;

        // ---- stage 2: lane = pair, 64 pairs per batch
        for (;;) {
            int bt = 0;
            if (lane == 0) bt = atomicAdd(&sh.next_batch, 1);
            bt = __builtin_amdgcn_readfirstlane(bt);
            if (bt * 64 >= npairs) break;
            const int idx = bt * 64 + lane;
            const bool act = idx < npairs;
            const unsigned e = act ? sh.plist[idx] : sh.plist[bt * 64];
            const int g = (int)(e & 3u), cand = (int)((e >> 2) & 511u);
            const int ci = pbase + cand;
            const rh_f32x4 *__restrict__ rowa = &sh.pa[g][0];
            const rh_f32x2 *__restrict__ rowb = &sh.pb[g][0];
            if (KIND == RH_PLANE) {
                rh_cls C;
#pragma unroll
                for (int f = 0; f < 9; f++) C.f[f] = cls[ci].f[f];
                const bool exact_only = weird || is_nan_bits(cls[ci].f[RH_CLS_FLAG]);
                int cs = 0, cm = 0;
#pragma unroll 8
                for (int j = 0; j < 64; j++) {
                    const rh_f32x4 a = rowa[j];
                    const rh_f32x2 b = rowb[j];
                    const float t = cls_plane_t(C, a.x, a.y, a.z, a.w, b.x, b.y);
                    cs += t > 0.0f ? 1 : 0;
                    cm += t > -1.0f ? 1 : 0;
                }
                const bool amb = act && (cs != cm || exact_only);
                if (act && !amb && cs != 0) atomicAdd(&sh.cnt[cand], cs);
                if (amb) sh.plist[idx] = (uint16_t)(e | 0x8000u);
            } else {
                constexpr int NF = KIND == RH_SPHERE ? 6 : (KIND == RH_CYLINDER ? 9 : 10);
                rh_cls C;
#pragma unroll
                for (int f = 0; f < NF; f++) C.f[f] = cls[ci].f[f];
                const bool exact_only = weird || is_nan_bits(cls[ci].f[RH_CLS_FLAG]);
                uint32_t mlo = 0, mhi = 0;
#pragma unroll 8
                for (int j = 0; j < 32; j++) {
                    const rh_f32x4 a = rowa[j];
                    mlo |= cls_pre<KIND>(C, a.x, a.y, a.z) ? (1u << j) : 0u;
                }
#pragma unroll 8
                for (int j = 0; j < 32; j++) {
                    const rh_f32x4 a = rowa[32 + j];
                    mhi |= cls_pre<KIND>(C, a.x, a.y, a.z) ? (1u << j) : 0u;
                }
                const uint64_t lg = sh.len[g];
                uint64_t mask = exact_only ? lg : ((((uint64_t)mhi << 32) | mlo) & lg);
                if (!act) mask = 0;

                int qah = 0, qan = 0, qbh = 0, qbn = 0;   // ring heads / fills (wave-uniform)
                // ---- 2c: lane = (pair, point) of ring B, the reference's binary64 test; the point comes from global memory
                auto drain_b = [&](int k) {
                    wave_lds_sync();
                    const bool on = lane < k;
                    const unsigned e2 = on ? sh.qb[wv][(qbh + lane) & 127] : 0u;
                    const unsigned pe = sh.plist[bt * 64 + (int)(e2 >> 6)];
                    const int g2 = (int)(pe & 3u), cand2 = (int)((pe >> 2) & 511u);
                    const int64_t gi = p0 + g2 * 64 + (int)(e2 & 63u);
                    const rh_prep Pv = prep[pbase + cand2];
                    const uint64_t r = test_point<KIND>(Pv, pts[gi], pts[stride + gi], pts[2 * stride + gi], pts[3 * stride + gi],
                                                        pts[4 * stride + gi], pts[5 * stride + gi], eps, cosa);
                    if (on && ((r >> lane) & 1ULL)) atomicAdd(&sh.cnt[cand2], 1);
                    qbh = (qbh + k) & 127;
                    qbn -= k;
                };
                auto push_b = [&](bool p, unsigned entry) {   // every lane calls; p: this lane pushes `entry`
                    const uint64_t mm = WB(p);
                    if (mm != 0) {
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
                        if (p) sh.qb[wv][(qbh + qbn + rank) & 127] = (uint16_t)entry;
                        qbn += __popcll(mm);
                        if (qbn >= 64) drain_b(64);
                    }
                };
                // ---- 2b: lane = (pair, point) of ring A (sphere / cylinder), the full classifier
                auto drain_a = [&](int k) {
                    wave_lds_sync();
                    const bool on = lane < k;
                    const unsigned e2 = on ? sh.qa[wv][(qah + lane) & 127] : 0u;
                    const unsigned pe = sh.plist[bt * 64 + (int)(e2 >> 6)];
                    const int g2 = (int)(pe & 3u), cand2 = (int)((pe >> 2) & 511u), j2 = (int)(e2 & 63u);
                    const rh_cls Cv = cls[pbase + cand2];
                    const rh_f32x4 a = sh.pa[g2][j2];
                    const rh_f32x2 b = sh.pb[g2][j2];
                    cls_bits cb = cls_full<KIND == RH_SPHERE ? RH_SPHERE : RH_CYLINDER>(Cv, a.x, a.y, a.z, a.w, b.x, b.y);
                    if (weird || is_nan_bits(Cv.f[RH_CLS_FLAG])) { cb.sure = false; cb.maybe = true; }
                    if (on && cb.sure) atomicAdd(&sh.cnt[cand2], 1);
                    qah = (qah + k) & 127;
                    qan -= k;
                    push_b(on && cb.maybe && !cb.sure, e2);
                };
                // the set bits of the 64 masks, one per lane and round, compacted onto the ring
                while (WB(mask != 0) != 0) {
                    const bool has = mask != 0;
                    const int j = has ? __builtin_ctzll(mask) : 0;
                    mask &= mask - 1;
                    const unsigned ent = (unsigned)((lane << 6) | j);
                    if (KIND == RH_CONE) {
                        push_b(has, ent);
                    } else {
                        const uint64_t mm = WB(has);
                        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
                        if (has) sh.qa[wv][(qah + qan + rank) & 127] = (uint16_t)ent;
                        qan += __popcll(mm);
                        if (qan >= 64) drain_a(64);
                    }
                }
                if (KIND != RH_CONE && qan > 0) drain_a(qan);
                if (qbn > 0) drain_b(qbn);
            }
        }
        __syncthreads();
        // ---- plane pairs the classifier could not decide: the exact test on the whole group, lane = point
        if (KIND == RH_PLANE) {
            for (int i0 = wv * 64; i0 < npairs; i0 += 256) {
                const int idx = i0 + lane;
                const unsigned e = idx < npairs ? sh.plist[idx] : 0u;
                uint64_t redo = WB((e & 0x8000u) != 0);
                while (redo != 0) {
                    const int k = __builtin_ctzll(redo);
                    redo &= redo - 1;
                    const unsigned ek = __builtin_amdgcn_readlane(e, k);
                    const int g = (int)(ek & 3u), cand = (int)((ek >> 2) & 511u);
                    const rh_prep P = rh_ld_prep_const(&prep[pbase + cand]);
                    const int64_t gi = p0 + g * 64 + lane;
                    const uint64_t mres = test_point<KIND>(P, pts[gi], pts[stride + gi], pts[2 * stride + gi], pts[3 * stride + gi],
                                                           pts[4 * stride + gi], pts[5 * stride + gi], eps, cosa) & sh.len[g];
                    if (lane == 0 && mres != 0) atomicAdd(&sh.cnt[cand], __popcll(mres));
                }
            }
            __syncthreads();
        }
        // ---- the pass's counts leave the block; counters and list are reset for the next pass
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int t = tid + h * 256;
            const int v = sh.cnt[t];
            if (v != 0) {
                atomicAdd(&counts[orig[pbase + t]], v);
                sh.cnt[t] = 0;
            }
        }
        if (tid == 0) { sh.npairs = 0; sh.next_batch = 0; }
        __syncthreads();
    }
}

// all four kinds in one launch (see score_groups_all_kernel in kernels.hip for the row / chunk layout)
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES == 8 ? 8 : 1)
score4_all_kernel(const double *__restrict__ pts, int64_t stride, int64_t s, int64_t ngroups, const S4AllArgs A,
                  int32_t *__restrict__ counts, int dbg)
{
    __shared__ S4Shared sh;
    const int64_t tile = blockIdx.x;
    const int row = blockIdx.y, rows = gridDim.y;
    if (tile >= A.ntiles) return;
    int nch[4], total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { nch[k] = (*A.k[k].nk + 63) >> 6; total += nch[k]; }
    const int cpb = (total + rows - 1) / rows;
    const int lo = row * cpb, hi = min(total, lo + cpb);
    if (lo >= hi) return;
    int base = 0;
    bool ran = false;
#define RH_S4_BODY(K)                                                                                                  \
    {                                                                                                                  \
        const int slo = max(lo, base) - base, shi = min(hi, base + nch[K]) - base;                                     \
        if (slo < shi) {                                                                                               \
            score4_body<K>(sh, slo, shi, pts, stride, s, A.k[K].en, A.gb32, ngroups, A.k[K].prep, A.k[K].cls,         \
                           A.k[K].box, A.bstride, A.k[K].orig, A.k[K].nk, A.k[K].eps, A.k[K].cosa, counts, dbg, tile, \
                           ran);                                                                                       \
            ran = true;                                                                                                \
        }                                                                                                              \
        base += nch[K];                                                                                                \
    }
    RH_S4_BODY(RH_CONE)
    RH_S4_BODY(RH_CYLINDER)
    RH_S4_BODY(RH_SPHERE)
    RH_S4_BODY(RH_PLANE)
#undef RH_S4_BODY
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

// the v4 launch: cls[k] / box[k] = the classifier / culling records of bin prep[k], slot for slot, made for eps / cosa
int rhk_score4_all(rh_cloud *c, const uint64_t *const en[4], const rh_prep *const prep[4], const void *const cls[4],
                   const float *const box[4], int64_t bstride, const int32_t *const orig[4], const int32_t *const nk[4],
                   int32_t nk_total_bound, const double eps[4], const double cosa[4], int32_t *d_counts)
{
    const int64_t ntiles = (c->ngroups + S4_TG - 1) / S4_TG;
    const int nchunks = cdiv4(nk_total_bound, 64) + 3;
    if (ntiles == 0 || nk_total_bound <= 0) return RH_OK;
    static int env_blocks = -1, dbg = -1, env_cpb = -1, env_swz = -1, env_w8 = -1;
    if (env_blocks < 0) { const char *e = getenv("RH_G2_BLOCKS"); env_blocks = e ? atoi(e) : 0; }
    if (dbg < 0) { const char *e = getenv("RH_G2_DBG"); dbg = e ? atoi(e) : 0; }
    if (env_cpb < 0) { const char *e = getenv("RH_G2_CPB"); env_cpb = e ? atoi(e) : 0; }
    if (env_swz < 0) { const char *e = getenv("RH_G2_XCD"); env_swz = e ? atoi(e) : 1; }
    if (env_w8 < 0) { const char *e = getenv("RH_G2_W8"); env_w8 = e ? atoi(e) : 1; }
    const int min_cpb = env_cpb > 0 ? env_cpb : (int)std::min<int64_t>(8, std::max<int64_t>(2, ntiles / 150));
    int64_t rows = (env_blocks > 0 ? env_blocks : 16384) / ntiles;
    if (rows > (nchunks + min_cpb - 1) / min_cpb) rows = (nchunks + min_cpb - 1) / min_cpb;
    if (rows < 1) rows = 1;
    if (rows > 65535) rows = 65535;
    S4AllArgs A;
    for (int k = 0; k < 4; k++)
        A.k[k] = { (const rh_cls *)cls[k], box[k], prep[k], orig[k], nk[k], en[k], eps[k], cosa[k] };
    A.ntiles = ntiles;
    A.bstride = bstride;
    A.gb32 = c->gb32;
    const bool pad8 = env_swz && ntiles >= 1024;   // XCD-aware grid (kernels.hip)
    dim3 grid((unsigned)(pad8 ? ((ntiles + 7) / 8) * 8 : ntiles), (unsigned)rows);
    const bool w8 = (env_w8 && (int64_t)grid.x * grid.y >= 8192) || env_w8 == 2;
    if (w8)
        hipLaunchKernelGGL((score4_all_kernel<8>), grid, dim3(256), 0, c->stream, c->sub, c->s_pad, c->s, c->ngroups, A, d_counts, dbg);
    else
        hipLaunchKernelGGL((score4_all_kernel<0>), grid, dim3(256), 0, c->stream, c->sub, c->s_pad, c->s, c->ngroups, A, d_counts, dbg);
    RH_HIP(hipGetLastError());
    return RH_OK;
}
