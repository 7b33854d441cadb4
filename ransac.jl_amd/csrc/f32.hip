// f32.hip -- Float32 clouds: RANSACCloud(...; force_eltype = Float32) (src/octree.jl:102-109).
//
// A Float32 cloud makes every operation of compatiblesPlane / Sphere / Cylinder / Cone (plane.jl:114-130,
// sphere.jl:144-172, cylinder.jl:194-221, cone.jl:68-85,132-153) a binary32 operation -- the points, the normals and the
// shapes fitted to them are Float32 -- while eps and cos(alpha) stay what the caller made them (Float64 unless
// setfloattype converted them, utilities.jl:488-504): Julia promotes the Float32 side of such a comparison exactly, so
// the kernels compute in float and compare the result, converted to double, with the double threshold.
//
// What a Float32 cloud has on the device: everything a Float64 cloud has (the points converted exactly to double feed the
// k-d leaf order, the enabled-bit machinery, masks and index lists), plus float copies of the two point sets --
// `full32` (original order, 24 bytes per point: what the refit scan streams, half the bytes of the Float64 scan) and
// `sub32` (subset 1 in k-d leaf order) -- and float candidate records.  Batched scoring runs on the culled kernel of
// score4.hip with the exact test in binary32 (its classifier margins bracket the reference's binary32 chain, score4_device.h)
// from 8192 subset points on, and on the
// brute-force float kernel below for smaller subsets (RH_SCORE_PATH=brute forces it); refit, masks and the enabled
// bits work as on a Float64 cloud.  rh_ransac runs on such a cloud too (binary32 fits: fit_shared.h; no cones); rh_refit_lsq
// stays Float64-only.
//
// The four tests below are the float twins of score_device.h, statement by statement; the oracle's twin is
// oracle/orc_f32.c.
#include <stdlib.h>

#include "det_math.h"
#include "rh_internal.h"
#include "score_device32.h"

namespace {

using namespace rhdev32;

// the batch is binned by kind exactly like the Float64 path (bin k at offset off[k] of prep / orig, its size in nk[k]); the
// float record of slot t of bin k is made from its source shape: shapes[off[k] + t] when the shapes array is sorted like
// the bins (rh_score_batch), shapes[orig[off[k] + t]] when it is in the caller's order (rh_score_batch_dev)
struct Off4 { int64_t o[4]; };

__global__ void prep32_kernel(const rh_shape *__restrict__ shapes, int via_orig, const int32_t *__restrict__ orig, const Off4 off,
                              const int32_t *__restrict__ nk, rh_prepf *__restrict__ prep32)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (t < nk[k]) {
            const int64_t slot = off.o[k] + t;
            prep_one32(shapes[via_orig ? (int64_t)orig[slot] : slot], prep32[slot]);
        }
    }
}

__global__ void to_float_kernel(const double *__restrict__ src, int64_t count, float *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (float)src[i];
}

// ---- batched score, brute force over float planes (the structure of score_kernel, kernels.hip) ----
template <int KIND, bool MASK>
__global__ void __launch_bounds__(RH_SC_THREADS)
score32_kernel(const float *__restrict__ pts, int64_t stride, int64_t s, const uint64_t *__restrict__ enabled_words,
               const rh_prepf *__restrict__ prep, const int32_t *__restrict__ orig, const int32_t *__restrict__ nk_ptr,
               double eps, double cosa, int32_t *__restrict__ counts, uint64_t *__restrict__ masks, int64_t mask_stride)
{
    const int nk = *nk_ptr;
    const int c0 = blockIdx.y * RH_SC_CT;
    if (c0 >= nk) return;
    const int nc = min(RH_SC_CT, nk - c0);
    __shared__ int32_t lcnt[RH_SC_CT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < RH_SC_CT) lcnt[tid] = 0;
    __syncthreads();
    const float *__restrict__ X = pts, *__restrict__ Y = pts + stride, *__restrict__ Z = pts + 2 * stride;
    const float *__restrict__ NX = pts + 3 * stride, *__restrict__ NY = pts + 4 * stride, *__restrict__ NZ = pts + 5 * stride;
    const int64_t ntiles = (s + RH_SC_TILE - 1) / RH_SC_TILE;
    const int64_t swords = (s + 63) >> 6;
    int acc = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t wbase = tile * RH_SC_TILE + (int64_t)wave * RH_SC_WAVE_PTS;
        float px[RH_SC_PPT], py[RH_SC_PPT], pz[RH_SC_PPT], qx[RH_SC_PPT], qy[RH_SC_PPT], qz[RH_SC_PPT];
        uint64_t en[RH_SC_PPT];
#pragma unroll
        for (int p = 0; p < RH_SC_PPT; p++) {
            const int64_t i = wbase + p * 64 + lane;
            px[p] = X[i]; py[p] = Y[i]; pz[p] = Z[i];
            qx[p] = NX[i]; qy[p] = NY[i]; qz[p] = NZ[i];
            const int64_t gb = wbase + p * 64;
            uint64_t v = valid_mask32(gb, s);
            if (enabled_words != nullptr && v != 0) v &= enabled_words[gb >> 6];
            en[p] = v;
        }
        rh_prepf P = ld_prepf(&prep[c0]);
        for (int c = 0; c < nc; c++) {
            const rh_prepf Pn = ld_prepf(&prep[c0 + min(c + 1, nc - 1)]);   // the next record is in flight while this one is used
            uint64_t bw[RH_SC_PPT];
            int n = 0;
#pragma unroll
            for (int p = 0; p < RH_SC_PPT; p++) {
                bw[p] = test_point32<KIND>(P, px[p], py[p], pz[p], qx[p], qy[p], qz[p], eps, cosa) & en[p];
                n += __popcll(bw[p]);
            }
            acc += (lane == c) ? n : 0;
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

// ---- refit scan over the float planes of the whole cloud: 24 bytes per point ----
constexpr int RF32_WPW = 4;   // 64-point words per wave (24 loads in flight per lane)

template <int KIND>
__global__ void __launch_bounds__(256)
refit32_mask_kernel(const float *__restrict__ pts, int64_t stride, int64_t n, int64_t nwords, const uint64_t *__restrict__ enabled,
                    const rh_prepf P, double eps, double cosa, uint64_t *__restrict__ mask_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const float *__restrict__ X = pts, *__restrict__ Y = pts + stride, *__restrict__ Z = pts + 2 * stride;
    const float *__restrict__ NX = pts + 3 * stride, *__restrict__ NY = pts + 4 * stride, *__restrict__ NZ = pts + 5 * stride;
    const int64_t ngroups = (nwords + RF32_WPW - 1) / RF32_WPW;
    for (int64_t g = wave0; g < ngroups; g += nwaves) {
        const int64_t w0 = g * RF32_WPW;
        uint64_t en[RF32_WPW];
        float px[RF32_WPW], py[RF32_WPW], pz[RF32_WPW], qx[RF32_WPW], qy[RF32_WPW], qz[RF32_WPW];
#pragma unroll
        for (int k = 0; k < RF32_WPW; k++) {
            const int64_t w = w0 + k;
            en[k] = w < nwords ? (enabled[w] & valid_mask32(w << 6, n)) : 0ULL;
        }
#pragma unroll
        for (int k = 0; k < RF32_WPW; k++) {
            const int64_t i = (en[k] != 0 ? ((w0 + k) << 6) : (int64_t)0) + lane;   // an all-disabled word re-reads the hot first line
            px[k] = X[i]; py[k] = Y[i]; pz[k] = Z[i];
            qx[k] = NX[i]; qy[k] = NY[i]; qz[k] = NZ[i];
        }
#pragma unroll
        for (int k = 0; k < RF32_WPW; k++) {
            const uint64_t b = test_point32<KIND>(P, px[k], py[k], pz[k], qx[k], qy[k], qz[k], eps, cosa) & en[k];
            if (lane == 0 && w0 + k < nwords) mask_out[w0 + k] = b;
        }
    }
}

inline int cdiv32(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace

// float copies of the two point sets (values of a Float32 cloud are binary32 numbers: the conversion is exact)
int rhk_f32_build(rh_cloud *c)
{
    if (c->n_pad > 0) {
        hipLaunchKernelGGL(to_float_kernel, dim3(cdiv32(6 * c->n_pad, 256)), dim3(256), 0, c->stream, c->full, 6 * c->n_pad, c->full32);
    }
    if (c->s_pad > 0) {
        hipLaunchKernelGGL(to_float_kernel, dim3(cdiv32(6 * c->s_pad, 256)), dim3(256), 0, c->stream, c->sub, 6 * c->s_pad, c->sub32);
    }
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_prep_f32(rh_cloud *c, const rh_shape *d_shapes, int via_orig, const int32_t *d_orig, const int64_t off[4],
                 const int32_t *d_nk, int32_t nmax)
{
    if (nmax <= 0) return RH_OK;
    Off4 o4;
    for (int k = 0; k < 4; k++) o4.o[k] = off[k];
    hipLaunchKernelGGL(prep32_kernel, dim3(cdiv32(nmax, 256)), dim3(256), 0, c->stream, d_shapes, via_orig, d_orig, o4, d_nk,
                       (rh_prepf *)c->d_prep32);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// all kinds of a binned batch (the Float64 path's bins: orig / nk, bin k at off[k]) against subset 1; masks (optional) in
// INTERNAL order like the other scorers
int rhk_score_all_f32(rh_cloud *c, const rh_shape *d_shapes, int via_orig, const uint64_t *const en[4], const int32_t *d_orig,
                      const int64_t off[4], const int32_t *d_nk, const int32_t nk_bound[4], const double eps[4],
                      const double cosa[4], int32_t *d_counts, uint64_t *d_masks_int)
{
    int nmax = 0;
    for (int k = 0; k < 4; k++) nmax = std::max(nmax, (int)nk_bound[k]);
    if (nmax == 0 || c->s == 0) return RH_OK;
    rh_prepf *prep32 = (rh_prepf *)c->d_prep32;
    RH_TRY(rhk_prep_f32(c, d_shapes, via_orig, d_orig, off, d_nk, nmax));
    const int64_t ntiles = (c->s + RH_SC_TILE - 1) / RH_SC_TILE;
    for (int k = 0; k < 4; k++) {
        if (nk_bound[k] == 0) continue;
        const int ctiles = cdiv32(nk_bound[k], RH_SC_CT);
        if (ctiles > 65535) { rh_set_error("batch of %d candidates is too large for one launch", nk_bound[k]); return RH_E_INVALID; }
        int64_t splits = 65536 / ctiles;
        if (splits < 1) splits = 1;
        if (splits > ntiles) splits = ntiles;
        dim3 grid((unsigned)splits, (unsigned)ctiles);
#define RH_S32(K, M)                                                                                                      \
    hipLaunchKernelGGL((score32_kernel<K, M>), grid, dim3(RH_SC_THREADS), 0, c->stream, c->sub32, c->s_pad, c->s, en[k],    \
                       prep32 + off[k], d_orig + off[k], d_nk + k, eps[k], cosa[k], d_counts, d_masks_int, c->swords)
        if (d_masks_int) {
            switch (k) {
            case RH_PLANE: RH_S32(RH_PLANE, true); break;
            case RH_SPHERE: RH_S32(RH_SPHERE, true); break;
            case RH_CYLINDER: RH_S32(RH_CYLINDER, true); break;
            default: RH_S32(RH_CONE, true); break;
            }
        } else {
            switch (k) {
            case RH_PLANE: RH_S32(RH_PLANE, false); break;
            case RH_SPHERE: RH_S32(RH_SPHERE, false); break;
            case RH_CYLINDER: RH_S32(RH_CYLINDER, false); break;
            default: RH_S32(RH_CONE, false); break;
            }
        }
#undef RH_S32
    }
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_refit_mask_f32(rh_cloud *c, const rh_shape &shape, double eps, double cosa, bool apply)
{
    c->k_applied = false;
    c->k_sums_ready = false;
    if (c->nwords == 0) return RH_OK;
    rh_prepf P;
    prep_one32(shape, P);
    if (rhk_refit_is_culled(c)) {   // box tests in binary64 on the shape's record, exact test in binary32 (korder.hip)
        rh_prep P64;
        rh_prep_host(shape, &P64);
        RH_TRY(rhk_refitk_mask_f32(c, &P, P64, shape.kind, eps, cosa, apply));
        c->k_applied = apply;
        c->k_sums_ready = apply;
        return RH_OK;
    }
    int64_t blocks = cdiv32(c->nwords, 4 * RF32_WPW);
    if (blocks > 32768) blocks = 32768;
    dim3 grid((unsigned)blocks), blk(256);
#define RH_R32(K)                                                                                                      \
    hipLaunchKernelGGL((refit32_mask_kernel<K>), grid, blk, 0, c->stream, c->full32, c->n_pad, c->n, c->nwords, c->enabled, \
                       P, eps, cosa, c->refit_mask)
    switch (shape.kind) {
    case RH_PLANE: RH_R32(RH_PLANE); break;
    case RH_SPHERE: RH_R32(RH_SPHERE); break;
    case RH_CYLINDER: RH_R32(RH_CYLINDER); break;
    case RH_CONE: RH_R32(RH_CONE); break;
    default: rh_set_error("unknown shape kind %d", shape.kind); return RH_E_INVALID;
    }
#undef RH_R32
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// a Float32 shape: fields rounded to binary32, the cone's cos / sin of -opang/2 as binary32 (cos / sin of a Float32 are
// Float32 in Julia); the deterministic kernels of det_math.h evaluated on the binary32 angle, rounded once
extern "C" void rh_shape_finalize_f32(rh_shape *s)
{
    if (!s) return;
    for (int i = 0; i < 7; i++) s->v[i] = (double)(float)s->v[i];
    if (s->kind == RH_CONE) {
        const float th = -(float)s->v[6] / 2.0f;
        s->v[7] = (double)(float)rh_cos((double)th);
        s->v[8] = (double)(float)rh_sin((double)th);
    }
}
