// korder.hip -- the full cloud a second time, in Morton order, and the culled refit scan that reads it.
//
// refit (plane.jl:137-143, sphere.jl:179-185, cylinder.jl:229-235, cone.jl:161-167) tests EVERY point of the cloud
// against one shape.  The scan of kernels.hip streams the cloud in original order: 48 B per point whatever the shape.
// An extracted shape is spatially compact, so here the cloud is kept a second time sorted by (Morton code, index) --
// the order of the linear octree (src/octree.jl: cells of the bounding cube), built on the device when the cloud is
// created and shared with octree_sampling -- with an axis-aligned box per 64 consecutive points:
//   pass 1 (lane = group):  the conservative box test of the culled score kernel (score_device.h: box_skip) on every
//                           group with an enabled point; survivors go to a list (order irrelevant);
//   pass 2 (wave = group):  the exact per-point test on the listed groups only; every inlier sets a byte at its
//                           ORIGINAL index (plain stores), and with `apply` the Morton-order enabled bits (the octree's
//                           `men`) lose the inliers in the same pass;
//   pass 3 (lane = word):   the bytes become the original-order mask words the compaction downstream expects, and are
//                           cleared again.
// Results are bit-identical to the scan: a group is skipped only when no point of it can pass the distance half of the
// test (same slack rules as the score kernel), and the exact test is the same code.
// 10M points (cfg3 plane): ~6 % of the groups survive.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "rh_internal.h"
#include "score_device.h"
#include "score_device32.h"

namespace {

using namespace rhdev;
using rhdev32::rh_prepf;

inline unsigned cdivk(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ uint64_t spread21_dev(uint64_t v)
{
    v &= 0x1FFFFFULL;
    v = (v | (v << 32)) & 0x1F00000000FFFFULL;
    v = (v | (v << 16)) & 0x1F0000FF0000FFULL;
    v = (v | (v << 8)) & 0x100F00F00F00F00FULL;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}

// 63-bit code of a point in the bounding cube (lo, size): the arithmetic of the octree (IEEE division, no contraction)
__global__ void __launch_bounds__(256)
morton_kernel(const double *__restrict__ xyz, int64_t n, double lox, double loy, double loz, double size,
              uint64_t *__restrict__ code, int32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lo[3] = { lox, loy, loz };
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double t = (xyz[3 * i + k] - lo[k]) / size;
        const double q = t * 2097152.0;
        const uint64_t qi = !(q >= 0) ? 0 : (q >= 2097151.0 ? 2097151ULL : (uint64_t)q);
        c |= spread21_dev(qi) << k;
    }
    code[i] = c;
    idx[i] = (int32_t)i;
}

__global__ void __launch_bounds__(256)
invert_perm_kernel(const int32_t *__restrict__ perm, int64_t n, int32_t *__restrict__ pos)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) pos[perm[j]] = (int32_t)j;
}

__global__ void __launch_bounds__(256)
to_float_k(const double *__restrict__ src, int64_t count, float *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (float)src[i];
}

// ---- pass 1: one lane per group ---------------------------------------------------------------------------------
template <int KIND, bool F32>
__global__ void __launch_bounds__(256)
refitk_boxes_kernel(const double *__restrict__ gb, int64_t gstride, int64_t ng, int64_t n, const uint64_t *__restrict__ men,
                    const rh_prep P, double eps, double mag, int32_t *__restrict__ list, int32_t *__restrict__ ctr)
{
    __shared__ int32_t wcnt[4];
    __shared__ int32_t base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (g < ng) {
        const uint64_t en = men[g] & valid_mask(g << 6, n);
        if (en != 0) {
            const double slack = F32 ? box_slack32<KIND>(P, mag) : box_slack(P, mag);
            keep = !box_skip<KIND, F32>(P, gb[g], gb[gstride + g], gb[2 * gstride + g], gb[3 * gstride + g],
                                        gb[4 * gstride + g], gb[5 * gstride + g], gb[6 * gstride + g], eps, slack);
        }
    }
    const uint64_t b = __builtin_amdgcn_ballot_w64(keep);
    if (lane == 0) wcnt[wave] = __popcll(b);
    __syncthreads();
    // one atomic per block on the list length (same-address atomics with a return serialise: one per wave cost 6 us here)
    if (threadIdx.x == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        base_s = tot > 0 ? atomicAdd(ctr, tot) : 0;
    }
    __syncthreads();
    if (keep) {
        int off = base_s;
        for (int k = 0; k < wave; k++) off += wcnt[k];
        off += __popcll(b & ((1ULL << lane) - 1ULL));
        if (off < ng) list[off] = (int32_t)g;   // (always, unless an earlier scan died between its passes and left its count)
    }
}

// ---- pass 2: one wave per listed group --------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ uint64_t exact64(const rh_prep &P, double px, double py, double pz, double qx, double qy, double qz,
                                            double eps, double cosa)
{
    return test_point<KIND>(P, px, py, pz, qx, qy, qz, eps, cosa);
}

template <int KIND>
__device__ __forceinline__ uint64_t exact32(const rh_prepf &P, float px, float py, float pz, float qx, float qy, float qz,
                                            double eps, double cosa)
{
    return rhdev32::test_point32<KIND>(P, px, py, pz, qx, qy, qz, eps, cosa);
}

template <int KIND, bool F32, typename T, typename PREP>
__global__ void __launch_bounds__(256)
refitk_groups_kernel(const T *__restrict__ pts, int64_t stride, int64_t n, const int32_t *__restrict__ perm,
                     uint64_t *__restrict__ men, int apply, const PREP P, double eps, double cosa,
                     const int32_t *__restrict__ list, const int32_t *__restrict__ ctr, uint8_t *__restrict__ flag)
{
    const int lane = threadIdx.x & 63;
    const int wave0 = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)gridDim.x * 4;
    const int count = min(__builtin_amdgcn_readfirstlane(*ctr), (int)((n + 63) >> 6));
    const T *__restrict__ X = pts, *__restrict__ Y = pts + stride, *__restrict__ Z = pts + 2 * stride;
    const T *__restrict__ NX = pts + 3 * stride, *__restrict__ NY = pts + 4 * stride, *__restrict__ NZ = pts + 5 * stride;
    for (int e = wave0; e < count; e += nwaves) {
        const int64_t g = __builtin_amdgcn_readfirstlane(list[e]);
        const int64_t i = (g << 6) + lane;   // (the planes are padded: a partial last group reads zeros)
        const T px = X[i], py = Y[i], pz = Z[i], qx = NX[i], qy = NY[i], qz = NZ[i];
        const int32_t o = perm[i < n ? i : g << 6];
        const uint64_t en = men[g] & valid_mask(g << 6, n);
        uint64_t b;
        if constexpr (F32) b = exact32<KIND>(P, px, py, pz, qx, qy, qz, eps, cosa);
        else b = exact64<KIND>(P, px, py, pz, qx, qy, qz, eps, cosa);
        b &= en;
        if (b == 0) continue;
        // one byte per inlier at its ORIGINAL index -- a plain store: bit atomics on the mask words serialise (a shape's
        // points are often neighbours in the original order as well: 64 atomics per word, 50 us at 175 000 inliers)
        if ((b >> lane) & 1ULL) flag[o] = 1;
        if (apply && lane == 0) men[g] = en & ~b;
    }
}

// ---- pass 3: byte flags -> mask words (original order) + their popcounts per RH_WORDS_PER_BLOCK words (what the
// compaction starts from); the flags found set are cleared again, and so is the list length, for the next scan
__global__ void __launch_bounds__(256)
refitk_flags_kernel(uint8_t *__restrict__ flag, int64_t nwords, uint64_t *__restrict__ mask_out, int32_t *__restrict__ ctr,
                    int32_t *__restrict__ block_sums)
{
    __shared__ int32_t red[4];
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr[0] = 0;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < RH_WORDS_PER_BLOCK / 256; k++) {
        const int64_t w = (int64_t)blockIdx.x * RH_WORDS_PER_BLOCK + k * 256 + threadIdx.x;   // a wave reads 4 KB in a row
        if (w >= nwords) continue;
        uint4 *src = (uint4 *)(flag + (w << 6));
        uint64_t m = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = src[q];
            const uint32_t x[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int r = 0; r < 4; r++) {
                // bytes are 0 or 1: bit 0 of each of the four bytes into a nibble
                const uint32_t t = x[r] & 0x01010101u;
                const uint32_t nib = (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu;
                m |= (uint64_t)nib << (q * 16 + r * 4);
            }
        }
        mask_out[w] = m;
        acc += __popcll(m);
        if (m != 0) {
            const uint4 z = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (int q = 0; q < 4; q++) src[q] = z;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace

// ------------------------------------------------------------------------------------------------- build ----
// the Morton order of the cloud, its inverse, the reordered planes and the group boxes (all on the device)
int rhk_korder_build(rh_cloud *c, const double *d_xyz, const double *d_nrm, const double lo[3], double size, double mag)
{
    const int64_t n = c->n;
    c->k_built = false;
    if (n == 0) return RH_OK;
    c->k_mag = mag;
    c->kg_pad = ((c->nwords + 63) / 64) * 64 + 64;
    uint64_t *code_in = nullptr;
    int32_t *idx_in = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    auto cleanup = [&]() { (void)hipFree(code_in); (void)hipFree(idx_in); (void)hipFree(tmp); };
#define KH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rh_set_error("%s: %s", #x, hipGetErrorString(e_)); cleanup(); return RH_E_NODEVICE; } } while (0)
    KH(hipMalloc((void **)&code_in, sizeof(uint64_t) * (size_t)n));
    KH(hipMalloc((void **)&idx_in, sizeof(int32_t) * (size_t)n));
    KH(hipMalloc((void **)&c->oct_code, sizeof(uint64_t) * (size_t)n));
    KH(hipMalloc((void **)&c->oct_perm, sizeof(int32_t) * (size_t)n));
    KH(hipMalloc((void **)&c->oct_pos, sizeof(int32_t) * (size_t)n));
    KH(hipMalloc((void **)&c->oct_men, sizeof(uint64_t) * (size_t)c->nwords));
    KH(hipMalloc((void **)&c->oct_prefix, sizeof(int32_t) * (size_t)(c->nwords + 1)));
    KH(hipMalloc((void **)&c->fullk, sizeof(double) * 6 * (size_t)c->n_pad));
    KH(hipMalloc((void **)&c->kgb, sizeof(double) * 7 * (size_t)c->kg_pad));
    KH(hipMalloc((void **)&c->klist, sizeof(int32_t) * (size_t)c->nwords));
    KH(hipMalloc((void **)&c->kctr, sizeof(int32_t) * 2));
    KH(hipMemsetAsync(c->kctr, 0, sizeof(int32_t) * 2, c->stream));
    KH(hipMalloc((void **)&c->kflag, (size_t)c->n_pad));
    KH(hipMemsetAsync(c->kflag, 0, (size_t)c->n_pad, c->stream));
    KH(hipMemsetAsync(c->fullk, 0, sizeof(double) * 6 * (size_t)c->n_pad, c->stream));
    KH(hipMemsetAsync(c->kgb, 0, sizeof(double) * 7 * (size_t)c->kg_pad, c->stream));
    hipLaunchKernelGGL(morton_kernel, dim3(cdivk(n, 256)), dim3(256), 0, c->stream, d_xyz, n, lo[0], lo[1], lo[2], size, code_in,
                       idx_in);
    // stable LSD radix sort of (code, index) pairs by code: equal codes keep ascending indices = sort by (code, index)
    KH(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, code_in, c->oct_code, idx_in, c->oct_perm, (int)n, 0, 63, c->stream));
    KH(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 1));
    KH(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, code_in, c->oct_code, idx_in, c->oct_perm, (int)n, 0, 63, c->stream));
    hipLaunchKernelGGL(invert_perm_kernel, dim3(cdivk(n, 256)), dim3(256), 0, c->stream, c->oct_perm, n, c->oct_pos);
    KH(hipGetLastError());
    int rc = rhk_transpose_aos(c, d_xyz, d_nrm, n, c->oct_perm, n, c->fullk, c->n_pad);
    if (rc == RH_OK) rc = rhk_group_bounds_of(c, c->fullk, c->n_pad, n, c->nwords, c->kgb, c->kg_pad);
    if (rc != RH_OK) { cleanup(); return rc; }
    KH(hipStreamSynchronize(c->stream));
#undef KH
    cleanup();
    c->k_built = true;
    return RH_OK;
}

int rhk_korder_build_f32(rh_cloud *c)
{
    if (!c->k_built) return RH_OK;
    RH_HIP(hipMalloc((void **)&c->fullk32, sizeof(float) * 6 * (size_t)c->n_pad));
    hipLaunchKernelGGL(to_float_k, dim3(cdivk(6 * c->n_pad, 256)), dim3(256), 0, c->stream, c->fullk, 6 * c->n_pad, c->fullk32);
    RH_HIP(hipGetLastError());
    return RH_OK;
}

// the Morton-order enabled bits follow `enabled` lazily: whoever rewrites `enabled` wholesale clears k_men_valid
int rhk_korder_sync_enabled(rh_cloud *c)
{
    if (!c->k_built || c->k_men_valid) return RH_OK;
    RH_TRY(rhk_oct_gather_enabled(c));
    c->k_men_valid = true;
    return RH_OK;
}

// which scan a refit takes: the culled one from RH_KREFIT_MIN points on (below it the whole cloud is a few
// microseconds of streaming and two launches cost more); RH_REFIT_PATH=scan|culled forces one (tests, A/B)
bool rhk_refit_is_culled(const rh_cloud *c)
{
    if (!c->k_built) return false;
    const int64_t e = rh_opt_int(c, RH_OPT_REFIT_PATH, RH_REFIT_PATH_AUTO);   // rh_set_option(.., "refit_path", ..)
    if (e == RH_REFIT_PATH_SCAN) return false;
    if (e == RH_REFIT_PATH_CULLED) return true;
    return c->n >= RH_KREFIT_MIN;
}

template <bool F32, typename T, typename PREP>
static int launch_refitk(rh_cloud *c, const T *pts, const PREP &PX, const rh_prep &P, int kind, double eps, double cosa, bool apply)
{
    RH_TRY(rhk_korder_sync_enabled(c));
    const int64_t ng = c->nwords;
    const dim3 g1(cdivk(ng, 256)), blk(256);
    // pass 2 learns the list length on the device: a grid that covers a long list with a few groups per wave
    int64_t b2 = (ng + 3) / 4;
    if (b2 > 2048) b2 = 2048;
    const dim3 g2((unsigned)b2);
    const int ap = apply ? 1 : 0;
    const int dbg = rh_opt_on(c, RH_OPT_KREFIT_DBG) ? 1 : 0;
#define RH_K(K)                                                                                                              \
    do {                                                                                                                     \
        hipLaunchKernelGGL((refitk_boxes_kernel<K, F32>), g1, blk, 0, c->stream, c->kgb, c->kg_pad, ng, c->n, c->oct_men, P, eps, \
                           c->k_mag, c->klist, c->kctr);                                                                     \
        if (dbg) {                                                                                                           \
            int32_t cnt = 0;                                                                                                 \
            (void)hipMemcpyAsync(&cnt, c->kctr, sizeof cnt, hipMemcpyDeviceToHost, c->stream);                               \
            (void)hipStreamSynchronize(c->stream);                                                                           \
            fprintf(stderr, "[refitk] kind %d: %d of %lld groups survive the box test\n", kind, cnt, (long long)ng);         \
        }                                                                                                                    \
        hipLaunchKernelGGL((refitk_groups_kernel<K, F32, T, PREP>), g2, blk, 0, c->stream, pts, c->n_pad, c->n, c->oct_perm,  \
                           c->oct_men, ap, PX, eps, cosa, c->klist, c->kctr, c->kflag);                                      \
        hipLaunchKernelGGL(refitk_flags_kernel, dim3((unsigned)c->nblocks), blk, 0, c->stream, c->kflag, ng, c->refit_mask,  \
                           c->kctr, c->block_sums);                                                                          \
    } while (0)
    switch (kind) {
    case RH_PLANE: RH_K(RH_PLANE); break;
    case RH_SPHERE: RH_K(RH_SPHERE); break;
    case RH_CYLINDER: RH_K(RH_CYLINDER); break;
    case RH_CONE: RH_K(RH_CONE); break;
    default: rh_set_error("unknown shape kind %d", kind); return RH_E_INVALID;
    }
#undef RH_K
    RH_HIP(hipGetLastError());
    return RH_OK;
}

int rhk_refitk_mask(rh_cloud *c, const rh_prep &P, int kind, double eps, double cosa, bool apply)
{
    if (c->nwords == 0) return RH_OK;
    return launch_refitk<false, double, rh_prep>(c, c->fullk, P, P, kind, eps, cosa, apply);
}

// Float32 cloud: `prepf` is the float record of the shape (rhdev32::rh_prepf), P its binary64 record for the box tests
int rhk_refitk_mask_f32(rh_cloud *c, const void *prepf, const rh_prep &P, int kind, double eps, double cosa, bool apply)
{
    if (c->nwords == 0) return RH_OK;
    return launch_refitk<true, float, rh_prepf>(c, c->fullk32, *(const rh_prepf *)prepf, P, kind, eps, cosa, apply);
}
