// cc.hip -- parameter-space bitmap + largest connected component
// (src/parameterspacebitmap.jl:12-60, 69-109; dead code in the reference, pinned by
// test/parameterspacebitmap.jl).  The labelling replaces Images.label_components with a
// lock-free union-find on the pixel grid: every set pixel is united with its forward
// neighbours, the root of a component is its smallest column-major linear index (= the pixel
// label_components meets first, so "first largest component" ties break the same way).
#include <math.h>
#include <string.h>

#include <vector>

#include "rh_internal.h"

namespace {

__device__ __forceinline__ int32_t ld(const int32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ int32_t uf_find(const int32_t *L, int32_t i)
{
    int32_t p = ld(&L[i]);
    while (p != i) { i = p; p = ld(&L[i]); }
    return i;
}

__device__ void uf_unite(int32_t *L, int32_t a, int32_t b)
{
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a > b) { const int32_t t = a; a = b; b = t; }
        const int32_t old = atomicMin(&L[b], a);   // hang the larger root under the smaller
        if (old == b) return;
        b = old;                                    // somebody re-parented b meanwhile: retry from there
    }
}

__global__ void cc_init_kernel(const uint8_t *__restrict__ bm, int32_t npx, int32_t *__restrict__ L,
                               int32_t *__restrict__ size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    L[i] = bm[i] ? i : -1;
    size[i] = 0;
}

__global__ void cc_union_kernel(const uint8_t *__restrict__ bm, int32_t xs, int32_t ys, int conn8, int32_t *L)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= xs * ys || !bm[i]) return;
    const int x = i % xs, y = i / xs;
    if (x + 1 < xs && bm[i + 1]) uf_unite(L, i, i + 1);
    if (y + 1 < ys && bm[i + xs]) uf_unite(L, i, i + xs);
    if (conn8 && y + 1 < ys) {
        if (x + 1 < xs && bm[i + xs + 1]) uf_unite(L, i, i + xs + 1);
        if (x > 0 && bm[i + xs - 1]) uf_unite(L, i, i + xs - 1);
    }
}

__global__ void cc_flatten_kernel(int32_t npx, int32_t *L, int32_t *__restrict__ size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    if (ld(&L[i]) < 0) return;
    const int32_t r = uf_find(L, i);
    L[i] = r;                        // safe: r is a root, roots never change in this launch
    atomicAdd(&size[r], 1);
}

// key = size << 32 | ~root : max key = largest component, smallest root on ties
__global__ void cc_best_kernel(int32_t npx, const int32_t *__restrict__ size, unsigned long long *best)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const int32_t s = size[i];
    if (s > 0) atomicMax(best, ((unsigned long long)s << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i));
}

__global__ void cc_mask_kernel(int32_t npx, const int32_t *__restrict__ L, const unsigned long long *best,
                               uint64_t *__restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long key = *best;
    const int32_t root = key ? (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu)) : -2;
    const bool in = i < npx && L[i] == root;
    const uint64_t b = __builtin_amdgcn_ballot_w64(in);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < (npx + 63) / 64) mask[i >> 6] = b;
}

inline double julia_round(double x) { return nearbyint(x); }   // RoundNearest, ties to even

}  // namespace

extern "C" int rh_largestconncomp(const uint8_t *bitmap, int32_t xs, int32_t ys, int32_t conn8, int device,
                                  int64_t *out, int64_t cap, int64_t *n_out)
{
    if (!bitmap || !n_out || xs < 0 || ys < 0 || cap < 0 || (cap > 0 && !out)) {
        rh_set_error("rh_largestconncomp: bad arguments");
        return RH_E_INVALID;
    }
    *n_out = 0;
    const int64_t npx64 = (int64_t)xs * ys;
    if (npx64 == 0) return RH_OK;
    if (npx64 > 0x7FFFFFF0) { rh_set_error("rh_largestconncomp: bitmap too large"); return RH_E_INVALID; }
    int ndev = 0;
    RH_TRY(rh_device_count(&ndev));
    if (ndev <= 0) { rh_set_error("no HIP device is visible; libransac_hip has no CPU fallback"); return RH_E_NODEVICE; }
    if (device < 0 || device >= ndev) { rh_set_error("device %d out of range", device); return RH_E_INVALID; }
    RH_HIP(hipSetDevice(device));
    const int32_t npx = (int32_t)npx64;
    const int64_t nwords = (npx64 + 63) / 64;
    const int64_t nb = (nwords + RH_WORDS_PER_BLOCK - 1) / RH_WORDS_PER_BLOCK;
    uint8_t *d_bm = nullptr;
    int32_t *d_L = nullptr, *d_size = nullptr, *d_bs = nullptr, *d_total = nullptr;
    unsigned long long *d_best = nullptr;
    uint64_t *d_mask = nullptr;
    int64_t *d_idx = nullptr;
    hipStream_t stream = nullptr;
    int rc = RH_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_bm); (void)hipFree(d_L); (void)hipFree(d_size); (void)hipFree(d_bs); (void)hipFree(d_total);
        (void)hipFree(d_best); (void)hipFree(d_mask); (void)hipFree(d_idx);
        if (stream) (void)hipStreamDestroy(stream);
    };
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rh_set_error("%s: %s", #x, hipGetErrorString(e_)); cleanup(); return RH_E_NODEVICE; } } while (0)
    CKH(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    CKH(hipMalloc((void **)&d_bm, (size_t)npx));
    CKH(hipMalloc((void **)&d_L, sizeof(int32_t) * (size_t)npx));
    CKH(hipMalloc((void **)&d_size, sizeof(int32_t) * (size_t)npx));
    CKH(hipMalloc((void **)&d_bs, sizeof(int32_t) * (size_t)(nb + 2)));
    CKH(hipMalloc((void **)&d_total, sizeof(int32_t)));
    CKH(hipMalloc((void **)&d_best, sizeof(unsigned long long)));
    CKH(hipMalloc((void **)&d_mask, sizeof(uint64_t) * (size_t)nwords));
    CKH(hipMalloc((void **)&d_idx, sizeof(int64_t) * (size_t)npx));
    CKH(hipMemcpyAsync(d_bm, bitmap, (size_t)npx, hipMemcpyHostToDevice, stream));
    CKH(hipMemsetAsync(d_best, 0, sizeof(unsigned long long), stream));
    {
        const dim3 grid((unsigned)((npx + 255) / 256)), blk(256);
        hipLaunchKernelGGL(cc_init_kernel, grid, blk, 0, stream, d_bm, npx, d_L, d_size);
        hipLaunchKernelGGL(cc_union_kernel, grid, blk, 0, stream, d_bm, xs, ys, conn8 ? 1 : 0, d_L);
        hipLaunchKernelGGL(cc_flatten_kernel, grid, blk, 0, stream, npx, d_L, d_size);
        hipLaunchKernelGGL(cc_best_kernel, grid, blk, 0, stream, npx, d_size, d_best);
        const dim3 gridm((unsigned)((nwords * 64 + 255) / 256));
        hipLaunchKernelGGL(cc_mask_kernel, gridm, blk, 0, stream, npx, d_L, d_best, d_mask);
        CKH(hipGetLastError());
    }
    rc = rhk_compact_generic(stream, d_mask, nwords, d_bs, d_idx, npx, d_total);
    if (rc != RH_OK) { cleanup(); return rc; }
    int32_t total = 0;
    CKH(hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, stream));
    CKH(hipStreamSynchronize(stream));
    *n_out = total;
    if (total > cap) {
        rh_set_error("rh_largestconncomp: component has %d pixels, capacity %lld", total, (long long)cap);
        cleanup();
        return RH_E_CAPACITY;
    }
    if (total > 0) {
        std::vector<int64_t> tmp((size_t)total);
        CKH(hipMemcpyAsync(tmp.data(), d_idx, sizeof(int64_t) * (size_t)total, hipMemcpyDeviceToHost, stream));
        CKH(hipStreamSynchronize(stream));
        for (int32_t i = 0; i < total; i++) out[i] = tmp[(size_t)i] - 1;   // compaction is 1-based
    }
#undef CKH
    cleanup();
    return RH_OK;
}

// bitmapparameters: parameterspacebitmap.jl:12-46.  Sequential "first writer wins" per pixel;
// O(n) on the host (the live reference never produces 2-D parameters, so there is no device
// producer to fuse with yet).
extern "C" int rh_bitmapparameters(const double *prm2, const uint8_t *compat, const int64_t *idsource, int64_t n,
                                   double beta, int32_t *xs_out, int32_t *ys_out, double *betax, double *betay,
                                   uint8_t *bitmap, int64_t *idxmap)
{
    if (!prm2 || !compat || n <= 0 || !xs_out || !ys_out || !betax || !betay) {
        rh_set_error("rh_bitmapparameters: bad arguments");
        return RH_E_INVALID;
    }
    double mn[2] = { prm2[0], prm2[1] }, mx[2] = { prm2[0], prm2[1] };   // findAABB: utilities.jl:125-136
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < 2; j++) {
            const double a = prm2[2 * i + j];
            mn[j] = mn[j] > a ? a : mn[j];
            mx[j] = mx[j] < a ? a : mx[j];
        }
    const double minv[2] = { mn[0] - 0.1, mn[1] - 0.1 }, maxv[2] = { mx[0] + 0.1, mx[1] + 0.1 };
    const int64_t xs = (int64_t)julia_round((maxv[0] - minv[0]) / beta), ys = (int64_t)julia_round((maxv[1] - minv[1]) / beta);
    if (!(xs > 0 && ys > 0)) { rh_set_error("max-min should be positive. xs: %lld, ys: %lld", (long long)xs, (long long)ys); return RH_E_INVALID; }
    const double bx = (maxv[0] - minv[0]) / (double)xs, by = (maxv[1] - minv[1]) / (double)ys;
    *xs_out = (int32_t)xs; *ys_out = (int32_t)ys; *betax = bx; *betay = by;
    if (!bitmap || !idxmap) return RH_OK;
    memset(bitmap, 0, (size_t)(xs * ys));
    memset(idxmap, 0, sizeof(int64_t) * (size_t)(xs * ys));
    for (int64_t i = 0; i < n; i++) {
        if (!compat[i]) continue;
        const int64_t xp = (int64_t)ceil((prm2[2 * i] - minv[0]) / bx), yp = (int64_t)ceil((prm2[2 * i + 1] - minv[1]) / by);
        if (xp != 0 && yp != 0 && xp != xs && yp != ys) {
            const int64_t li = (xp - 1) + xs * (yp - 1);
            if (!bitmap[li]) { bitmap[li] = 1; idxmap[li] = idsource ? idsource[i] : i + 1; }
        }
    }
    return RH_OK;
}
