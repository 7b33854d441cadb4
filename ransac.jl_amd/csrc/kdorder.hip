// kdorder.hip -- the internal order of subset 1 on the device: the leaves of a balanced k-d tree, 64 points each, which is
// what gives the culled score kernel its compact groups (cloud.hip builds the same tree on host threads -- std::nth_element
// recursion, 70 ms at 312 500 points, 300 ms at 1.56 M -- and keeps that form behind RH_KD_HOST=1 as the A/B).
//
// The tree's SHAPE depends on the point count alone (a node of cnt > 64 points gives its left child ((cnt / 64 + 1) / 2) * 64
// of them), so the host lays the levels out and the device only orders: per level
//   1. the bounding box of every node that still splits (finite coordinates only; wave-uniform fast path + atomics),
//   2. a 64-bit key per position -- node number << 32 | the point's coordinate along the node's widest axis as an ordered
//      32-bit pattern (binary32 of the coordinate: NaN last, like the host's comparator) -- positions of finished nodes
//      keep their place,
//   3. one stable radix sort of (key, point) pairs over the whole subset (hipCUB): inside a node the points are then
//      ascending along its axis, and the node's children are simply its left and right part.
// log2(s / 64) levels of about 0.1-0.3 ms.  The order differs from the host's in ties and where two coordinates share
// a binary32 value -- any order gives the same counts and masks (they are un-permuted on the way out; every parity test
// holds whatever the order), only the boxes' tightness is at stake, and that is measured.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "rh_internal.h"

namespace {

inline unsigned cdivq(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ uint32_t ord32(float v)
{
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;   // NaN sorts last
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unord32(uint32_t u)
{
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, b);
}

// subset position j -> its coordinates as three float planes; max |coordinate| and max |normal component| over the finite
// values of the subset (the margins of the binary32 classifier and the slack of the box tests scale with them) as the bit
// patterns of non-negative doubles (they order like the doubles)
__global__ void __launch_bounds__(256)
kd_gather_kernel(const double *__restrict__ xyz, const double *__restrict__ nrm, const int32_t *__restrict__ idx0, int64_t s,
                 float *__restrict__ px, float *__restrict__ py, float *__restrict__ pz, int32_t *__restrict__ ord,
                 unsigned long long *__restrict__ mags)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double cm = 0.0, nm = 0.0;
    if (j < s) {
        const int64_t i = idx0[j];
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        px[j] = (float)x; py[j] = (float)y; pz[j] = (float)z;
        ord[j] = (int32_t)j;
        const double v[6] = { x, y, z, nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2] };
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double a = fabs(v[k]);
            if (a - a == 0) { if (k < 3) cm = a > cm ? a : cm; else nm = a > nm ? a : nm; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(cm, off), on = __shfl_xor(nm, off);
        cm = oc > cm ? oc : cm;
        nm = on > nm ? on : nm;
    }
    if ((threadIdx.x & 63) == 0) {
        if (cm > 0) atomicMax(&mags[0], __builtin_bit_cast(unsigned long long, cm));
        if (nm > 0) atomicMax(&mags[1], __builtin_bit_cast(unsigned long long, nm));
    }
}

// the segment (node of the current level) of position i: last k with seg_lo[k] <= i
__device__ __forceinline__ int seg_of(const int32_t *__restrict__ seg_lo, int nseg, int32_t i)
{
    int lo = 0, hi = nseg;   // seg_lo[0] = 0 <= i < seg_lo[nseg] = s
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_lo[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// box[seg * 6 + {0,1,2}] = min, {3,4,5} = max of the ordered patterns of the finite coordinates (init: 0xffffffff / 0)
__global__ void __launch_bounds__(256)
kd_box_kernel(const float *__restrict__ px, const float *__restrict__ py, const float *__restrict__ pz, const int32_t *__restrict__ ord,
              int64_t s, const int32_t *__restrict__ seg_lo, const uint8_t *__restrict__ seg_active, int nseg, uint32_t *__restrict__ box)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < s;
    const int seg = in ? seg_of(seg_lo, nseg, (int32_t)i) : -1;
    const bool act = in && seg_active[seg] != 0;
    uint32_t mn[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, mx[3] = { 0u, 0u, 0u };
    if (act) {
        const int32_t j = ord[i];
        const float v[3] = { px[j], py[j], pz[j] };
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (v[k] - v[k] == 0.0f) { mn[k] = mx[k] = ord32(v[k]); }
    }
    // (the wave's positions are consecutive: in the upper levels they all belong to one node)
    const int seg0 = __shfl(seg, 0);
    const bool uniform = __builtin_amdgcn_ballot_w64(seg != seg0) == 0;
    if (uniform) {
        if (seg0 < 0 || !__shfl((int)act, 0)) return;
#pragma unroll
        for (int k = 0; k < 3; k++)
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t a = __shfl_xor(mn[k], off), b = __shfl_xor(mx[k], off);
                mn[k] = a < mn[k] ? a : mn[k];
                mx[k] = b > mx[k] ? b : mx[k];
            }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) { atomicMin(&box[seg0 * 6 + k], mn[k]); atomicMax(&box[seg0 * 6 + 3 + k], mx[k]); }
        }
    } else if (act) {
#pragma unroll
        for (int k = 0; k < 3; k++) { atomicMin(&box[seg * 6 + k], mn[k]); atomicMax(&box[seg * 6 + 3 + k], mx[k]); }
    }
}

__global__ void __launch_bounds__(256)
kd_key_kernel(const float *__restrict__ px, const float *__restrict__ py, const float *__restrict__ pz, const int32_t *__restrict__ ord,
              int64_t s, const int32_t *__restrict__ seg_lo, const uint8_t *__restrict__ seg_active, int nseg,
              const uint32_t *__restrict__ box, uint64_t *__restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s) return;
    const int seg = seg_of(seg_lo, nseg, (int32_t)i);
    uint32_t sub = (uint32_t)i;   // a finished node: the position itself (ascending inside the node, like before)
    if (seg_active[seg] != 0) {
        // widest axis of the node's box (the first of equally wide ones, like the host's `e > best`); an axis without a
        // finite coordinate has extent -1
        int ax = 0;
        float best = -2.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t lo = box[seg * 6 + k], hi = box[seg * 6 + 3 + k];
            const float e = lo <= hi ? unord32(hi) - unord32(lo) : -1.0f;
            if (e > best) { best = e; ax = k; }
        }
        const int32_t j = ord[i];
        sub = ord32(ax == 0 ? px[j] : (ax == 1 ? py[j] : pz[j]));
    }
    keys[i] = ((uint64_t)(uint32_t)seg << 32) | sub;
}

__global__ void kd_fill_box_kernel(uint32_t *__restrict__ box, int nseg)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nseg * 6) box[t] = (t % 6) < 3 ? 0xffffffffu : 0u;
}

// sub_perm[i] = subset position of internal position i; sub_idx0[i] = its 0-based index in the cloud
__global__ void kd_finish_kernel(const int32_t *__restrict__ ord, const int32_t *__restrict__ idx0, int64_t s,
                                 int32_t *__restrict__ sub_perm, int32_t *__restrict__ sub_idx0)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s) return;
    const int32_t j = ord[i];
    sub_perm[i] = j;
    sub_idx0[i] = idx0[j];
}

struct KdNode { int32_t lo, hi; };

// ---- findAABB (utilities.jl:125-136) of the uploaded cloud: per axis the smallest and the largest value that is not a NaN
// (infinities count, like in the host loop this replaces), and the largest finite magnitude.  Doubles as ordered 64-bit
// patterns; out[0..2] min (init all ones), out[3..5] max (init 0), out[6] the magnitude's bits (init 0).
__device__ __forceinline__ unsigned long long ord64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__global__ void __launch_bounds__(256)
aabb_kernel(const double *__restrict__ xyz, int64_t n, unsigned long long *__restrict__ out)
{
    unsigned long long mn[3] = { ~0ULL, ~0ULL, ~0ULL }, mx[3] = { 0ULL, 0ULL, 0ULL };
    double mag = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double a = xyz[3 * i + k];
            if (a == a) {
                const unsigned long long o = ord64(a);
                mn[k] = o < mn[k] ? o : mn[k];
                mx[k] = o > mx[k] ? o : mx[k];
                const double m = fabs(a);
                if (m - m == 0 && m > mag) mag = m;
            }
        }
    }
    unsigned long long mb = __builtin_bit_cast(unsigned long long, mag);
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const unsigned long long a = __shfl_xor(mn[k], off), b = __shfl_xor(mx[k], off);
            mn[k] = a < mn[k] ? a : mn[k];
            mx[k] = b > mx[k] ? b : mx[k];
        }
        const unsigned long long c = __shfl_xor(mb, off);
        mb = c > mb ? c : mb;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { atomicMin(&out[k], mn[k]); atomicMax(&out[3 + k], mx[k]); }
        atomicMax(&out[6], mb);
    }
}

}  // namespace

// lo / hi: per axis the extreme non-NaN values (lo > hi bit patterns: no such value), mag: largest finite |coordinate|
int rhk_cloud_aabb(rh_cloud *c, const double *d_xyz, int64_t n, double lo[3], double hi[3], bool has[3], double *mag)
{
    for (int k = 0; k < 3; k++) { lo[k] = hi[k] = 0; has[k] = false; }
    *mag = 0;
    if (n <= 0) return RH_OK;
    unsigned long long *d = nullptr, h[7];
    RH_HIP(hipMalloc((void **)&d, sizeof h));
    for (int k = 0; k < 7; k++) h[k] = k < 3 ? ~0ULL : 0ULL;
    hipError_t e = hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(aabb_kernel, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, c->stream, d_xyz, n, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { rh_set_error("rhk_cloud_aabb: %s", hipGetErrorString(e)); return RH_E_NODEVICE; }
    auto un = [](unsigned long long u) { const unsigned long long b = (u >> 63) ? (u & 0x7fffffffffffffffULL) : ~u; return __builtin_bit_cast(double, b); };
    for (int k = 0; k < 3; k++) {
        has[k] = h[k] <= h[3 + k];
        if (has[k]) { lo[k] = un(h[k]); hi[k] = un(h[3 + k]); }
    }
    *mag = __builtin_bit_cast(double, h[6]);
    return RH_OK;
}

// d_xyz / d_nrm: the cloud as uploaded (AoS, n x 3); d_idx0: subset position -> 0-based cloud index (s entries).  Fills
// c->sub_perm, c->sub_idx0, c->coord_mag, c->nrm_mag.  Synchronises the stream once (the two magnitudes come back).
int rhk_kd_order(rh_cloud *c, const double *d_xyz, const double *d_nrm, const int32_t *d_idx0)
{
    const int64_t s = c->s;
    if (s <= 0) return RH_OK;
    if (s > (int64_t)0x7fffffff) { rh_set_error("rhk_kd_order: subset of %lld points", (long long)s); return RH_E_INVALID; }
    float *pf = nullptr;
    int32_t *ord[2] = { nullptr, nullptr };
    uint64_t *keys[2] = { nullptr, nullptr };
    uint32_t *box = nullptr;
    int32_t *seg_lo = nullptr;
    uint8_t *seg_act = nullptr;
    unsigned long long *mags = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    auto cleanup = [&]() {
        (void)hipFree(pf); (void)hipFree(ord[0]); (void)hipFree(ord[1]); (void)hipFree(keys[0]); (void)hipFree(keys[1]); (void)hipFree(box);
        (void)hipFree(seg_lo); (void)hipFree(seg_act); (void)hipFree(mags); (void)hipFree(tmp);
    };
#define QH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rh_set_error("%s: %s", #x, hipGetErrorString(e_)); cleanup(); return RH_E_NODEVICE; } } while (0)
    // the levels: every node of a level in position order, finished ones (<= 64 points) included
    std::vector<std::vector<KdNode>> levels;
    {
        std::vector<KdNode> cur(1, KdNode{ 0, (int32_t)s });
        for (;;) {
            bool any = false;
            for (const KdNode &nd : cur) any = any || nd.hi - nd.lo > 64;
            if (!any) break;
            levels.push_back(cur);
            std::vector<KdNode> nxt;
            nxt.reserve(cur.size() * 2);
            for (const KdNode &nd : cur) {
                const int32_t cnt = nd.hi - nd.lo;
                if (cnt <= 64) { nxt.push_back(nd); continue; }
                const int32_t nl = ((cnt / 64 + 1) / 2) * 64;
                nxt.push_back(KdNode{ nd.lo, nd.lo + nl });
                nxt.push_back(KdNode{ nd.lo + nl, nd.hi });
            }
            cur.swap(nxt);
        }
    }
    size_t max_seg = 1;
    for (const auto &lv : levels) max_seg = std::max(max_seg, lv.size());
    QH(hipMalloc((void **)&pf, sizeof(float) * 3 * (size_t)s));
    QH(hipMalloc((void **)&ord[0], sizeof(int32_t) * (size_t)s));
    QH(hipMalloc((void **)&ord[1], sizeof(int32_t) * (size_t)s));
    QH(hipMalloc((void **)&keys[0], sizeof(uint64_t) * (size_t)s));
    QH(hipMalloc((void **)&keys[1], sizeof(uint64_t) * (size_t)s));
    QH(hipMalloc((void **)&box, sizeof(uint32_t) * 6 * max_seg));
    QH(hipMalloc((void **)&mags, sizeof(unsigned long long) * 2));
    QH(hipMemsetAsync(mags, 0, sizeof(unsigned long long) * 2, c->stream));
    float *px = pf, *py = pf + s, *pz = pf + 2 * s;
    const dim3 gs(cdivq(s, 256)), blk(256);
    hipLaunchKernelGGL(kd_gather_kernel, gs, blk, 0, c->stream, d_xyz, d_nrm, d_idx0, s, px, py, pz, ord[0], mags);
    QH(hipGetLastError());
    QH(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys[0], keys[1], ord[0], ord[1], (int)s, 0, 64, c->stream));
    QH(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 1));
    // every level's node starts / flags, uploaded once (pageable source: the arrays live until the final wait)
    std::vector<int32_t> h_lo;
    std::vector<uint8_t> h_act;
    std::vector<size_t> off_lo, off_act;
    for (const auto &lv : levels) {
        off_lo.push_back(h_lo.size());
        off_act.push_back(h_act.size());
        for (const KdNode &nd : lv) { h_lo.push_back(nd.lo); h_act.push_back(nd.hi - nd.lo > 64 ? 1 : 0); }
        h_lo.push_back((int32_t)s);
    }
    (void)hipFree(seg_lo); (void)hipFree(seg_act);
    seg_lo = nullptr; seg_act = nullptr;
    QH(hipMalloc((void **)&seg_lo, sizeof(int32_t) * std::max<size_t>(h_lo.size(), 1)));
    QH(hipMalloc((void **)&seg_act, std::max<size_t>(h_act.size(), 1)));
    if (!h_lo.empty()) QH(hipMemcpyAsync(seg_lo, h_lo.data(), sizeof(int32_t) * h_lo.size(), hipMemcpyHostToDevice, c->stream));
    if (!h_act.empty()) QH(hipMemcpyAsync(seg_act, h_act.data(), h_act.size(), hipMemcpyHostToDevice, c->stream));
    int cur = 0;
    for (size_t L = 0; L < levels.size(); L++) {
        const int nseg = (int)levels[L].size();
        const int32_t *lo_l = seg_lo + off_lo[L];
        const uint8_t *act_l = seg_act + off_act[L];
        hipLaunchKernelGGL(kd_fill_box_kernel, dim3(cdivq((int64_t)nseg * 6, 256)), blk, 0, c->stream, box, nseg);
        hipLaunchKernelGGL(kd_box_kernel, gs, blk, 0, c->stream, px, py, pz, ord[cur], s, lo_l, act_l, nseg, box);
        hipLaunchKernelGGL(kd_key_kernel, gs, blk, 0, c->stream, px, py, pz, ord[cur], s, lo_l, act_l, nseg, box, keys[0]);
        QH(hipGetLastError());
        int seg_bits = 1;
        while ((1 << seg_bits) < nseg) seg_bits++;
        QH(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys[0], keys[1], ord[cur], ord[cur ^ 1], (int)s, 0, 32 + seg_bits, c->stream));
        cur ^= 1;
    }
    hipLaunchKernelGGL(kd_finish_kernel, gs, blk, 0, c->stream, ord[cur], d_idx0, s, c->sub_perm, c->sub_idx0);
    QH(hipGetLastError());
    unsigned long long h_mags[2] = { 0, 0 };
    QH(hipMemcpyAsync(h_mags, mags, sizeof h_mags, hipMemcpyDeviceToHost, c->stream));
    QH(hipStreamSynchronize(c->stream));
#undef QH
    c->coord_mag = __builtin_bit_cast(double, h_mags[0]);
    c->nrm_mag = __builtin_bit_cast(double, h_mags[1]);
    cleanup();
    return RH_OK;
}
