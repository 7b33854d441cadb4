// octree.hip -- the reference's own octree: buildoctree / OctreeRefinery / needs_refinement / refine_data /
// iswithinrectangle / octreedepth (/root/reference/src/octree.jl:237-244, 158-177, 187-196, 212-230), findAABB
// (utilities.jl:125-136), getnthcell (octree.jl:11-22), RegionTrees' findleaf as samplepointcloud4! uses it
// (fitting.jl:397) and the enabled-cell gather of fitting.jl:405-407.
//
// The tree is what RANSACCloud carries as `pc.octree`.  It never influences a result of ransac() -- the constructor's
// levelweight / levelscore swap pins every sample to the root cell, which is why the loop keeps no such
// tree on its path and the fixed-behaviour mode samples from a linear Morton octree instead (korder.hip) -- but it is
// part of the API surface, so it exists here with the reference's geometry, quirks included:
//   * the root is Cell(minV, maxV, ...) and RegionTrees reads the second argument as WIDTHS: the root spans
//     [minV, minV + maxV] (octree.jl:240);
//   * a cell with more than 8 points splits at origin + widths / 2; a child is the cell's origin or the division with
//     width division - origin or origin + width - division, every one of those a floating-point operation; a child keeps
//     the parent's points with vmin < p <= vmax per axis, vmax = origin + width (Q3: points on a minimum face fall out;
//     rounding can put a point on a shared face into both neighbours or neither -- the eight tests are independent);
//   * more than 8 coincident points would refine for ever (Q17): the build stops at depth 48 and says so.
// The build is set-up (host, a queue of cells); what runs per sample in the reference -- gathering the enabled points of
// a cell -- is a device pass over the cell's index list against the cloud's enabled bits.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rh_internal.h"

struct rh_octree_node {
    double origin[3], widths[3], div[3];
    int32_t depth, parent, child[8];      // child[i + 2 j + 4 k]; -1: a leaf
    int64_t first, count;                  // the cell's points: idx[first .. first + count)
};

struct rh_octree {
    std::vector<rh_octree_node> nodes;
    std::vector<int64_t> idx;              // 1-based point indices, every cell's list in the order of its parent's
    int64_t n = 0;
    int overflow = 0;
    // device copy of the index lists (made on the first gather, for that cloud's device)
    int64_t *d_idx = nullptr;          // rh_octree_cell_enabled's device scratch (the requested cell's list, its enabled points, their number)
    int64_t d_cap = 0;
    int d_device = -1;
};

namespace {

// (T = the cloud's element type: on a Float32 cloud the reference computes origin + widths in Float32, octree.jl:187-196)
template <class T>
inline bool within(const double o[3], const double w[3], const T *p)
{
    for (int i = 0; i < 3; i++) {
        const T vmin = (T)o[i], vmax = (T)o[i] + (T)w[i];   // vertices(rect)[1,1,1], [2,2,2]
        if (!(vmin < p[i])) return false;
        if (!(vmax >= p[i])) return false;
    }
    return true;
}

// block-wide ordered compaction of the cell's enabled points (one block: cells are small except near the root)
__global__ void __launch_bounds__(1024)
cell_enabled_kernel(const int64_t *__restrict__ idx, int64_t count, const uint64_t *__restrict__ enabled, int64_t npoints,
                    int64_t *__restrict__ out, int64_t cap, int64_t *__restrict__ n_out)
{
    __shared__ int64_t base;
    __shared__ int32_t wsum[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < count; i0 += 1024) {
        const int64_t i = i0 + threadIdx.x;
        bool on = false;
        int64_t id = 0;
        if (i < count) {
            id = idx[i];
            on = id >= 1 && id <= npoints && ((enabled[(id - 1) >> 6] >> ((id - 1) & 63)) & 1ULL);
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(on);
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int64_t off = base;
        int32_t tot = 0;
        for (int w = 0; w < 16; w++) { if (w < wv) off += wsum[w]; tot += wsum[w]; }
        if (on) {
            const int64_t pos = off + __popcll(m & ((1ULL << lane) - 1ULL));
            if (pos < cap) out[pos] = id;
        }
        __syncthreads();
        if (threadIdx.x == 0) base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = base;
}

}  // namespace

// the build in the cloud's element type T: every corner, division and width is a T operation; the nodes keep the values
// widened to double (exact), so that findleaf / node_info serve both kinds of tree
template <class T>
static int octree_build_impl(const T *xyz, int64_t n, rh_octree **out)
{
    if (!out || n < 0 || (n > 0 && !xyz)) { rh_set_error("rh_octree_build: bad arguments"); return RH_E_INVALID; }
    *out = nullptr;
    rh_octree *t = new (std::nothrow) rh_octree();
    if (!t) { rh_set_error("out of host memory"); return RH_E_NOMEM; }
    t->n = n;
    // findAABB: both corners start at the first point and move by plain comparisons
    T lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
    for (int j = 0; j < 3 && n > 0; j++) { lo[j] = xyz[j]; hi[j] = xyz[j]; }
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            const T a = xyz[3 * i + j];
            lo[j] = lo[j] > a ? a : lo[j];
            hi[j] = hi[j] < a ? a : hi[j];
        }
    rh_octree_node root;
    memset(&root, 0, sizeof root);
    for (int j = 0; j < 3; j++) { root.origin[j] = lo[j]; root.widths[j] = hi[j]; }   // Cell(minV, maxV): maxV taken as the widths
    root.depth = 1;
    root.parent = -1;
    for (int k = 0; k < 8; k++) root.child[k] = -1;
    root.first = 0;
    root.count = n;
    try {
        t->idx.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) t->idx[(size_t)i] = i + 1;            // collect(1:l)
        t->nodes.push_back(root);
        for (size_t cur = 0; cur < t->nodes.size(); cur++) {                  // (the order of refinement does not change the tree)
            if (!(t->nodes[cur].count > 8)) continue;                          // needs_refinement
            if (t->nodes[cur].depth >= 48) { t->overflow = 1; continue; }
            rh_octree_node par = t->nodes[cur];
            for (int j = 0; j < 3; j++) par.div[j] = (T)par.origin[j] + (T)par.widths[j] / 2;
            for (int ci = 0; ci < 8; ci++) {
                rh_octree_node ch;
                memset(&ch, 0, sizeof ch);
                for (int j = 0; j < 3; j++) {
                    const bool upper = (ci >> j) & 1;
                    ch.origin[j] = upper ? par.div[j] : par.origin[j];
                    ch.widths[j] = upper ? (T)par.origin[j] + (T)par.widths[j] - (T)par.div[j] : (T)par.div[j] - (T)par.origin[j];
                }
                ch.depth = par.depth + 1;
                ch.parent = (int32_t)cur;
                for (int k = 0; k < 8; k++) ch.child[k] = -1;
                ch.first = (int64_t)t->idx.size();
                for (int64_t k = 0; k < par.count; k++) {                      // refine_data: every point of the parent is re-tested
                    const int64_t id = t->idx[(size_t)(par.first + k)];
                    if (within(ch.origin, ch.widths, xyz + 3 * (id - 1))) t->idx.push_back(id);
                }
                ch.count = (int64_t)t->idx.size() - ch.first;
                par.child[ci] = (int32_t)t->nodes.size();
                t->nodes.push_back(ch);
                if (t->nodes.size() > (size_t)0x7ffffff0) { rh_set_error("rh_octree_build: too many cells"); delete t; return RH_E_CAPACITY; }
            }
            t->nodes[cur] = par;
        }
    } catch (const std::bad_alloc &) {
        delete t;
        rh_set_error("rh_octree_build: out of host memory");
        return RH_E_NOMEM;
    }
    *out = t;
    return RH_OK;
}

extern "C" int rh_octree_build(const double *xyz, int64_t n, rh_octree **out) { return octree_build_impl<double>(xyz, n, out); }
extern "C" int rh_octree_build_f32(const float *xyz, int64_t n, rh_octree **out) { return octree_build_impl<float>(xyz, n, out); }

extern "C" int rh_octree_destroy(rh_octree *t)
{
    if (!t) return RH_OK;
    if (t->d_idx) {
        if (t->d_device >= 0) (void)hipSetDevice(t->d_device);
        (void)hipFree(t->d_idx);
    }
    delete t;
    return RH_OK;
}

extern "C" int rh_octree_info(const rh_octree *t, int32_t *n_nodes, int32_t *depth, int32_t *overflow)
{
    if (!t) { rh_set_error("rh_octree_info: NULL tree"); return RH_E_INVALID; }
    if (n_nodes) *n_nodes = (int32_t)t->nodes.size();
    if (depth) {   // octreedepth: the deepest leaf
        int32_t d = t->nodes.empty() ? 0 : t->nodes[0].depth;
        for (const rh_octree_node &nd : t->nodes)
            if (nd.child[0] < 0 && nd.depth > d) d = nd.depth;
        *depth = d;
    }
    if (overflow) *overflow = t->overflow;
    return RH_OK;
}

extern "C" int rh_octree_findleaf(const rh_octree *t, const double *p, int32_t *node_out)
{
    if (!t || !p || !node_out || t->nodes.empty()) { rh_set_error("rh_octree_findleaf: bad arguments"); return RH_E_INVALID; }
    int32_t cur = 0;
    while (t->nodes[(size_t)cur].child[0] >= 0) {
        const rh_octree_node &nd = t->nodes[(size_t)cur];
        const int ci = (p[0] >= nd.div[0] ? 1 : 0) | (p[1] >= nd.div[1] ? 2 : 0) | (p[2] >= nd.div[2] ? 4 : 0);
        cur = nd.child[ci];
    }
    *node_out = cur;
    return RH_OK;
}

// getnthcell(c, n): the ancestor of cell `node` (or the cell itself) at depth n; *node_out = -1 is `nothing`
extern "C" int rh_octree_getnthcell(const rh_octree *t, int32_t node, int32_t n, int32_t *node_out)
{
    if (!t || !node_out || node < 0 || (size_t)node >= t->nodes.size()) { rh_set_error("rh_octree_getnthcell: bad arguments"); return RH_E_INVALID; }
    *node_out = -1;
    if (n < 1) return RH_OK;
    int32_t c = node;
    if (t->nodes[(size_t)c].depth == n) { *node_out = c; return RH_OK; }
    for (;;) {
        c = t->nodes[(size_t)c].parent;
        if (c < 0) return RH_OK;
        if (t->nodes[(size_t)c].depth == n) { *node_out = c; return RH_OK; }
    }
}

extern "C" int rh_octree_node_info(const rh_octree *t, int32_t node, double *origin3, double *widths3, int32_t *depth, int32_t *parent,
                                   int32_t *children8, int64_t *npoints)
{
    if (!t || node < 0 || (size_t)node >= t->nodes.size()) { rh_set_error("rh_octree_node_info: bad arguments"); return RH_E_INVALID; }
    const rh_octree_node &nd = t->nodes[(size_t)node];
    for (int j = 0; j < 3; j++) { if (origin3) origin3[j] = nd.origin[j]; if (widths3) widths3[j] = nd.widths[j]; }
    if (depth) *depth = nd.depth;
    if (parent) *parent = nd.parent;
    if (children8) for (int k = 0; k < 8; k++) children8[k] = nd.child[k];
    if (npoints) *npoints = nd.count;
    return RH_OK;
}

// cell.data.incellpoints
extern "C" int rh_octree_node_points(const rh_octree *t, int32_t node, int64_t *idx_out, int64_t cap)
{
    if (!t || node < 0 || (size_t)node >= t->nodes.size() || cap < 0 || (cap > 0 && !idx_out)) { rh_set_error("rh_octree_node_points: bad arguments"); return RH_E_INVALID; }
    const rh_octree_node &nd = t->nodes[(size_t)node];
    if (nd.count > cap) { rh_set_error("rh_octree_node_points: %lld points, capacity %lld", (long long)nd.count, (long long)cap); return RH_E_CAPACITY; }
    memcpy(idx_out, t->idx.data() + nd.first, sizeof(int64_t) * (size_t)nd.count);
    return RH_OK;
}

// enabled_inds = cell.data.incellpoints[pc.isenabled[cell.data.incellpoints]] (fitting.jl:405-407): the points of the cell
// that are enabled in cloud c, in the order of the cell's list -- a device pass over the list against the cloud's bits
extern "C" int rh_octree_cell_enabled(rh_cloud *c, rh_octree *t, int32_t node, int64_t *idx_out, int64_t cap, int64_t *n_out)
{
    if (!c || !t || !n_out || node < 0 || (size_t)node >= t->nodes.size() || cap < 0 || (cap > 0 && !idx_out)) { rh_set_error("rh_octree_cell_enabled: bad arguments"); return RH_E_INVALID; }
    if (t->n != c->n) { rh_set_error("rh_octree_cell_enabled: the tree was built over %lld points, the cloud holds %lld", (long long)t->n, (long long)c->n); return RH_E_INVALID; }
    *n_out = 0;
    const rh_octree_node &nd = t->nodes[(size_t)node];
    if (nd.count == 0) return RH_OK;
    RH_HIP(hipSetDevice(c->device));
    RH_TRY(rh_join_batches(c));
    // the cell's own slice of the index lists goes up (not the tree's whole multi-level list: n x depth x 8 bytes), into
    // scratch the tree keeps for the next call: [cell list | enabled list | count]
    const int64_t ocap = cap < nd.count ? cap : nd.count;
    const int64_t need = nd.count + (ocap > 0 ? ocap : 1) + 1;
    if (t->d_idx == nullptr || t->d_device != c->device || t->d_cap < need) {
        if (t->d_idx) { if (t->d_device >= 0) (void)hipSetDevice(t->d_device); (void)hipFree(t->d_idx); (void)hipSetDevice(c->device); }
        t->d_idx = nullptr;
        t->d_cap = 0;
        RH_HIP(hipMalloc((void **)&t->d_idx, sizeof(int64_t) * (size_t)need));
        t->d_cap = need;
        t->d_device = c->device;
    }
    int64_t *d_out = t->d_idx + nd.count, *d_n = d_out + (ocap > 0 ? ocap : 1);
    RH_HIP(hipMemcpyAsync(t->d_idx, t->idx.data() + nd.first, sizeof(int64_t) * (size_t)nd.count, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(cell_enabled_kernel, dim3(1), dim3(1024), 0, c->stream, t->d_idx, nd.count, c->enabled, c->n, d_out, ocap, d_n);
    int64_t total = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&total, d_n, sizeof total, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && total > 0 && total <= cap)
        e = hipMemcpy(idx_out, d_out, sizeof(int64_t) * (size_t)total, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { rh_set_error("rh_octree_cell_enabled: %s", hipGetErrorString(e)); return RH_E_NODEVICE; }
    *n_out = total;
    if (total > cap) { rh_set_error("rh_octree_cell_enabled: %lld enabled points in the cell, capacity %lld", (long long)total, (long long)cap); return RH_E_CAPACITY; }
    return RH_OK;
}
