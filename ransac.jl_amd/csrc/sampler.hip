// sampler.hip -- device-side minimal-set sampling + fitting (sampling_streams = 1).
// One thread per (iteration, minimal set): draws the set from its own splitmix64 stream
// (fit_shared.h), gathers the points, runs the plane / sphere / cylinder fits in the order of
// iteration.shape_types (forcefitshapes!, src/fitting.jl:165-173) and appends every fitted
// candidate, tagged with its slot = ((iteration, set), shape type), to a compact list.  The host
// sorts the list by slot, which restores the reference's candidate order exactly.
#include "fit_shared.h"
#include "rh_internal.h"

namespace {

struct DevEnabled {
    const uint64_t *w;
    const int32_t *prefix;   // exclusive popcount prefix per word
    int64_t nwords;
    int32_t total;
    __device__ bool test(int64_t i0) const { return (w[i0 >> 6] >> (i0 & 63)) & 1ULL; }
    __device__ int64_t select(int64_t r) const
    {
        if (r < 1 || r > total) return 0;
        int64_t lo = 0, hi = nwords;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (prefix[mid] < r) lo = mid; else hi = mid;
        }
        uint64_t m = w[lo];
        const int rem = (int)(r - prefix[lo]);
        for (int t = 1; t < rem; t++) m &= m - 1;
        return (lo << 6) + __ffsll((unsigned long long)m);
    }
};

constexpr int RH_MAX_DRAWN = 8;   // device path; larger minimal sets use the host sampler

template <int DN>
__global__ void __launch_bounds__(128)
sample_fit_kernel(const double *__restrict__ rec, int64_t n, DevEnabled en, int32_t n_enabled,
                  const rhfit::OctView oc, const double *__restrict__ Pwin, const rh_params prm, uint64_t seed,
                  int64_t k0, int32_t n_iters, rh_cand_entry *__restrict__ out,
                  int32_t cap, int32_t *__restrict__ out_count, unsigned long long *__restrict__ draws_per_iter,
                  int32_t *__restrict__ gave_up_flag)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n_iters * prm.minsubsetN;
    if (t >= total) return;
    const int32_t it = (int32_t)(t / prm.minsubsetN);
    const int32_t j = (int32_t)(t - (int64_t)it * prm.minsubsetN);
    uint64_t x = rhfit::set_stream_init(seed, (uint64_t)(k0 + it), (uint64_t)j);
    constexpr int CAP = DN > 0 ? DN : RH_MAX_DRAWN;
    int64_t sd[CAP];
    uint32_t nd = 0;
    bool gave_up = false;
    const int drawN = DN > 0 ? DN : prm.drawN;
    int level = 1;
    const bool ok = Pwin != nullptr
                        ? rhfit::sample_minimal_set_octree<DN>(en, oc, Pwin + (int64_t)it * oc.depth, n, (int64_t)n_enabled,
                                                               drawN, &x, sd, &nd, &gave_up, &level)
                        : rhfit::sample_minimal_set<DN>(en, n, (int64_t)n_enabled, drawN, &x, sd, &nd, &gave_up);
    // draws per iteration: one atomic per (wave, iteration) instead of one per set -- thousands of
    // same-address atomics serialise in L2 and dominated the kernel
    {
        uint64_t todo = __builtin_amdgcn_ballot_w64(true);
        const int lane = threadIdx.x & 63;
        while (todo != 0) {
            const int leader = __builtin_ctzll(todo);
            const int32_t it0 = __shfl(it, leader);
            const uint64_t grp = __builtin_amdgcn_ballot_w64(it == it0) & todo;
            unsigned v = (it == it0) ? nd : 0u;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == leader) atomicAdd(&draws_per_iter[it0], (unsigned long long)v);
            todo &= ~grp;
        }
    }
    if (gave_up) atomicExch(gave_up_flag, 1);
    if (!ok) return;
    double fp[3 * CAP], fn[3 * CAP];
#pragma unroll
    for (int q = 0; q < drawN; q++) {
        const int64_t i0 = sd[q] - 1;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2 *r = (const f64x2 *)(rec + 8 * i0);   // one 64-byte record per point
        const f64x2 a = r[0], b = r[1], c = r[2];
        fp[3 * q] = a.x; fp[3 * q + 1] = a.y; fp[3 * q + 2] = b.x;
        fn[3 * q] = b.y; fn[3 * q + 1] = c.x; fn[3 * q + 2] = c.y;
    }
    for (int ti = 0; ti < prm.n_shape_types; ti++) {
        rh_shape s;
        for (int q = 0; q < 10; q++) s.v[q] = 0.0;
        s.kind = -1;
        s.outwards = 0;
        bool fitted = false;
        switch (prm.shape_types[ti]) {
        case RH_PLANE: fitted = rhfit::fit_plane(fp, fn, drawN, prm, &s); break;
        case RH_SPHERE: fitted = rhfit::fit_sphere(fp, fn, drawN, prm, &s); break;
        case RH_CYLINDER: fitted = rhfit::fit_cylinder(fp, fn, drawN, prm, &s); break;
        case RH_CONE: fitted = rhfit::fit_cone(fp, fn, drawN, prm, &s); break;
        default: break;
        }
        if (!fitted) continue;
        const int32_t pos = atomicAdd(out_count, 1);
        if (pos < cap) {
            out[pos].slot = (int64_t)t * prm.n_shape_types + ti;
            out[pos].level = level;
            out[pos].pad = 0;
            out[pos].shape = s;
        }
    }
}

}  // namespace

int rhk_sample_fit(rh_cloud *c, const rh_params *prm, uint64_t seed, int64_t k0, int32_t n_iters, int32_t n_enabled,
                   const double *d_P, rh_cand_entry *d_out, int32_t cap, int32_t *d_count, unsigned long long *d_draws,
                   int32_t *d_gave_up)
{
    if (prm->drawN > RH_MAX_DRAWN) { rh_set_error("device sampler supports drawN <= %d", RH_MAX_DRAWN); return RH_E_INVALID; }
    if (!c->select_valid) RH_TRY(rhk_build_select(c));
    RH_HIP(hipMemsetAsync(d_count, 0, sizeof(int32_t), c->stream));
    RH_HIP(hipMemsetAsync(d_gave_up, 0, sizeof(int32_t), c->stream));
    RH_HIP(hipMemsetAsync(d_draws, 0, sizeof(unsigned long long) * (size_t)n_iters, c->stream));
    const int64_t total = (int64_t)n_iters * prm->minsubsetN;
    if (total == 0) return RH_OK;
    DevEnabled en;
    en.w = c->enabled;
    en.prefix = c->word_prefix;
    en.nwords = c->nwords;
    en.total = n_enabled;
    rhfit::OctView oc;
    oc.code = c->oct_code; oc.perm = c->oct_perm; oc.pos = c->oct_pos; oc.men = c->oct_men; oc.prefix = c->oct_prefix;
    oc.n = c->n; oc.nwords = c->nwords; oc.depth = c->oct_depth;
    if (prm->drawN == 3)   // the reference's default: fully unrolled, no scratch
        hipLaunchKernelGGL(sample_fit_kernel<3>, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, c->stream, c->rec,
                           c->n, en, n_enabled, oc, d_P, *prm, seed, k0, n_iters, d_out, cap, d_count, d_draws,
                           d_gave_up);
    else
        hipLaunchKernelGGL(sample_fit_kernel<0>, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, c->stream, c->rec,
                           c->n, en, n_enabled, oc, d_P, *prm, seed, k0, n_iters, d_out, cap, d_count, d_draws,
                           d_gave_up);
    RH_HIP(hipGetLastError());
    return RH_OK;
}
