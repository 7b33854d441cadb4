/* score_demo.c -- the C ABI of libransac_hip.so used from plain C (no Python, no torch).
 * Builds a small synthetic cloud (a noisy plane patch + uniform outliers), scores two plane
 * candidates in one batch (host buffers, then resident buffers with batches in flight), refits the better one, invalidates its points and runs the whole
 * ransac() loop.  Build (done by __graft_entry__.build()):
 *   gcc -O2 -Iinclude examples/score_demo.c -Lransac.jl_amd -lransac_hip -lm -Wl,-rpath,$PWD/ransac.jl_amd -o examples/score_demo
 * Needs an MI355X to run; prints "score_demo ok" on success, the rh_last_error() text otherwise. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ransac_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != RH_OK) {                                                      \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rh_last_error());      \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static unsigned long long lcg = 88172645463325252ULL;
static double urand(void)
{
    lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17;
    return (double)(lcg >> 11) * (1.0 / 9007199254740992.0);
}

int main(void)
{
    const int64_t n = 20000, n_plane = 12000;
    double *xyz = malloc(sizeof(double) * 3 * n), *nrm = malloc(sizeof(double) * 3 * n);
    int64_t *subset = malloc(sizeof(int64_t) * (n / 2));
    for (int64_t i = 0; i < n; i++) {
        if (i < n_plane) { /* z = 5 plane patch, tiny noise, normal +z */
            xyz[3 * i] = 100 * urand(); xyz[3 * i + 1] = 100 * urand(); xyz[3 * i + 2] = 5.0 + 0.02 * (urand() - 0.5);
            nrm[3 * i] = 0; nrm[3 * i + 1] = 0; nrm[3 * i + 2] = 1;
        } else {
            double a = 2 * 3.14159265358979 * urand(), z = 2 * urand() - 1, r = sqrt(1 - z * z);
            xyz[3 * i] = 100 * urand(); xyz[3 * i + 1] = 100 * urand(); xyz[3 * i + 2] = 100 * urand();
            nrm[3 * i] = r * cos(a); nrm[3 * i + 1] = r * sin(a); nrm[3 * i + 2] = z;
        }
    }
    for (int64_t j = 0; j < n / 2; j++) subset[j] = 2 * j + 1; /* 1-based, every other point */

    rh_params p;
    rh_default_params(&p);
    rh_cloud *c = NULL;
    CHECK(rh_cloud_create(xyz, nrm, n, subset, n / 2, 0, &c));

    rh_shape cand[2] = { { RH_PLANE, 0, { 50, 50, 5.0, 0, 0, 1 } }, { RH_PLANE, 0, { 50, 50, 60.0, 0, 0, 1 } } };
    int32_t counts[2];
    CHECK(rh_score_batch(c, cand, 2, &p, counts, NULL));
    printf("counts on subset 1: %d %d (of %lld)\n", counts[0], counts[1], (long long)(n / 2));
    if (counts[0] != n_plane / 2 || counts[1] > 50) { fprintf(stderr, "unexpected counts\n"); return 1; }

    /* the same batch with resident buffers, three batches in flight (rh_set_option "batches_in_flight"): call k writes count
     * buffer k mod 3; rh_cloud_sync (like every other call on the cloud) joins the streams */
    {
        void *d_cand = NULL, *d_cnt[3] = { NULL, NULL, NULL };
        CHECK(rh_dev_alloc(c, sizeof cand, &d_cand));
        CHECK(rh_dev_upload(c, d_cand, cand, sizeof cand));
        for (int k = 0; k < 3; k++) CHECK(rh_dev_alloc(c, sizeof counts, &d_cnt[k]));
        CHECK(rh_set_option(c, "batches_in_flight", 3));
        for (int k = 0; k < 6; k++) CHECK(rh_score_batch_dev(c, (const rh_shape *)d_cand, 2, &p, (int32_t *)d_cnt[k % 3], NULL));
        CHECK(rh_cloud_sync(c));
        for (int k = 0; k < 3; k++) {
            int32_t got[2];
            CHECK(rh_dev_download(c, got, d_cnt[k], sizeof got));
            if (got[0] != counts[0] || got[1] != counts[1]) { fprintf(stderr, "batches in flight: unexpected counts\n"); return 1; }
            CHECK(rh_dev_free(c, d_cnt[k]));
        }
        CHECK(rh_set_option(c, "batches_in_flight", RH_OPTION_UNSET));
        CHECK(rh_dev_free(c, d_cand));
    }

    int64_t *idx = malloc(sizeof(int64_t) * n), n_in = 0, left = 0;
    CHECK(rh_refit(c, &cand[0], &p, idx, n, &n_in));
    CHECK(rh_invalidate(c, idx, n_in));
    CHECK(rh_cloud_count_enabled(c, &left));
    printf("refit: %lld inliers, %lld points left enabled\n", (long long)n_in, (long long)left);
    /* a stray outlier can sit on the plane with a matching normal: allow a handful */
    if (n_in < n_plane || n_in > n_plane + 8 || left != n - n_in) { fprintf(stderr, "unexpected refit result\n"); return 1; }

    CHECK(rh_cloud_enable_all(c));
    p.n_shape_types = 1; p.shape_types[0] = RH_PLANE; p.tau = 500; p.itermax = 50; p.minsubsetN = 64;
    rh_rng rng;
    rh_rng_seed(&rng, 1234);
    rh_result res;
    CHECK(rh_ransac(c, xyz, nrm, &p, &rng, &res));
    printf("ransac: %lld shape(s) in %lld iteration(s); first has %lld points\n", (long long)res.n_shapes,
           (long long)res.iterations, res.n_shapes ? (long long)res.shapes[0].n_inpoints : 0LL);
    const int ok = res.n_shapes >= 1 && res.shapes[0].n_inpoints >= n_plane && res.shapes[0].n_inpoints <= n_plane + 8;
    rh_result_free(&res);
    CHECK(rh_cloud_destroy(c));
    free(xyz); free(nrm); free(subset); free(idx);
    if (!ok) { fprintf(stderr, "ransac did not extract the plane\n"); return 1; }
    printf("score_demo ok\n");
    return 0;
}
