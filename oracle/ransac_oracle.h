/*
 * ransac_oracle.h -- CPU oracle for the Efficient-RANSAC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a single-threaded, plain-C restatement of
 * the reference's algorithm (cserteGT3/RANSAC.jl v0.6.0, Julia) for the path
 * named in BASELINE.json: per-point compatibility tests, scorecandidate, refit,
 * score statistics, minimal-set fits, sampling and the ransac() loop, plus the
 * (dormant) parameter-space connected-component pass and the octree build.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product (libransac_hip.so) never links, loads or calls it.
 *
 * Every function cites the reference file:line (paths under /root/reference)
 * it follows.  Arithmetic is IEEE binary64 in the reference's operation order;
 * build with -ffp-contract=off (see Makefile).
 *
 * PARITY PINNING.  The reference is Julia and no Julia runtime exists in the
 * build container, so the oracle cannot be checked against the reference run
 * live.  It is pinned by the reference's own known-answer tests where they
 * touch the path (tests/golden/, see tests/test_oracle_golden.py):
 * test/octree.jl, test/dummyspheretest.jl, test/utilitytests.jl,
 * test/confidenceintervals.jl, test/fitting.jl, test/parameterspacebitmap.jl.
 * compatibles*, scorecandidate, refit, estimatescore, prob, samplepointcloud4!
 * and ransac() have NO reference test or stored output: for those, parity is
 * UNPINNED (restated from source text only).  Third-party arithmetic that is
 * not under /root/reference (StaticArrays 0.11-1.2 dot/cross/norm/normalize,
 * LinearAlgebra rank/\, Random) is restated from its published algorithm.
 */
#ifndef RANSAC_ORACLE_H
#define RANSAC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* shape kinds -- order of DEFAULT_PARAMETERS is plane, cone, cylinder, sphere
 * (src/RANSAC.jl:94) but the numeric tags here are ours. */
enum { ORC_PLANE = 0, ORC_SPHERE = 1, ORC_CYLINDER = 2, ORC_CONE = 3 };

/* POD candidate.  Same byte layout as rh_shape in include/ransac_hip.h.
 *   PLANE    (plane.jl:8-11)      v[0..2]=point  v[3..5]=normal
 *   SPHERE   (sphere.jl:9-13)     v[0..2]=center v[3]=radius            outwards
 *   CYLINDER (cylinder.jl:11-16)  v[0..2]=axis   v[3..5]=center v[6]=R  outwards
 *   CONE     (cone.jl:11-19)      v[0..2]=apex   v[3..5]=axis   v[6]=opang
 *                                 v[7]=cos(-opang/2) v[8]=sin(-opang/2) outwards
 * v[7], v[8] are host-computed (the reference evaluates cos/sin inside
 * rodrigues, utilities.jl:21-22). orc_shape_finalize fills them with the oracle's own fdlibm-algorithm
 * kernels (orc_trig.h; what Julia's Base.sin/cos port; <= 1 ulp from any libm). */
typedef struct {
    int32_t kind;
    int32_t outwards;
    double v[10];
} orc_shape;

enum { ORC_SCORE_INT64_WRAP = 0, ORC_SCORE_F64 = 1 };
enum { ORC_S_LENGTHC = 1, ORC_S_ALLCAND = 2, ORC_S_NOFMINSET = 3 };

/* Parameters (utilities.jl:332-399; plane.jl:22; sphere.jl:25; cylinder.jl:27;
 * cone.jl:30).  Per-kind arrays are indexed by ORC_* kind.  cos_alpha and
 * cos_parallelthr are host-computed thresholds. */
typedef struct {
    double eps[4];
    double alpha[4];
    double cos_alpha[4];       /* cos(alpha[k]) */
    double collin_threshold;   /* common.collin_threshold = 0.2 */
    double parallelthrdeg;     /* common.parallelthrdeg = 1.0 */
    double cos_parallelthr;    /* cosd(parallelthrdeg) */
    double sphere_par;         /* sphere.sphere_par = 0.02 */
    double minconeopang;       /* cone.minconeopang = deg2rad(2) */
    double prob_det;           /* iteration.prob_det = 0.9 */
    int64_t tau;               /* iteration.tau = 900 */
    int64_t itermax;           /* iteration.itermax = 1000 */
    int32_t drawN;             /* iteration.drawN = 3 */
    int32_t minsubsetN;        /* iteration.minsubsetN = 15 */
    int32_t extract_s;         /* :nofminset */
    int32_t terminate_s;       /* :nofminset */
    int32_t n_shape_types;
    int32_t shape_types[8];    /* ORC_* kinds in iteration.shape_types order */
    int32_t score_mode;        /* ORC_SCORE_INT64_WRAP (faithful) or ORC_SCORE_F64 */
    int32_t sphere_uses_enabled; /* 0 = faithful Q4 (sphere.jl:121,131), 1 = fixed */
    int32_t sampling_streams;  /* 0 = one sequential stream (reference structure); 1 = one stream per
                                  (iteration, minimal set): pure function of (seed, k, j) */
    int32_t octree_sampling;   /* 0 = root cell always (the reference's live behaviour);
                                  1 = level-weighted sampling on a linear octree (docs/src/ransac.md:73-96) */
    int32_t octree_max_depth;  /* depth cap of the linear octree (default 10) */
} orc_params;

void orc_default_params(orc_params *p);
void orc_params_finalize(orc_params *p); /* recompute cos_alpha, cos_parallelthr */
void orc_shape_finalize(orc_shape *s);   /* fill cone cos/sin fields */

/* ---- per-point compatibility (SoA or AoS agnostic: single point) ---- */
int orc_compatible(const orc_shape *s, const double p[3], const double n[3],
                   double eps, double cos_alpha);

/* the two compared quantities of one test (distance side vs eps, angle side vs cos alpha) */
void orc_compat_values(const orc_shape *s, const double p[3], const double n[3], double out[2]);
int orc_variant(void); /* the rounding-order variant this build restates (0 = default; ransac_oracle.c header) */

/* ---- confidence interval / score statistics (confidenceintervals.jl) ---- */
typedef struct { double min, max, E; } orc_ci;
int orc_confidence_interval(double x, double y, orc_ci *out); /* -1 = "out of order" */
orc_ci orc_notsoconfident(double x, double y);
int orc_isoverlap(orc_ci a, orc_ci b);
orc_ci orc_estimatescore(int64_t S1length, int64_t Plength, int64_t sigma, int score_mode);
void orc_trace_set(double *buf, int64_t cap_records);   /* decision trace of orc_ransac: (iteration, E(best), s, ppp) per evaluated extraction test */
int64_t orc_trace_count(void);
double orc_prob(double n, int64_t s, int64_t N, int64_t k);

/* ---- cloud ---- */
typedef struct orc_cloud orc_cloud;
/* xyz/nrm: AoS n x 3 doubles; subset1: 1-based indices (length s) */
orc_cloud *orc_cloud_create(const double *xyz, const double *nrm, int64_t n,
                            const int64_t *subset1_1based, int64_t s);
void orc_cloud_destroy(orc_cloud *c);
void orc_cloud_set_enabled(orc_cloud *c, const uint64_t *chunks, int64_t nchunks);
void orc_cloud_get_enabled(const orc_cloud *c, uint64_t *chunks, int64_t nchunks);
void orc_cloud_enable_all(orc_cloud *c);
int64_t orc_cloud_count_enabled(const orc_cloud *c);

/* scorecandidate (plane.jl:61-71 etc.): returns count; if inpoints!=NULL writes
 * 1-based original indices in subset order (capacity >= s); if mask!=NULL writes
 * ceil(s/64) words, bit j = subset position j compatible(&enabled). */
int64_t orc_scorecandidate(const orc_cloud *c, const orc_shape *s, const orc_params *p,
                           int64_t *inpoints, uint64_t *mask);
/* batched: counts[b] */
void orc_score_batch(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                     int32_t *counts, uint64_t *masks /* b x ceil(s/64) or NULL */);
void orc_score_batch_mt(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                        int32_t *counts, int32_t nthreads);   /* OpenMP over candidates (bench steelman) */
void orc_score_masks_mt(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                        int32_t *counts, uint64_t *masks, int32_t nthreads);
void orc_margin_census_mt(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                          const double *edges, int nedges, int64_t *hist, int32_t nthreads);
/* refit (plane.jl:137-143 etc.): ascending 1-based indices; returns count */
int64_t orc_refit(const orc_cloud *c, const orc_shape *s, const orc_params *p,
                  int64_t *idx_out, int64_t cap);
void orc_invalidate(orc_cloud *c, const int64_t *idx_1based, int64_t n);
/* least-squares refit (our specification; the reference has none): same selection, model and
 * Gauss-Newton steps as rh_refit_lsq, sums taken sequentially in index order */
int orc_refit_lsq(const orc_cloud *c, const orc_shape *s, const orc_params *p, int max_iter, orc_shape *out,
                  int64_t *n_used, double *rms, int *iters_done);
/* k-th (1-based) enabled point in ascending index order, 1-based; 0 if none */
int64_t orc_select_enabled(const orc_cloud *c, int64_t k);

/* ---- Float32 clouds (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109): orc_f32.c, the binary32 twin of the
 *      per-point part.  Shapes are orc_shape whose fields hold binary32 numbers (orc32_shape_finalize rounds them and
 *      fills the cone's cos / sin as binary32); eps / cos_alpha stay double and are compared after exact promotion. ---- */
typedef struct orc_cloud32 orc_cloud32;
void orc32_shape_finalize(orc_shape *s);
int orc32_compatible(const orc_shape *s, const float p[3], const float n[3], double eps, double cos_alpha);
orc_cloud32 *orc32_cloud_create(const float *xyz, const float *nrm, int64_t n, const int64_t *subset1_1based, int64_t s);
void orc32_cloud_destroy(orc_cloud32 *c);
void orc32_cloud_enable_all(orc_cloud32 *c);
void orc32_cloud_set_enabled(orc_cloud32 *c, const uint64_t *chunks, int64_t nchunks);
void orc32_cloud_get_enabled(const orc_cloud32 *c, uint64_t *chunks, int64_t nchunks);
int64_t orc32_scorecandidate(const orc_cloud32 *c, const orc_shape *s, const orc_params *p, int64_t *inpoints, uint64_t *mask);
void orc32_score_masks_mt(const orc_cloud32 *c, const orc_shape *s, int32_t b, const orc_params *p, int32_t *counts,
                          uint64_t *masks /* or NULL */, int32_t nthreads);
int64_t orc32_refit(const orc_cloud32 *c, const orc_shape *s, const orc_params *p, int64_t *idx_out, int64_t cap);
/* fit on Float32 points (p, n: the Float32 values as doubles); ORC_CONE never fits (not restated in binary32) */
int orc32_fit(int kind, const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out);
/* A cloud of Float32 values (given as doubles) whose per-point tests and fits run in binary32: orc_scorecandidate,
 * orc_refit and orc_ransac then take the orc32 paths -- the whole loop on a Float32 cloud (RANSACCloud(...;
 * force_eltype = Float32), octree.jl:102-109).  All four kinds are fitted (the cone: orc_f32.c, round 5). */
void orc_cloud_set_f32(orc_cloud *c, int f32);
void orc32_invalidate(orc_cloud32 *c, const int64_t *idx_1based, int64_t n);

/* ---- minimal-set fits: p,n are lp x 3 AoS; return 1 = fitted, 0 = nothing ---- */
int orc_fit(int kind, const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out);
int orc_fit2pointsphere(const double *v, const double *n, const orc_params *prm, orc_shape *out);
int orc_fit2pointcylinder(const double *p, const double *n, const orc_params *prm, orc_shape *out);
int orc_fit3pointcone(const double *p, const double *n, orc_shape *out);

/* ---- small math helpers exposed for golden tests ---- */
void orc_rodriguesrad(const double nv[3], double theta, double R[9] /* row-major */);
void orc_pluscrossprod(double A[9] /* row-major */, double value, const double v[3]);
int orc_rank(const double *A, int m, int n); /* row-major, m,n <= 4 */
void orc_findAABB(const double *pts, int64_t n, int dim, double *minv, double *maxv);
int orc_iswithinrectangle(const double origin[3], const double widths[3], const double p[3]);

/* ---- octree (octree.jl:158-244, RegionTrees ^0.3 semantics) ---- */
typedef struct orc_octree orc_octree;
orc_octree *orc_octree_build(const double *xyz, int64_t n);
orc_octree *orc_octree_build_f32(const float *xyz, int64_t n);   /* the tree of a Float32 cloud: binary32 geometry */
void orc_octree_destroy(orc_octree *t);
int orc_octree_depth(const orc_octree *t);
/* findleaf(root, p): returns leaf depth; path[d-1] = node id at depth d (root = path[0]) */
int orc_octree_findleaf(const orc_octree *t, const double p[3], int32_t *path, int cap);
int64_t orc_octree_node_npoints(const orc_octree *t, int32_t node);
const int64_t *orc_octree_node_points(const orc_octree *t, int32_t node); /* 1-based */

/* ---- RNG: xoshiro256++ seeded through splitmix64, or an injected u64 stream ---- */
typedef struct {
    uint64_t s[4];
    const uint64_t *stream; /* optional injected raw draws, consumed first */
    int64_t stream_len, stream_pos;
    int64_t draws;
} orc_rng;
void orc_rng_seed(orc_rng *r, uint64_t seed);
uint64_t orc_rng_next(orc_rng *r);
int64_t orc_rng_range(orc_rng *r, int64_t n); /* rand(1:n) = 1 + mulhi(next, n) */

/* ---- driver (iterations.jl:35-162) ---- */
typedef struct {
    orc_shape shape;
    int64_t n_inpoints;
    int64_t *inpoints; /* ascending 1-based, malloc'd */
    double score_E;
    int64_t iteration;
} orc_extracted;

typedef struct {
    orc_extracted *shapes;
    int64_t n_shapes;
    int64_t iterations;        /* iterations executed */
    int64_t candidates_scored; /* countcandidates[2] */
    int64_t scored_left;       /* length(scoredshapes) at exit */
    double seconds;
} orc_result;

int orc_ransac(orc_cloud *c, const double *xyz, const double *nrm, const orc_params *p,
               orc_rng *rng, int octree_depth, orc_result *out);
void orc_result_free(orc_result *r);

/* ---- parameter-space bitmap + largest connected component
 *      (parameterspacebitmap.jl:12-60, 69-109) ---- */
/* bitmap: xs*ys bytes column-major (Julia BitMatrix [x,y] -> bitmap[x + xs*y], 0-based);
 * conn8: 0 = 4-connectivity (1:ndims), 1 = trues(3,3).
 * Writes 0-based linear indices (column-major ascending) of the largest component
 * into out (cap entries); returns its size (0 when the bitmap is empty). */
int64_t orc_largestconncomp(const uint8_t *bitmap, int32_t xs, int32_t ys, int conn8,
                            int64_t *out, int64_t cap);
/* bitmapparameters: params2d n x 2 AoS; returns 0 ok; writes xs,ys,betax,betay,
 * bitmap (caller allocs after a sizing call with bitmap==NULL) and idxmap (int64, 0 = empty). */
int orc_bitmapparameters(const double *params2d, const uint8_t *compat, const int64_t *idsource,
                         int64_t n, double beta, int32_t *xs, int32_t *ys,
                         double *betax, double *betay, uint8_t *bitmap, int64_t *idxmap);

#ifdef __cplusplus
}
#endif
#endif
