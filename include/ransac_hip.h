/*
 * ransac_hip.h -- C ABI of libransac_hip.so: the MI355X-native (gfx950, HIP)
 * replacement for the data-parallel hot path of cserteGT3/RANSAC.jl v0.6.0.
 *
 * The reference has no FFI; its boundary is Julia multiple dispatch on
 * FittedShape subtypes (src/fitting.jl:8-66).  Each entry point below names the
 * reference function it replaces (paths under /root/reference).  A Julia
 * `ccall` shim (julia/RANSACHIP.jl) or any FFI binds these.
 *
 * Conventions
 *  - every call returns int: 0 = ok, negative = error (RH_E_*); the message is
 *    available from rh_last_error() on the calling thread;
 *  - the caller allocates every output; the library never keeps a host pointer
 *    past the call and never calls back into the host runtime (GC-safe);
 *  - point indices cross the boundary 1-based int64, like `inpoints::Vector{Int}`
 *    (src/fitting.jl:81-84);
 *  - enabled bits use BitVector's layout: uint64 chunks, bit i%64 of chunk i/64,
 *    LSB first (src/octree.jl:42);
 *  - one host thread per rh_cloud at a time; calls are synchronous unless the
 *    name ends in _dev (those enqueue on the cloud's HIP stream);
 *  - all arithmetic is IEEE binary64 in the reference's operation order
 *    (no FMA contraction), so inlier sets are bit-identical to the CPU path.
 *  - there is NO CPU fallback: without a usable HIP device every cloud call
 *    fails with RH_E_NODEVICE.
 */
#ifndef RANSAC_HIP_H
#define RANSAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_VERSION 109

enum {
    RH_OK = 0,
    RH_E_INVALID = -1,   /* bad argument (the reference would hit an @assert) */
    RH_E_NODEVICE = -2,  /* no HIP device / HIP runtime error */
    RH_E_NOMEM = -3,
    RH_E_CAPACITY = -4,  /* caller-provided output too small; *n_out holds the needed size */
    RH_E_INTERNAL = -5,
};

/* shape kinds */
enum { RH_PLANE = 0, RH_SPHERE = 1, RH_CYLINDER = 2, RH_CONE = 3 };

/* POD candidate = the reference's FittedShape structs flattened:
 *   RH_PLANE    FittedPlane    (src/shapes/plane.jl:8-11)     v[0..2]=point  v[3..5]=normal
 *   RH_SPHERE   FittedSphere   (src/shapes/sphere.jl:9-13)    v[0..2]=center v[3]=radius
 *   RH_CYLINDER FittedCylinder (src/shapes/cylinder.jl:11-16) v[0..2]=axis   v[3..5]=center v[6]=radius
 *   RH_CONE     FittedCone     (src/shapes/cone.jl:11-19)     v[0..2]=apex   v[3..5]=axis   v[6]=opang
 *                              v[7]=cos(-opang/2), v[8]=sin(-opang/2): computed by the HOST
 *                              (the reference evaluates them in rodrigues, src/utilities.jl:21-22);
 *                              rh_shape_finalize fills them with fdlibm-algorithm kernels (det_math.h: the
 *                              same bits on host and device, <= 1 ulp from any libm).
 * `outwards` is the Bool field of sphere/cylinder/cone; ignored for planes. */
typedef struct {
    int32_t kind;
    int32_t outwards;
    double v[10];
} rh_shape;

enum { RH_SCORE_INT64_WRAP = 0, RH_SCORE_F64 = 1 };
enum { RH_S_LENGTHC = 1, RH_S_ALLCAND = 2, RH_S_NOFMINSET = 3 };

/* Parameters = the reference's nested NamedTuple (src/utilities.jl:332-399 and the
 * defaultshapeparameters of each shape file), flattened.  Per-kind arrays are
 * indexed by RH_* kind.  cos_alpha[] / cos_parallelthr are thresholds computed by
 * the host (rh_params_finalize uses the C libm). */
typedef struct {
    double eps[4];             /* <shape>.eps  */
    double alpha[4];           /* <shape>.alpha (rad) */
    double cos_alpha[4];       /* cos(alpha): isparallel, src/utilities.jl:115-117 */
    double collin_threshold;   /* common.collin_threshold */
    double parallelthrdeg;     /* common.parallelthrdeg */
    double cos_parallelthr;    /* cosd(parallelthrdeg) */
    double sphere_par;         /* sphere.sphere_par */
    double minconeopang;       /* cone.minconeopang */
    double prob_det;           /* iteration.prob_det */
    int64_t tau;               /* iteration.tau */
    int64_t itermax;           /* iteration.itermax */
    int32_t drawN;             /* iteration.drawN */
    int32_t minsubsetN;        /* iteration.minsubsetN */
    int32_t extract_s;         /* iteration.extract_s  (RH_S_*) */
    int32_t terminate_s;       /* iteration.terminate_s */
    int32_t n_shape_types;
    int32_t shape_types[8];    /* iteration.shape_types as RH_* kinds, in order */
    int32_t score_mode;        /* RH_SCORE_INT64_WRAP = the reference's wrapping Int64 product
                                  (src/confidenceintervals.jl:54,72); RH_SCORE_F64 = fixed */
    int32_t sphere_uses_enabled; /* 0 = reference behaviour (src/shapes/sphere.jl:121,131) */
    int32_t sampling_streams;  /* 0 = one sequential random stream (the reference's structure, host-side
                                  sampling); 1 = one stream per (iteration, minimal set), a pure function
                                  of (seed, k, j): sampling + plane/sphere/cylinder fits run on the device */
    int32_t octree_sampling;   /* 0 = every minimal set from the root cell: the reference's live behaviour
                                  (constructor bug); 1 = what docs/src/ransac.md:73-96 describes:
                                  a level is drawn from the level distribution and the other points come from
                                  the first point's cell at that level (linear Morton octree; needs
                                  sampling_streams = 1) */
    int32_t octree_max_depth;  /* depth cap of the linear octree (default 10) */
} rh_params;

typedef struct rh_cloud rh_cloud;

/* ---- library ---- */
int rh_version(void);
const char *rh_last_error(void);
int rh_device_count(int *n_out);

/* ---- parameters (src/utilities.jl:332-399; RANSAC.jl:94) ---- */
void rh_default_params(rh_params *p);
void rh_params_finalize(rh_params *p);
void rh_shape_finalize(rh_shape *s);

/* ---- cloud: replaces RANSACCloud (src/octree.jl:37-59, ctors :78-138) ----
 * xyz_aos / nrm_aos: n x 3 doubles, i.e. Julia's Vector{SVector{3,Float64}} memory
 * as is; subset1_idx_1based: pc.subsets[1] (the only subset the reference scores,
 * src/iterations.jl:95).  Transposes to SoA in HBM, stores subset 1 contiguously in
 * subset order; all points start enabled (src/octree.jl:84). */
int rh_cloud_create(const double *xyz_aos, const double *nrm_aos, int64_t n,
                    const int64_t *subset1_idx_1based, int64_t s, int device, rh_cloud **out);
/* RANSACCloud(...; force_eltype = Float32) (src/octree.jl:102-109): xyz_aos / nrm_aos are Julia's
 * Vector{SVector{3,Float32}} memory as is.  On such a cloud rh_score_batch(_dev), rh_refit, rh_invalidate,
 * rh_select_enabled and the enabled-bit calls work and every per-point operation is a binary32 operation (the shapes'
 * fields are rounded to binary32 on entry; rh_shape_finalize_f32 prepares a Float32 shape: fields rounded, the cone's
 * cos / sin as binary32); eps and cos_alpha stay doubles and are compared after exact promotion, like Julia compares a
 * Float32 with a Float64.  rh_ransac runs on such a cloud too (rh_ransac_f32), all four kinds; rh_refit_lsq is Float64-only. */
int rh_cloud_create_f32(const float *xyz_aos, const float *nrm_aos, int64_t n,
                        const int64_t *subset1_idx_1based, int64_t s, int device, rh_cloud **out);
void rh_shape_finalize_f32(rh_shape *s);
int rh_cloud_destroy(rh_cloud *c);
int rh_cloud_info(const rh_cloud *c, int64_t *n, int64_t *s, int *device);
/* pc.isenabled (BitVector.chunks) in / out; nchunks must be ceil(n/64) */
int rh_cloud_set_enabled(rh_cloud *c, const uint64_t *chunks, int64_t nchunks);
int rh_cloud_get_enabled(rh_cloud *c, uint64_t *chunks, int64_t nchunks);
int rh_cloud_enable_all(rh_cloud *c);                   /* ransac(pc, params, true): src/iterations.jl:14-21 */
int rh_cloud_count_enabled(rh_cloud *c, int64_t *out);  /* count(pc.isenabled): src/iterations.jl:75 */

/* ---- hot path ---- */

/* Replaces scorecandidates! (src/fitting.jl:181-190) = B x scorecandidate
 * (plane.jl:61-71, sphere.jl:118-134, cylinder.jl:172-183, cone.jl:155-167) with one
 * batched launch for all candidates of all kinds.  counts_out[b] = number of compatible (and, except
 * for spheres in reference mode, enabled) points of subset 1.  masks_out (optional):
 * b rows of ceil(s/64) words; bit j of row i = subset position j is an inpoint of
 * candidate i, so inpoints = subsets[1][mask] in subset order. */
int rh_score_batch(rh_cloud *c, const rh_shape *shapes, int32_t b, const rh_params *p,
                   int32_t *counts_out, uint64_t *masks_out_or_null);

/* Same, all buffers resident in HBM on the cloud's device, enqueued on the cloud's
 * stream without synchronising (bench / multi-GPU plumbing). d_counts must hold b
 * int32 (overwritten). */
int rh_score_batch_dev(rh_cloud *c, const rh_shape *d_shapes, int32_t b, const rh_params *p,
                       int32_t *d_counts, uint64_t *d_masks_or_null);

/* Replaces refit (plane.jl:137-143, sphere.jl:179-190, cylinder.jl:228-234,
 * cone.jl:176-182): all enabled compatible points of the WHOLE cloud, ascending
 * 1-based.  If more than cap are found returns RH_E_CAPACITY with *n_out = needed. */
int rh_refit(rh_cloud *c, const rh_shape *shape, const rh_params *p,
             int64_t *idx_out_1based, int64_t cap, int64_t *n_out);

/* Least-squares refit -- the step of the paper the reference leaves out (docs/src/ransac.md:163-168;
 * its `refit` returns the shape unchanged).  NOT part of parity runs.  Selects the enabled points
 * compatible with `shape` at 3*eps, then fits: plane = total least squares; sphere / cylinder / cone
 * = Gauss-Newton on the geometric distance (normal equations accumulated on the device with f64
 * MFMA, solved on the host).  rms = root mean square distance of the selected points (for the
 * iterative kinds: before the last step). */
int rh_refit_lsq(rh_cloud *c, const rh_shape *shape, const rh_params *p, int32_t max_iter, rh_shape *out,
                 int64_t *n_used, double *rms, int32_t *iters_done);

/* Replaces invalidate_indexes! (src/fitting.jl:197-202). */
int rh_invalidate(rh_cloud *c, const int64_t *idx_1based, int64_t n);

/* Replaces the enabled-cell gather of samplepointcloud4! (src/fitting.jl:405-407,
 * 415-422) for the root cell: idx_out[i] = the ranks[i]-th enabled point (1-based
 * rank, ascending index order), 0 if the rank is out of range. */
int rh_select_enabled(rh_cloud *c, const int64_t *ranks_1based, int32_t k, int64_t *idx_out_1based);

/* ---- host-side pieces of the plugin API (O(1) per minimal set) ---- */

/* fit(::Type{T}, p, n, pc, params) (plane.jl:33-57, sphere.jl:87-114,
 * cylinder.jl:135-168, cone.jl:123-128).  p, n: lp x 3 AoS.  *fitted = 0 is the
 * reference's `nothing`. */
int rh_fit(int kind, const double *p, const double *n, int32_t lp, const rh_params *prm,
           rh_shape *out, int32_t *fitted);
/* The same on the points of a Float32 cloud (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109: p[i], n[i] are
 * SVector{3,Float32}, so plane.jl:33-57 / sphere.jl:29-114 / cylinder.jl:34-168 run in Float32 and return Float32 shapes;
 * the parameters stay what the caller made them, utilities.jl:488-503).  p, n carry the Float32 values as doubles.
 * RH_CONE (round 5): cone.jl:39-61 + 87-128 in binary32 -- rank() and \ of a Matrix{Float32} (LAPACK's single-precision
 * SVD and LU upstream) are a one-sided Jacobi SVD and an LU with partial pivoting in float here, acos / cos / sin the
 * deterministic double kernels rounded once: like the Float64 cone fit a restatement the reference cannot pin (it holds no
 * cone fixture); held bit for bit against the oracle's own binary32 twin. */
int rh_fit_f32(int kind, const double *p, const double *n, int32_t lp, const rh_params *prm,
               rh_shape *out, int32_t *fitted);

/* forcefitshapes! (src/fitting.jl:165-173) for the k minimal sets of an iteration in one call (what rh_sample_sets drew):
 * fit(T, ...) for every type of p->shape_types, in that order, on every set with ok[j] != 0 (ok = NULL: every set); the shapes
 * that fit are appended to shapes_out in (set, type) order -- the reference's candidate order -- with set_out[i] = the set a
 * shape came from.  xyz_aos / nrm_aos: the cloud's host arrays (n x 3 doubles; for f32 != 0 they hold a Float32 cloud's values
 * and the fits run in binary32, rh_fit_f32).  Host-side, O(1) per set.  RH_E_CAPACITY with *n_out = the needed size. */
int rh_fit_sets(const double *xyz_aos, const double *nrm_aos, const int64_t *idx_1based, const int32_t *ok_or_null, int32_t k,
                int32_t drawN, const rh_params *p, int32_t f32, rh_shape *shapes_out, int32_t *set_out_or_null, int32_t cap,
                int32_t *n_out);

/* estimatescore / ConfidenceInterval (src/confidenceintervals.jl:71-74, 53-59, 1-6) */
int rh_estimatescore(int64_t S1length, int64_t Plength, int64_t sigma, int32_t score_mode,
                     double *ci_min, double *ci_max, double *ci_E);
/* prob(n, s, N, k) (src/utilities.jl:262) */
double rh_prob(double n, int64_t s, int64_t N, int64_t k);

/* ---- driver: replaces ransac(pc, params) (src/iterations.jl:35-162) ---- */
typedef struct {
    uint64_t s[4];           /* xoshiro256++ state (rh_rng_seed) */
    const uint64_t *stream;  /* optional injected raw 64-bit draws, consumed first */
    int64_t stream_len, stream_pos;
    int64_t draws;
} rh_rng;
void rh_rng_seed(rh_rng *r, uint64_t seed);
/* rand(1:n) = 1 + floor(next * n / 2^64) */
int64_t rh_rng_range(rh_rng *r, int64_t n);

/* samplepointcloud4!(pc, ..) (src/fitting.jl:383-430) k times in a row, as ONE launch: what the reference's loop does once
 * per minimal set (iterations.jl:80-99) -- first point by rejection on rand(1:n) (:388-395), the other drawN - 1 as the
 * rand(1:count)-th enabled point of the root cell with one redraw when it repeats the first (:414-423), reject when two
 * coincide (:425-428).  The draws are the caller's generator's (an injected stream first), consumed exactly as k sequential
 * calls would consume them -- same number, same order -- so a loop that samples a whole iteration's sets through this call
 * takes the same decisions as one that calls rh_rng_range / rh_select_enabled per point (the device evaluates the call
 * for every start position of a window of draws, the host follows the chain).  idx_out: k x drawN points (1-based);
 * ok_out[j]: the reference's first return value; level_out (optional): its second (1: every set comes from the root cell,
 * SURVEY.md 0.5; 0 with ok = 0).  RH_E_INVALID without an enabled point (the reference would draw for ever). */
int rh_sample_sets(rh_cloud *c, int32_t drawN, rh_rng *rng, int32_t k, int64_t *idx_out_1based, int32_t *ok_out,
                   int32_t *level_out_or_null);

typedef struct {
    rh_shape shape;
    int64_t n_inpoints;
    int64_t *inpoints;       /* ascending, 1-based; points into the result's arena (rh_result_free) */
    double score_E;
    int64_t iteration;
} rh_extracted;              /* ExtractedShape, src/fitting.jl:81-84 */

typedef struct {
    rh_extracted *shapes;
    int64_t n_shapes;
    int64_t iterations;
    int64_t candidates_scored;
    int64_t scored_left;
    double seconds;          /* wall time of the loop (src/iterations.jl:46,159; not truncated) */
    double seconds_score;    /* device time in score launches + count read-back */
    double seconds_extract;  /* refit + invalidate + candidate liveness */
    double seconds_host;     /* sampling + fit + bookkeeping */
    double seconds_to_last_extraction;   /* wall time from the start of the loop to the end of the last extraction (0: none) */
    void *arena;             /* internal: the pinned host block that holds every inpoints list */
} rh_result;

/* xyz_aos / nrm_aos: the same host arrays given to rh_cloud_create (read for the
 * minimal-set fits only).  The cloud's enabled bits are updated in place, like
 * pc.isenabled. */
int rh_ransac(rh_cloud *c, const double *xyz_aos, const double *nrm_aos, const rh_params *p,
              rh_rng *rng, rh_result *out);
/* ransac(pc, params) on a Float32 cloud (rh_cloud_create_f32; octree.jl:102-109): the minimal-set fits (plane.jl:33-57,
 * sphere.jl:29-114, cylinder.jl:34-168), scoring, candidate liveness and refit all run in binary32, the thresholds stay
 * what the caller made them (utilities.jl:488-503); extracted shapes hold binary32 numbers (cones included: rh_fit_f32).
 * rh_ransac itself accepts such a cloud when handed
 * the Float32 values as doubles; this entry takes Julia's Vector{SVector{3,Float32}} memory as is. */
int rh_ransac_f32(rh_cloud *c, const float *xyz_aos, const float *nrm_aos, const rh_params *p,
                  rh_rng *rng, rh_result *out);
void rh_result_free(rh_result *r);

/* ---- one scene on several GPUs of a node (no counterpart in the reference, which is single-threaded:
 *      src/iterations.jl:35-162 run by `world` processes, one per GPU) ----
 * rh_mp_open is collective: every rank calls it with the same name (a POSIX shared-memory name, "/..."), rank 0
 * creates the segment.  slot_bytes bounds one rank's candidate list of one window (<= 0: 1 MiB).  rh_ransac_mp:
 * every rank holds a replica of the cloud in the same state and passes the same parameters and seed; the minimal sets
 * of every iteration are dealt round-robin to the ranks, the ranks exchange their windows' candidate lists through
 * the segment, and every rank returns exactly what rh_ransac returns for the same inputs.  Float64 clouds only
 * (RH_E_INVALID on a Float32 cloud: that combination has never been held against the single-GPU run). */
typedef struct rh_mp rh_mp;
int rh_mp_open(const char *shm_name, int32_t rank, int32_t world, int64_t slot_bytes, rh_mp **out);
int rh_mp_close(rh_mp *m);
/* the exchange on its own: every rank contributes `bytes` bytes (the same number everywhere), out gets world x bytes in
 * rank order; host memory only */
int rh_mp_allgather(rh_mp *m, const void *payload, int64_t bytes, void *out);
int rh_ransac_mp(rh_cloud *c, const double *xyz_aos, const double *nrm_aos, const rh_params *p,
                 rh_rng *rng, rh_mp *mp, rh_result *out);

/* ---- parameter-space bitmap + largest connected component
 *      (src/parameterspacebitmap.jl:12-60, 69-109; dead code upstream) ----
 * bitmap: xs*ys bytes, column-major like a Julia BitMatrix (pixel [x,y] at x + xs*y,
 * 0-based).  conn8 = 0: 4-connectivity (`1:ndims`), 1: `trues(3,3)`.  Writes the
 * 0-based linear indices (ascending) of the largest component. */
int rh_largestconncomp(const uint8_t *bitmap, int32_t xs, int32_t ys, int32_t conn8, int device,
                       int64_t *out, int64_t cap, int64_t *n_out);
int rh_bitmapparameters(const double *params2d, const uint8_t *compat, const int64_t *idsource_or_null,
                        int64_t n, double beta, int32_t *xs, int32_t *ys, double *betax, double *betay,
                        uint8_t *bitmap_or_null, int64_t *idxmap_or_null);

/* ---- measurement plumbing (bench.py): HIP events on the cloud's stream ---- */
/* rh_score_batch_dev twice, bracketed by HIP events on the cloud's stream: first the way
 * rh_score_batch_dev runs it (one kernel for all kinds) -> ms_out[4]; then one launch per kind
 * with an event before each and after the last -> ms_out[0..3] (0 for kinds without candidates
 * ~ an empty launch).  Waits for the batch.  ms_out[0] < 0 on entry: the first form only
 * (profiling passes that must see the product's launches alone). */
/* (a launch that takes super-tile lists, "st_cull": ms_out[4] is the score launch alone; the list launch in front of it is
 * returned by rh_last_list_launch_ms after the call, 0 when the launch took none) */
int rh_last_list_launch_ms(rh_cloud *c, float *ms_out);
int rh_score_batch_dev_timed(rh_cloud *c, const rh_shape *d_shapes, int32_t b, const rh_params *p,
                             int32_t *d_counts, uint64_t *d_masks_or_null, float *ms_out /* [5] */);
/* device time of the most recent rh_refit on this cloud: the full-cloud scan kernel and the
 * compaction (popcount + scan + expansion), from HIP events on the cloud's stream */
int rh_last_refit_ms(rh_cloud *c, float *ms_scan_out, float *ms_compact_out);
/* Stream the cloud's kernels and copies are enqueued on.  By default a cloud owns a non-blocking
 * stream.  use_external = 1: use the caller's hipStream_t (0 = the null stream) from now on, so the
 * caller can order its own work (a fill before, an RCCL collective after) against a
 * rh_score_batch_dev WITHOUT host synchronisation; use_external = 0: back to the own stream.
 * Waits for the work already enqueued on the previous stream.  The caller keeps the stream alive. */
int rh_cloud_set_stream(rh_cloud *c, void *hip_stream, int use_external);
int rh_timer_start(rh_cloud *c);
int rh_timer_stop(rh_cloud *c, float *ms_out); /* synchronises the stream */
int rh_cloud_sync(rh_cloud *c);
/* device allocations on the cloud's device for the *_dev entry points */
int rh_dev_alloc(rh_cloud *c, int64_t bytes, void **d_out);
int rh_dev_free(rh_cloud *c, void *d);
int rh_dev_upload(rh_cloud *c, void *d_dst, const void *h_src, int64_t bytes);
int rh_dev_download(rh_cloud *c, void *h_dst, const void *d_src, int64_t bytes);

/* ---- multi-GPU: candidate-sharded scoring with the score all-reduce inside the library ----
 * Candidates are independent (src/fitting.jl:181-190).  One process per GPU, each with a replica of the cloud; every
 * rank scores a slice of the batch, ONE RCCL all-reduce (sum, int32[b_total]) over xGMI gives every rank every count --
 * integer sums: bit-identical to one GPU.  librccl is loaded on first use (a copy already in the process is shared).
 *   rank 0:      rh_comm_unique_id(id)          ... the host program hands the 128 bytes to the other ranks ...
 *   every rank:  rh_comm_create(cloud, rank, world, id, &comm)
 *   per batch:   rh_score_batch_allreduce_dev(cloud, comm, d_my_shapes, b, offset, b_total, &params, d_counts_total)
 *                (enqueues: zero, score into [offset, offset + b), all-reduce on the communicator's own stream -- so the
 *                next batch's scoring overlaps it when the caller alternates two count buffers: a buffer may come back two
 *                calls later, the library orders that call behind the collective that read it)
 *   then:        rh_comm_fence(comm, cloud)  (the cloud's stream waits, no host sync)  or  rh_comm_sync(comm)  (host waits) */
#define RH_COMM_ID_BYTES 128
typedef struct rh_comm rh_comm;
int rh_comm_unique_id(void *id_out /* RH_COMM_ID_BYTES */);
int rh_comm_create(rh_cloud *c, int32_t rank, int32_t world, const void *unique_id, rh_comm **out);
int rh_comm_destroy(rh_comm *m);
int rh_score_batch_allreduce_dev(rh_cloud *c, rh_comm *m, const rh_shape *d_shapes, int32_t b, int32_t offset, int32_t b_total,
                                 const rh_params *p, int32_t *d_counts_total);
int rh_comm_fence(rh_comm *m, rh_cloud *c);
int rh_comm_sync(rh_comm *m);

/* wall time of the rh_cloud_create call that made the cloud, ms: [0] total, [1] the k-d leaf order of subset 1 (what
 * gives the culled score kernel its compact 64-point groups; built on the device, a radix sort per tree level --
 * RH_KD_HOST=1 keeps the host's nth_element recursion as the A/B), [2] before it (allocations, uploads, AoS -> SoA,
 * bounding box, Morton order of the cloud on the device), [3] after it (enabled bits).  RH_CREATE_PROF=1 prints the
 * stages on stderr. */
int rh_cloud_create_ms(const rh_cloud *c, double *out4);

/* ---- the reference's octree: pc.octree (src/octree.jl:237-244 buildoctree, :158-177 OctreeRefinery / needs_refinement /
 * refine_data, :187-196 iswithinrectangle, :212-230 octreedepth, :11-22 getnthcell; utilities.jl:125-136 findAABB;
 * RegionTrees' findleaf as src/fitting.jl:397 uses it; the enabled-cell gather of fitting.jl:405-407) ----
 * Cells are numbered from 0 (the root) in creation order; depth of the root is 1 (octree.jl:240).  The tree has the
 * reference's geometry with its quirks (root = Cell(minV, maxV) with maxV read as widths; vmin < p <= vmax membership;
 * refinement while a cell holds more than 8 points, stopped at depth 48: *overflow).  It never changes what ransac()
 * returns (every sample comes from the root cell) and rh_ransac does not build it. */
typedef struct rh_octree rh_octree;
int rh_octree_build(const double *xyz_aos, int64_t n, rh_octree **out);                 /* buildoctree(vertices); host-side set-up */
/* the tree of a Float32 cloud (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109): findAABB, the divisions origin +
 * widths / 2, the children's widths and the vmin < p <= vmax tests are binary32 operations there, so cells and their point lists
 * can differ from the binary64 tree of the same (widened) points.  The queries below serve both kinds of tree (the geometry
 * comes back widened to double, exactly). */
int rh_octree_build_f32(const float *xyz_aos, int64_t n, rh_octree **out);
int rh_octree_destroy(rh_octree *t);
int rh_octree_info(const rh_octree *t, int32_t *n_cells, int32_t *octreedepth, int32_t *overflow);
int rh_octree_findleaf(const rh_octree *t, const double *p3, int32_t *cell_out);        /* findleaf(pc.octree, p) */
int rh_octree_getnthcell(const rh_octree *t, int32_t cell, int32_t n, int32_t *cell_out /* -1 = nothing */);
int rh_octree_node_info(const rh_octree *t, int32_t cell, double *origin3, double *widths3, int32_t *depth, int32_t *parent,
                        int32_t *children8 /* -1: a leaf */, int64_t *npoints);
int rh_octree_node_points(const rh_octree *t, int32_t cell, int64_t *idx_out_1based, int64_t cap);   /* cell.data.incellpoints */
/* cell.data.incellpoints[pc.isenabled[cell.data.incellpoints]] (fitting.jl:405-407) on the device, against cloud c's bits */
int rh_octree_cell_enabled(rh_cloud *c, rh_octree *t, int32_t cell, int64_t *idx_out_1based, int64_t cap, int64_t *n_out);

/* ---- tuning options ----
 * The library reads NO environment variable: what a caller may tune goes through this call, for one cloud or, with
 * cloud = NULL, process-wide (the value a cloud without its own setting sees).  value = RH_OPTION_UNSET clears a setting.
 *   "score_path"  RH_SCORE_PATH_AUTO (default: the culled kernel from 8192 subset points on) / _BRUTE / _GROUPS -- which
 *                 batched score kernel clouds created from now on use (scorecandidates!, src/fitting.jl:181-190); fixed when
 *                 a cloud is created because its internal point order depends on it: process-wide only
 *   "refit_path"  RH_REFIT_PATH_AUTO (default: the culled scan from 2^21 points on) / _SCAN / _CULLED -- which full-cloud
 *                 scan rh_refit and rh_ransac take (refit, src/shapes/plane.jl:137-143); read on every refit
 *   "s4_rows"     0 (default: by the launch's size) / 4 / 8 / 12 / 16 -- 64-candidate chunks per block row of the culled
 *                 score kernel; read on every launch
 *   "unp_words"   0 (default) or the segment width, in 64-bit words, of the pass that turns the score kernel's inlier
 *                 lists into dense subset-order mask rows (inpoints, src/shapes/plane.jl:68); read on every call
 *   "st_cull"     0 (default: by the launch's size) / 1 (whenever possible) / 2 (never) -- a small launch in front of the
 *                 culled score kernel tests every candidate against the boxes of the SUPER-TILES (16 groups of 64 points)
 *                 and leaves per super-tile the list of candidates that may meet it; the score kernel's blocks then walk
 *                 those lists instead of every candidate of the batch.  Read on every launch.
 *   "batches_in_flight"  1 (default) .. 4 -- with F > 1, rh_score_batch_dev calls take turns on the cloud's
 *                 stream and F - 1 more with workspaces of their own, so one batch's prepare + score launches start while
 *                 the previous batches' launches drain.  The caller keeps F count (and mask) buffers and gives call k of a
 *                 run of such calls buffer k mod F (a buffer is written again only by the stream that wrote it last; a call
 *                 that hands in a buffer another batch in flight is writing is simply run after it, alone);
 *                 every other call on the cloud, rh_cloud_sync and rh_timer_stop included, first lets the cloud's stream
 *                 wait for the others, so whatever follows sees every batch's counts.  Ignored while the cloud runs on a
 *                 caller's stream (rh_cloud_set_stream).
 * Results never depend on any of them (tests/test_parity_gpu.py runs every parity test under both score paths).
 * Unknown keys and out-of-range values: RH_E_INVALID.  The diag build (libransac_hip_diag.so, include/ransac_hip_diag.h)
 * knows more keys -- the A/B switches of the experiments -- and falls back to RH_* environment variables; the product
 * build does neither. */
#define RH_OPTION_UNSET INT64_MIN
enum { RH_SCORE_PATH_AUTO = 0, RH_SCORE_PATH_BRUTE = 1, RH_SCORE_PATH_GROUPS = 2 };
enum { RH_REFIT_PATH_AUTO = 0, RH_REFIT_PATH_SCAN = 1, RH_REFIT_PATH_CULLED = 2 };
int rh_set_option(rh_cloud *c_or_null, const char *key, int64_t value);
int rh_get_option(const rh_cloud *c_or_null, const char *key, int64_t *value_out, int32_t *is_set_out_or_null);
int rh_build_variant(void);   /* 0 = product, 1 = diag (-DRH_DIAG) */
/* what the cloud's last batch launch of the culled score kernel looked like (bench.py names the measured kernel with it):
 * out4 = { chunks of 64 candidates per block row, 1 if the rows walked super-tile lists ("st_cull") else 0, rows, tiles } */
int rh_score_launch_info(rh_cloud *c, int32_t *out4);

#ifdef __cplusplus
}
#endif
#endif
