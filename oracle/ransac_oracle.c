/*
 * ransac_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see ransac_oracle.h).
 *
 * Plain-C restatement of cserteGT3/RANSAC.jl v0.6.0 for the hot path.  Paths in
 * comments are under /root/reference.  All floating point is binary64, written
 * in the reference's operation order; compile with -ffp-contract=off.
 *
 * StaticArrays semantics used throughout [recalled, compat 0.11-1.2]:
 *   dot(a,b)      = (a1*b1 + a2*b2) + a3*b3          (no muladd)
 *   norm(a)       = sqrt((a1*a1 + a2*a2) + a3*a3)
 *   normalize(a)  = inv(norm(a)) * a                  (reciprocal-multiply)
 *   cross(a,b)    = (a2*b3-a3*b2, a3*b1-a1*b3, a1*b2-a2*b1)
 *   (M*v)_i       = (M_i1*v1 + M_i2*v2) + M_i3*v3
 *
 * ROUNDING-ORDER VARIANTS (-DORC_VARIANT=n, see Makefile: liboracle_v<n>.so).  The semantics above
 * cannot be checked against Julia here, so tests/golden/make_rounding_sensitivity.py measures how many
 * inlier decisions change when they are replaced by the other plausible readings:
 *   0  the default above
 *   1  "fma":      every a*b + c of dot / norm / cross / M*v and of the shape formulas fused (muladd)
 *   2  "div":      normalize(a) = a / norm(a)
 *   3  "scaled":   norm(a) = m * sqrt(sum((a_i / m)^2)), m = max |a_i|   (LinearAlgebra.generic_norm2)
 *   4  "pairwise": dot(a,b) = a1*b1 + (a2*b2 + a3*b3), same for the sum of squares
 *   5  "libm":     the cone's acos / cos / sin from the platform libm instead of orc_trig.h
 * The variants are test tooling for that measurement only; every parity test uses variant 0.
 */
#include "ransac_oracle.h"

#include <math.h>

#ifndef ORC_VARIANT
#define ORC_VARIANT 0
#endif

/* acos / cos / sin of the cone code (fit3pointcone cone.jl:58, rodrigues utilities.jl:21-22): the oracle's
 * OWN restatement of the fdlibm algorithms (orc_trig.h; what Julia's Base.acos / sin / cos port) -- nothing
 * of the product is included here.  tests/test_oracle_golden.py checks it against the platform libm
 * (<= 1 ulp) and, bit for bit, against the product's rh_* twins through the C ABI. */
#include "orc_trig.h"
#if ORC_VARIANT == 5
#define ORC_ACOS(x) acos(x)
#define ORC_COS(x) cos(x)
#define ORC_SIN(x) sin(x)
#else
#define ORC_ACOS(x) orc_acos(x)
#define ORC_COS(x) orc_cos(x)
#define ORC_SIN(x) orc_sin(x)
#endif
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ vec3 */
typedef struct { double x, y, z; } v3;

static inline v3 V(const double *p) { v3 r = { p[0], p[1], p[2] }; return r; }
static inline v3 vsub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline v3 vadd(v3 a, v3 b) { v3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline v3 vscale(v3 a, double s) { v3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline v3 vdiv(v3 a, double s) { v3 r = { a.x / s, a.y / s, a.z / s }; return r; }
static inline v3 vneg(v3 a) { v3 r = { -a.x, -a.y, -a.z }; return r; }
/* a*b + c and a*b - c*d as the variant reads them (MA: one rounding with "fma") */
#if ORC_VARIANT == 1
#define MA(a, b, c) fma((a), (b), (c))
#define MS2(a, b, c, d) fma((a), (b), -((c) * (d)))      /* a*b - c*d */
#else
#define MA(a, b, c) ((a) * (b) + (c))
#define MS2(a, b, c, d) ((a) * (b) - (c) * (d))
#endif
#if ORC_VARIANT == 4
static inline double sum3(double a1b1, double a2, double b2, double a3, double b3) { return a1b1 + (a2 * b2 + a3 * b3); }
#else
static inline double sum3(double a1b1, double a2, double b2, double a3, double b3) { return MA(a3, b3, MA(a2, b2, a1b1)); }
#endif
static inline double vdot(v3 a, v3 b) { return sum3(a.x * b.x, a.y, b.y, a.z, b.z); }
#if ORC_VARIANT == 3
static inline double vnorm(v3 a)
{
    double m = fabs(a.x) > fabs(a.y) ? fabs(a.x) : fabs(a.y);
    if (fabs(a.z) > m) m = fabs(a.z);
    if (m == 0.0 || m != m || m > 1.7976931348623157e308) return sqrt((a.x * a.x + a.y * a.y) + a.z * a.z);
    double x = a.x / m, y = a.y / m, z = a.z / m;
    return m * sqrt((x * x + y * y) + z * z);
}
#else
static inline double vnorm(v3 a) { return sqrt(sum3(a.x * a.x, a.y, a.y, a.z, a.z)); }
#endif
#if ORC_VARIANT == 2
static inline v3 vnormalize(v3 a) { double nr = vnorm(a); v3 r = { a.x / nr, a.y / nr, a.z / nr }; return r; }
#else
static inline v3 vnormalize(v3 a) { double inv = 1.0 / vnorm(a); v3 r = { inv * a.x, inv * a.y, inv * a.z }; return r; }
#endif
static inline v3 vcross(v3 a, v3 b)
{
    v3 r = { MS2(a.y, b.z, a.z, b.y), MS2(a.z, b.x, a.x, b.z), MS2(a.x, b.y, a.y, b.x) };
    return r;
}

/* ------------------------------------------------------------ parameters */
#define ORC_PI 3.14159265358979323846

void orc_params_finalize(orc_params *p)
{
    for (int k = 0; k < 4; k++) p->cos_alpha[k] = cos(p->alpha[k]);
    /* cosd(parallelthrdeg), sphere.jl:37, cylinder.jl:40 */
    p->cos_parallelthr = cos(p->parallelthrdeg * ORC_PI / 180.0);
}

/* utilities.jl:345,371; plane.jl:22; sphere.jl:25; cylinder.jl:27; cone.jl:30;
 * shape_types order of DEFAULT_PARAMETERS: RANSAC.jl:94 */
void orc_default_params(orc_params *p)
{
    memset(p, 0, sizeof *p);
    for (int k = 0; k < 4; k++) { p->eps[k] = 0.3; p->alpha[k] = 5.0 * ORC_PI / 180.0; }
    p->collin_threshold = 0.2;
    p->parallelthrdeg = 1.0;
    p->sphere_par = 0.02;
    p->minconeopang = 2.0 * ORC_PI / 180.0;
    p->prob_det = 0.9;
    p->tau = 900;
    p->itermax = 1000;
    p->drawN = 3;
    p->minsubsetN = 15;
    p->extract_s = ORC_S_NOFMINSET;
    p->terminate_s = ORC_S_NOFMINSET;
    p->n_shape_types = 4;
    p->shape_types[0] = ORC_PLANE;
    p->shape_types[1] = ORC_CONE;
    p->shape_types[2] = ORC_CYLINDER;
    p->shape_types[3] = ORC_SPHERE;
    p->score_mode = ORC_SCORE_INT64_WRAP;
    p->sphere_uses_enabled = 0;
    p->octree_max_depth = 10;
    orc_params_finalize(p);
}

void orc_shape_finalize(orc_shape *s)
{
    if (s->kind == ORC_CONE) {
        /* rodriguesrad(rot_ax, -cone.opang/2): cone.jl:76; cos/sin at utilities.jl:21-22 */
        double th = -s->v[6] / 2;
        s->v[7] = ORC_COS(th);
        s->v[8] = ORC_SIN(th);
    }
}

/* ----------------------------------------------- rodrigues (utilities.jl) */
/* pluscrossprod!(A, value, v): utilities.jl:32-43.  A row-major 3x3. */
void orc_pluscrossprod(double A[9], double value, const double v[3])
{
    A[0 * 3 + 1] = MA(-value, v[2], A[0 * 3 + 1]);
    A[0 * 3 + 2] = MA(value, v[1], A[0 * 3 + 2]);
    A[1 * 3 + 0] = MA(value, v[2], A[1 * 3 + 0]);
    A[1 * 3 + 2] = MA(-value, v[0], A[1 * 3 + 2]);
    A[2 * 3 + 0] = MA(-value, v[1], A[2 * 3 + 0]);
    A[2 * 3 + 1] = MA(value, v[0], A[2 * 3 + 1]);
}

/* rodrigues(nv::StaticArray, theta) with precomputed cos/sin: utilities.jl:19-24
 * R = nv*nv' + cos .* (I - nv*nv'); pluscrossprod!(R, sin, nv) */
static void rodrigues_cs(v3 nv, double c, double s, double R[9])
{
    double n[3] = { nv.x, nv.y, nv.z };
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double nn = n[i] * n[j];
            double id = (i == j) ? 1.0 : 0.0;
            R[i * 3 + j] = MA(c, id - nn, nn);
        }
    orc_pluscrossprod(R, s, n);
}

/* rodriguesrad(nv, theta): utilities.jl:61-64 -- re-normalizes the axis */
void orc_rodriguesrad(const double nv[3], double theta, double R[9])
{
    v3 nvn = vnormalize(V(nv));
    rodrigues_cs(nvn, cos(theta), sin(theta), R);
}

/* ---------------------------------------------- per-point compatibility */

/* compatiblesPlane: plane.jl:114-130 with project2plane plane.jl:82-95
 * (o_z = normalize(plane.normal); third coordinate dot(o_z, p - plane.point));
 * isparallel utilities.jl:115-117.  The two in-plane coordinates are unused. */
static int compat_plane(const orc_shape *s, v3 p, v3 n, double eps, double cosa)
{
    v3 point = V(&s->v[0]), normal = V(&s->v[3]);
    v3 o_z = vnormalize(normal);
    v3 v = vsub(p, point);
    double d = vdot(o_z, v);
    return (vdot(normal, n) > cosa) && (fabs(d) < eps);
}

/* compatiblesSphere: sphere.jl:144-172 */
static int compat_sphere(const orc_shape *s, v3 p, v3 n, double eps, double cosa)
{
    v3 o = V(&s->v[0]);
    double R = s->v[3];
    if (s->outwards)
        return (vdot(vnormalize(vsub(p, o)), n) > cosa) && (fabs(vnorm(vsub(p, o)) - R) < eps);
    else
        return (vdot(vnormalize(vsub(o, p)), n) > cosa) && (fabs(vnorm(vsub(p, o)) - R) < eps);
}

/* compatiblesCylinder: cylinder.jl:194-221 */
static int compat_cylinder(const orc_shape *s, v3 p, v3 n, double eps, double cosa)
{
    v3 a = V(&s->v[0]), c = V(&s->v[3]);
    double R = s->v[6];
    /* curr_norm = points[i] - a*dot(a, points[i]-c) - c */
    double sd = vdot(a, vsub(p, c));
    v3 pa = { MA(-a.x, sd, p.x), MA(-a.y, sd, p.y), MA(-a.z, sd, p.z) };   /* p - a*sd */
    v3 curr_norm = vsub(pa, c);
    if (fabs(vnorm(curr_norm) - R) < eps) {
        if (s->outwards)
            return vdot(vnormalize(curr_norm), n) > cosa;
        else
            return vdot(vneg(vnormalize(curr_norm)), n) > cosa;
    }
    return 0;
}

/* project2cone: cone.jl:68-85 */
static void project2cone(const orc_shape *s, v3 p, double *dist, v3 *normal)
{
    v3 apex = V(&s->v[0]), axis = V(&s->v[3]);
    v3 to_point = vsub(apex, p);
    v3 to_pointn = vnormalize(to_point);
    v3 rot_ax = vnormalize(vcross(axis, to_pointn));
    v3 comp_n = vnormalize(vcross(axis, rot_ax));
    /* rM = rodriguesrad(rot_ax, -cone.opang/2) */
    double R[9];
    v3 nvn = vnormalize(rot_ax);
    rodrigues_cs(nvn, s->v[7], s->v[8], R);
    v3 rc = {
        sum3(R[0] * comp_n.x, R[1], comp_n.y, R[2], comp_n.z),
        sum3(R[3] * comp_n.x, R[4], comp_n.y, R[5], comp_n.z),
        sum3(R[6] * comp_n.x, R[7], comp_n.y, R[8], comp_n.z),
    };
    v3 current_normal = vnormalize(rc);
    *dist = vdot(vneg(current_normal), vneg(to_point));
    *normal = current_normal;
}

/* compatiblesCone: cone.jl:132-153 */
static int compat_cone(const orc_shape *s, v3 p, v3 n, double eps, double cosa)
{
    double dist;
    v3 cn;
    project2cone(s, p, &dist, &cn);
    if (s->outwards)
        return (vdot(cn, n) > cosa) && (fabs(dist) < eps);
    else
        return (vdot(vneg(cn), n) > cosa) && (fabs(dist) < eps);
}

static inline int compat(const orc_shape *s, v3 p, v3 n, double eps, double cosa)
{
    switch (s->kind) {
    case ORC_PLANE: return compat_plane(s, p, n, eps, cosa);
    case ORC_SPHERE: return compat_sphere(s, p, n, eps, cosa);
    case ORC_CYLINDER: return compat_cylinder(s, p, n, eps, cosa);
    case ORC_CONE: return compat_cone(s, p, n, eps, cosa);
    }
    return 0;
}

int orc_compatible(const orc_shape *s, const double p[3], const double n[3], double eps, double cos_alpha)
{
    return compat(s, V(p), V(n), eps, cos_alpha);
}

/* The two compared quantities of one test, for the near-threshold census of
 * tests/golden/make_rounding_sensitivity.py: out[0] = the distance side (|d|, |norm - R| or |dist|, compared
 * with eps), out[1] = the angle side (the dot product compared with cos(alpha)). */
void orc_compat_values(const orc_shape *s, const double pp[3], const double nn[3], double out[2])
{
    v3 p = V(pp), n = V(nn);
    switch (s->kind) {
    case ORC_PLANE: {
        v3 normal = V(&s->v[3]);
        out[0] = fabs(vdot(vnormalize(normal), vsub(p, V(&s->v[0]))));
        out[1] = vdot(normal, n);
        break;
    }
    case ORC_SPHERE: {
        v3 o = V(&s->v[0]);
        out[0] = fabs(vnorm(vsub(p, o)) - s->v[3]);
        out[1] = s->outwards ? vdot(vnormalize(vsub(p, o)), n) : vdot(vnormalize(vsub(o, p)), n);
        break;
    }
    case ORC_CYLINDER: {
        v3 a = V(&s->v[0]), c = V(&s->v[3]);
        double sd = vdot(a, vsub(p, c));
        v3 pa = { MA(-a.x, sd, p.x), MA(-a.y, sd, p.y), MA(-a.z, sd, p.z) };
        v3 cn = vsub(pa, c);
        out[0] = fabs(vnorm(cn) - s->v[6]);
        out[1] = s->outwards ? vdot(vnormalize(cn), n) : vdot(vneg(vnormalize(cn)), n);
        break;
    }
    default: {
        double dist;
        v3 cn;
        project2cone(s, p, &dist, &cn);
        out[0] = fabs(dist);
        out[1] = s->outwards ? vdot(cn, n) : vdot(vneg(cn), n);
    }
    }
}

int orc_variant(void) { return ORC_VARIANT; }

/* ------------------------------------------------- confidence intervals */

/* ConfidenceInterval(x, y): confidenceintervals.jl:1-6 */
int orc_confidence_interval(double x, double y, orc_ci *out)
{
    if (x > y) return -1; /* error("out of order") */
    out->min = x;
    out->max = y;
    out->E = (x + y) / 2;
    return 0;
}

/* notsoconfident: confidenceintervals.jl:20-22.  Julia min/max propagate NaN. */
static double jl_min(double x, double y) { return (x != x) ? x : (y != y) ? y : (y < x ? y : x); }
static double jl_max(double x, double y) { return (x != x) ? x : (y != y) ? y : (x < y ? y : x); }

orc_ci orc_notsoconfident(double x, double y)
{
    orc_ci c;
    c.min = jl_min(x, y);
    c.max = jl_max(x, y);
    c.E = (c.min + c.max) / 2;
    return c;
}

/* isoverlap: confidenceintervals.jl:29-36 */
int orc_isoverlap(orc_ci i1, orc_ci i2)
{
    if (i1.min == i2.min) return 1;
    if (i1.min < i2.min) return i2.min <= i1.max;
    if (i2.min < i1.min) return i1.min <= i2.max;
    return 0; /* NaN: the reference would recurse forever; unreachable for real scores */
}

/* estimatescore + hypergeomdev: confidenceintervals.jl:71-74, 53-59.
 * INT64_WRAP reproduces Julia's silent Int64 wrap of x*n*(N-x)*(N-n). */
orc_ci orc_estimatescore(int64_t S1length, int64_t Plength, int64_t sigma, int score_mode)
{
    int64_t N = -2 - S1length, x = -2 - Plength, n = -1 - sigma;
    double sq_, xn;
    if (score_mode == ORC_SCORE_INT64_WRAP) {
        uint64_t prod = (uint64_t)x * (uint64_t)n;
        int64_t xn_i = (int64_t)prod;
        prod = prod * (uint64_t)(N - x);
        prod = prod * (uint64_t)(N - n);
        sq_ = (double)(int64_t)prod / (double)(N - 1);
        xn = (double)xn_i;
    } else {
        double xd = (double)x, nd = (double)n, Nd = (double)N;
        sq_ = (xd * nd * (Nd - xd) * (Nd - nd)) / (Nd - 1);
        xn = xd * nd;
    }
    double sq = sq_ < 0 ? 0.0 : sqrt(sq_);
    double gmin = (xn + sq) / (double)N;
    double gmax = (xn - sq) / (double)N;
    return orc_notsoconfident(-1 - gmin, -1 - gmax);
}

/* Decision trace (tests/golden/make_rounding_sensitivity.py): every evaluation of the extraction test
 * `prob(E(best), s, N, drawN) > prob_det` (iterations.jl:114-123) of a run, four doubles each:
 * iteration, best score E, s, ppp.  Off unless a buffer is set. */
static double *g_trace_buf = NULL;
static int64_t g_trace_cap = 0, g_trace_n = 0;
void orc_trace_set(double *buf, int64_t cap_records) { g_trace_buf = buf; g_trace_cap = cap_records; g_trace_n = 0; }
int64_t orc_trace_count(void) { return g_trace_n; }

/* prob(n, s, N, k) = 1-(1-(n/N)^k)^s: utilities.jl:262.  Float64^Int is libm pow
 * in Julia <= 1.7 (llvm.pow); later versions differ in the last ulp [recalled]. */
double orc_prob(double n, int64_t s, int64_t N, int64_t k)
{
    return 1 - pow(1 - pow(n / (double)N, (double)k), (double)s);
}

/* ----------------------------------------------------------------- cloud */
struct orc_cloud {
    int64_t n, s, nchunks;
    double *xyz, *nrm;   /* AoS, like Vector{SVector{3,Float64}} (octree.jl:106-107) */
    int64_t *subset1;    /* 1-based */
    uint64_t *enabled;   /* BitVector chunks, LSB first */
    int64_t *dir;        /* select directory: enabled count before each 64-chunk block */
    int dir_valid;
    int f32;             /* the values are Float32 numbers and every per-point test / fit is the binary32 one (orc_f32.c) */
};

#define DIR_BLOCK 64 /* chunks per directory block (4096 bits) */

orc_cloud *orc_cloud_create(const double *xyz, const double *nrm, int64_t n,
                            const int64_t *subset1, int64_t s)
{
    orc_cloud *c = (orc_cloud *)calloc(1, sizeof *c);
    c->n = n;
    c->s = s;
    c->nchunks = (n + 63) / 64;
    c->xyz = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    c->nrm = (double *)malloc(sizeof(double) * 3 * (size_t)(n ? n : 1));
    memcpy(c->xyz, xyz, sizeof(double) * 3 * (size_t)n);
    memcpy(c->nrm, nrm, sizeof(double) * 3 * (size_t)n);
    c->subset1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s ? s : 1));
    memcpy(c->subset1, subset1, sizeof(int64_t) * (size_t)s);
    c->enabled = (uint64_t *)calloc((size_t)(c->nchunks ? c->nchunks : 1), 8);
    c->dir = (int64_t *)calloc((size_t)(c->nchunks / DIR_BLOCK + 2), 8);
    orc_cloud_enable_all(c); /* trues(s): octree.jl:84 */
    return c;
}

void orc_cloud_destroy(orc_cloud *c)
{
    if (!c) return;
    free(c->xyz); free(c->nrm); free(c->subset1); free(c->enabled); free(c->dir);
    free(c);
}

void orc_cloud_enable_all(orc_cloud *c)
{
    for (int64_t i = 0; i < c->nchunks; i++) c->enabled[i] = ~0ULL;
    if (c->n % 64) c->enabled[c->nchunks - 1] = (~0ULL) >> (64 - c->n % 64);
    c->dir_valid = 0;
}

void orc_cloud_set_enabled(orc_cloud *c, const uint64_t *chunks, int64_t nchunks)
{
    int64_t m = nchunks < c->nchunks ? nchunks : c->nchunks;
    memcpy(c->enabled, chunks, 8 * (size_t)m);
    if (c->n % 64 && m == c->nchunks) c->enabled[c->nchunks - 1] &= (~0ULL) >> (64 - c->n % 64);
    c->dir_valid = 0;
}

void orc_cloud_get_enabled(const orc_cloud *c, uint64_t *chunks, int64_t nchunks)
{
    int64_t m = nchunks < c->nchunks ? nchunks : c->nchunks;
    memcpy(chunks, c->enabled, 8 * (size_t)m);
}

int64_t orc_cloud_count_enabled(const orc_cloud *c)
{
    int64_t t = 0;
    for (int64_t i = 0; i < c->nchunks; i++) t += __builtin_popcountll(c->enabled[i]);
    return t;
}

static inline int is_enabled(const orc_cloud *c, int64_t i0) { return (int)((c->enabled[i0 >> 6] >> (i0 & 63)) & 1); }

void orc_cloud_set_f32(orc_cloud *c, int f32) { c->f32 = f32 ? 1 : 0; }

/* the per-point test of the cloud's element type */
static inline int compat_pt(const orc_cloud *c, const orc_shape *s, int64_t i0, double eps, double cosa)
{
    if (c->f32) {
        const double *pp = &c->xyz[3 * i0], *nn = &c->nrm[3 * i0];
        float pf[3] = { (float)pp[0], (float)pp[1], (float)pp[2] }, nf[3] = { (float)nn[0], (float)nn[1], (float)nn[2] };
        return orc32_compatible(s, pf, nf, eps, cosa);
    }
    return compat(s, V(&c->xyz[3 * i0]), V(&c->nrm[3 * i0]), eps, cosa);
}

/* scorecandidate: plane.jl:61-71, sphere.jl:118-134, cylinder.jl:172-183, cone.jl:155-167.
 * Points are gathered through subsets[1] exactly like the reference's views. */
int64_t orc_scorecandidate(const orc_cloud *c, const orc_shape *s, const orc_params *p,
                           int64_t *inpoints, uint64_t *mask)
{
    double eps = p->eps[s->kind], cosa = p->cos_alpha[s->kind];
    int use_en = (s->kind != ORC_SPHERE) || p->sphere_uses_enabled; /* Q4: sphere.jl:121,131 */
    int64_t cnt = 0;
    if (mask) memset(mask, 0, 8 * (size_t)((c->s + 63) / 64));
    for (int64_t j = 0; j < c->s; j++) {
        int64_t i0 = c->subset1[j] - 1;
        int ok = compat_pt(c, s, i0, eps, cosa);
        if (use_en) ok = ok & is_enabled(c, i0);
        if (ok) {
            if (inpoints) inpoints[cnt] = i0 + 1;
            if (mask) mask[j >> 6] |= 1ULL << (j & 63);
            cnt++;
        }
    }
    return cnt;
}

void orc_score_batch(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                     int32_t *counts, uint64_t *masks)
{
    int64_t w = (c->s + 63) / 64;
    for (int32_t i = 0; i < b; i++)
        counts[i] = (int32_t)orc_scorecandidate(c, &s[i], p, NULL, masks ? masks + (size_t)i * w : NULL);
}

/* Steelman of the CPU side for bench.py: the same per-candidate passes spread over host threads (the
 * reference has no threading at all; candidates are independent, fitting.jl:182-186). */
void orc_score_batch_mt(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                        int32_t *counts, int32_t nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int32_t i = 0; i < b; i++)
        counts[i] = (int32_t)orc_scorecandidate(c, &s[i], p, NULL, NULL);
}

/* masks (b x ceil(s/64) words, subset order) with the candidates spread over host threads: tooling of the
 * rounding-sensitivity measurement and of the bench's oracle checks */
void orc_score_masks_mt(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                        int32_t *counts, uint64_t *masks, int32_t nthreads)
{
    int64_t w = (c->s + 63) / 64;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int32_t i = 0; i < b; i++)
        counts[i] = (int32_t)orc_scorecandidate(c, &s[i], p, NULL, masks ? masks + (size_t)i * w : NULL);
}

/* census of near-threshold decisions over subset 1: hist[0][k] counts tests whose distance side lies within
 * edges[k] of eps, hist[1][k] the same for the angle side and cos(alpha) (absolute differences; nedges <= 8) */
void orc_margin_census_mt(const orc_cloud *c, const orc_shape *s, int32_t b, const orc_params *p,
                          const double *edges, int nedges, int64_t *hist /* 2 x nedges */, int32_t nthreads)
{
    if (nthreads < 1) nthreads = 1;
    int64_t h0[8] = { 0 }, h1[8] = { 0 };
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : h0[:8], h1[:8])
    for (int32_t i = 0; i < b; i++) {
        double eps = p->eps[s[i].kind], cosa = p->cos_alpha[s[i].kind];
        for (int64_t j = 0; j < c->s; j++) {
            int64_t i0 = c->subset1[j] - 1;
            double v[2];
            orc_compat_values(&s[i], &c->xyz[3 * i0], &c->nrm[3 * i0], v);
            double m0 = fabs(v[0] - eps), m1 = fabs(v[1] - cosa);
            for (int k = 0; k < nedges; k++) {
                if (m0 <= edges[k]) h0[k]++;
                if (m1 <= edges[k]) h1[k]++;
            }
        }
    }
    for (int k = 0; k < nedges; k++) { hist[k] = h0[k]; hist[nedges + k] = h1[k]; }
}

/* refit: plane.jl:137-143, sphere.jl:179-190, cylinder.jl:228-234, cone.jl:176-182 */
int64_t orc_refit(const orc_cloud *c, const orc_shape *s, const orc_params *p, int64_t *idx_out, int64_t cap)
{
    double eps = p->eps[s->kind], cosa = p->cos_alpha[s->kind];
    int64_t cnt = 0;
    for (int64_t i0 = 0; i0 < c->n; i0++) {
        if (!is_enabled(c, i0)) continue;
        if (compat_pt(c, s, i0, eps, cosa)) {
            if (cnt < cap) idx_out[cnt] = i0 + 1;
            cnt++;
        }
    }
    return cnt;
}

/* invalidate_indexes!: fitting.jl:197-202 */
void orc_invalidate(orc_cloud *c, const int64_t *idx, int64_t n)
{
    for (int64_t k = 0; k < n; k++) {
        int64_t i0 = idx[k] - 1;
        c->enabled[i0 >> 6] &= ~(1ULL << (i0 & 63));
    }
    c->dir_valid = 0;
}

/* ---- least-squares refit (specification shared with ransac.jl_amd/csrc/lsq.hip) ---- */
static void lsq_unit(double *v) { double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= n; v[1] /= n; v[2] /= n; }

static void lsq_frame(const double a[3], double e1[3], double e2[3])
{
    int k = 0;
    if (fabs(a[1]) < fabs(a[k])) k = 1;
    if (fabs(a[2]) < fabs(a[k])) k = 2;
    double t[3] = { 0, 0, 0 };
    t[k] = 1.0;
    e1[0] = a[1] * t[2] - a[2] * t[1]; e1[1] = a[2] * t[0] - a[0] * t[2]; e1[2] = a[0] * t[1] - a[1] * t[0];
    lsq_unit(e1);
    e2[0] = a[1] * e1[2] - a[2] * e1[1]; e2[1] = a[2] * e1[0] - a[0] * e1[2]; e2[2] = a[0] * e1[1] - a[1] * e1[0];
    lsq_unit(e2);
}

static int lsq_solve(int m, const double *A, const double *b, double *x)
{
    double M[8][9];
    for (int i = 0; i < m; i++) { for (int j = 0; j < m; j++) M[i][j] = A[i * 8 + j]; M[i][m] = b[i]; }
    for (int k = 0; k < m; k++) {
        int piv = k;
        for (int i = k + 1; i < m; i++) if (fabs(M[i][k]) > fabs(M[piv][k])) piv = i;
        if (M[piv][k] == 0.0 || !(M[piv][k] == M[piv][k])) return 0;
        if (piv != k) for (int j = 0; j <= m; j++) { double t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
        for (int i = k + 1; i < m; i++) {
            double l = M[i][k] / M[k][k];
            for (int j = k; j <= m; j++) M[i][j] -= l * M[k][j];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = M[i][m];
        for (int j = i + 1; j < m; j++) s -= M[i][j] * x[j];
        x[i] = s / M[i][i];
    }
    return 1;
}

static void lsq_eig3(double S[3][3], double evec[3][3], double eval[3])
{
    double V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (S[p][q] == 0.0) continue;
                double theta = (S[q][q] - S[p][p]) / (2 * S[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) { double a = S[k][p], b = S[k][q]; S[k][p] = c * a - s * b; S[k][q] = s * a + c * b; }
                for (int k = 0; k < 3; k++) { double a = S[p][k], b = S[q][k]; S[p][k] = c * a - s * b; S[q][k] = s * a + c * b; }
                for (int k = 0; k < 3; k++) { double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
            }
    }
    for (int i = 0; i < 3; i++) { eval[i] = S[i][i]; for (int k = 0; k < 3; k++) evec[i][k] = V[k][i]; }
}

int orc_refit_lsq(const orc_cloud *c, const orc_shape *shape, const orc_params *p, int max_iter, orc_shape *out,
                  int64_t *n_used, double *rms, int *iters_done)
{
    int kind = shape->kind;
    if (max_iter < 1) max_iter = 1;
    double eps3 = 3.0 * p->eps[kind], cosa = p->cos_alpha[kind];
    uint8_t *sel = (uint8_t *)calloc((size_t)(c->n ? c->n : 1), 1);
    int64_t cnt = 0;
    for (int64_t i = 0; i < c->n; i++)
        if (is_enabled(c, i) && compat(shape, V(&c->xyz[3 * i]), V(&c->nrm[3 * i]), eps3, cosa)) { sel[i] = 1; cnt++; }
    if (cnt < 8) { free(sel); return -1; }
    orc_shape cur = *shape;
    double last = 0;
    int it;
    for (it = 0; it < max_iter; it++) {
        double M[64];
        memset(M, 0, sizeof M);
        double e1[3] = { 0, 0, 0 }, e2[3] = { 0, 0, 0 }, cph = 0, sph = 0;
        if (kind == ORC_CYLINDER) lsq_frame(&cur.v[0], e1, e2);
        if (kind == ORC_CONE) { lsq_frame(&cur.v[3], e1, e2); cph = cos(cur.v[6] / 2); sph = sin(cur.v[6] / 2); }
        for (int64_t i = 0; i < c->n; i++) {
            if (!sel[i]) continue;
            double row[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            const double *pp = &c->xyz[3 * i];
            if (kind == ORC_PLANE) {
                row[0] = pp[0] - cur.v[0]; row[1] = pp[1] - cur.v[1]; row[2] = pp[2] - cur.v[2]; row[3] = 1.0;
            } else if (kind == ORC_SPHERE) {
                double dx = pp[0] - cur.v[0], dy = pp[1] - cur.v[1], dz = pp[2] - cur.v[2];
                double nr = sqrt(dx * dx + dy * dy + dz * dz), inv = 1.0 / nr;
                row[0] = -dx * inv; row[1] = -dy * inv; row[2] = -dz * inv; row[3] = -1.0; row[4] = nr - cur.v[3];
            } else {
                const double *o = kind == ORC_CYLINDER ? &cur.v[3] : &cur.v[0];
                const double *a = kind == ORC_CYLINDER ? &cur.v[0] : &cur.v[3];
                double tx = pp[0] - o[0], ty = pp[1] - o[1], tz = pp[2] - o[2];
                double h = a[0] * tx + a[1] * ty + a[2] * tz;
                double qx = tx - a[0] * h, qy = ty - a[1] * h, qz = tz - a[2] * h;
                double rho = sqrt(qx * qx + qy * qy + qz * qz), inv = 1.0 / rho;
                double ux = qx * inv, uy = qy * inv, uz = qz * inv;
                double u1 = ux * e1[0] + uy * e1[1] + uz * e1[2], u2 = ux * e2[0] + uy * e2[1] + uz * e2[2];
                if (kind == ORC_CYLINDER) {
                    row[0] = -u1; row[1] = -u2; row[2] = -h * u1; row[3] = -h * u2; row[4] = -1.0; row[5] = rho - cur.v[6];
                } else {
                    double t1 = tx * e1[0] + ty * e1[1] + tz * e1[2], t2 = tx * e2[0] + ty * e2[1] + tz * e2[2];
                    row[0] = -ux * cph + a[0] * sph; row[1] = -uy * cph + a[1] * sph; row[2] = -uz * cph + a[2] * sph;
                    row[3] = -h * u1 * cph - t1 * sph; row[4] = -h * u2 * cph - t2 * sph;
                    row[5] = -rho * sph - h * cph; row[6] = rho * cph - h * sph;
                }
            }
            for (int a2 = 0; a2 < 8; a2++) for (int b2 = 0; b2 < 8; b2++) M[a2 * 8 + b2] += row[a2] * row[b2];
        }
        if (kind == ORC_PLANE) {
            double N = M[3 * 8 + 3];
            double sv[3] = { M[0 * 8 + 3], M[1 * 8 + 3], M[2 * 8 + 3] }, S[3][3], evec[3][3], eval[3];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S[i][j] = M[i * 8 + j] - sv[i] * sv[j] / N;
            lsq_eig3(S, evec, eval);
            int mn = 0;
            for (int i = 1; i < 3; i++) if (eval[i] < eval[mn]) mn = i;
            double nn[3] = { evec[mn][0], evec[mn][1], evec[mn][2] };
            lsq_unit(nn);
            if (nn[0] * shape->v[3] + nn[1] * shape->v[4] + nn[2] * shape->v[5] < 0) for (int i = 0; i < 3; i++) nn[i] = -nn[i];
            for (int i = 0; i < 3; i++) { cur.v[i] = cur.v[i] + sv[i] / N; cur.v[3 + i] = nn[i]; }
            last = sqrt((eval[mn] > 0 ? eval[mn] : 0.0) / N);
            it++;
            break;
        }
        int m = kind == ORC_SPHERE ? 4 : (kind == ORC_CYLINDER ? 5 : 6);
        double A[64], b[8], x[8], tr = 0;
        for (int i = 0; i < m; i++) tr += M[i * 8 + i];
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < m; j++) A[i * 8 + j] = M[i * 8 + j];
            A[i * 8 + i] += 1e-12 * tr;
            b[i] = -M[i * 8 + m];
        }
        last = M[m * 8 + m];
        if (!lsq_solve(m, A, b, x)) { free(sel); return -2; }
        if (kind == ORC_SPHERE) { for (int i = 0; i < 4; i++) cur.v[i] += x[i]; }
        else if (kind == ORC_CYLINDER) {
            for (int i = 0; i < 3; i++) { cur.v[3 + i] += x[0] * e1[i] + x[1] * e2[i]; cur.v[i] += x[2] * e1[i] + x[3] * e2[i]; }
            lsq_unit(&cur.v[0]);
            cur.v[6] += x[4];
        } else {
            for (int i = 0; i < 3; i++) { cur.v[i] += x[i]; cur.v[3 + i] += x[3] * e1[i] + x[4] * e2[i]; }
            lsq_unit(&cur.v[3]);
            cur.v[6] += 2 * x[5];
        }
        double step = 0;
        for (int i = 0; i < m; i++) step += x[i] * x[i];
        if (sqrt(step) < 1e-11) { it++; break; }
    }
    if (kind != ORC_PLANE) last = cnt > 0 ? sqrt(last / (double)cnt) : 0.0;
    orc_shape_finalize(&cur);
    *out = cur;
    if (n_used) *n_used = cnt;
    if (rms) *rms = last;
    if (iters_done) *iters_done = it;
    free(sel);
    return 0;
}

static void build_dir(orc_cloud *c)
{
    int64_t acc = 0, nb = c->nchunks / DIR_BLOCK + 1;
    for (int64_t b = 0; b < nb; b++) {
        c->dir[b] = acc;
        int64_t lo = b * DIR_BLOCK, hi = lo + DIR_BLOCK;
        if (hi > c->nchunks) hi = c->nchunks;
        for (int64_t i = lo; i < hi; i++) acc += __builtin_popcountll(c->enabled[i]);
    }
    c->dir[nb] = acc;
    c->dir_valid = 1;
}

/* enabled_inds[k] for the root cell (incellpoints = 1:N, octree.jl:240;
 * fitting.jl:405-407): the k-th enabled point in ascending index order. */
int64_t orc_select_enabled(const orc_cloud *cc, int64_t k)
{
    orc_cloud *c = (orc_cloud *)cc;
    if (!c->dir_valid) build_dir(c);
    int64_t nb = c->nchunks / DIR_BLOCK + 1;
    if (k < 1 || k > c->dir[nb]) return 0;
    int64_t lo = 0, hi = nb; /* find block b with dir[b] < k <= dir[b+1] */
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) / 2;
        if (c->dir[mid] < k) lo = mid; else hi = mid;
    }
    int64_t rem = k - c->dir[lo];
    for (int64_t i = lo * DIR_BLOCK; i < c->nchunks; i++) {
        int pc = __builtin_popcountll(c->enabled[i]);
        if (rem <= pc) {
            uint64_t w = c->enabled[i];
            for (int64_t r = 1; r < rem; r++) w &= w - 1;
            return i * 64 + __builtin_ctzll(w) + 1;
        }
        rem -= pc;
    }
    return 0;
}

/* ------------------------------------------------- small dense lin. alg. */

/* Singular values by one-sided Jacobi (Hestenes) on the columns of an r x c
 * matrix (r >= c), high relative accuracy -- stands in for LAPACK svdvals. */
static void svdvals_cols(double *M, int r, int c, double *sv)
{
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < c - 1; p++)
            for (int q = p + 1; q < c; q++) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < r; i++) {
                    a += M[i * c + p] * M[i * c + p];
                    b += M[i * c + q] * M[i * c + q];
                    g += M[i * c + p] * M[i * c + q];
                }
                if (g == 0.0 || fabs(g) <= 1e-300 + 2.2e-16 * sqrt(a * b)) continue;
                rotated = 1;
                double zeta = (b - a) / (2 * g);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                double cs = 1 / sqrt(1 + t * t), sn = cs * t;
                for (int i = 0; i < r; i++) {
                    double mp = M[i * c + p], mq = M[i * c + q];
                    M[i * c + p] = cs * mp - sn * mq;
                    M[i * c + q] = sn * mp + cs * mq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < c; j++) {
        double a = 0;
        for (int i = 0; i < r; i++) a += M[i * c + j] * M[i * c + j];
        sv[j] = sqrt(a);
    }
}

/* rank(A): LinearAlgebra -- count(svdvals .> min(m,n)*eps*maximum(svdvals)) */
int orc_rank(const double *A, int m, int n)
{
    double M[16], sv[4];
    int r, c;
    if (m >= n) { r = m; c = n; for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[i * c + j] = A[i * n + j]; }
    else { r = n; c = m; for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[j * c + i] = A[i * n + j]; }
    svdvals_cols(M, r, c, sv);
    double smax = 0;
    for (int j = 0; j < c; j++) if (sv[j] > smax) smax = sv[j];
    double tol = (double)(m < n ? m : n) * 2.220446049250313e-16 * smax;
    int cnt = 0;
    for (int j = 0; j < c; j++) if (sv[j] > tol) cnt++;
    return cnt;
}

/* r \ ds for a general square matrix: LU with partial pivoting (LinearAlgebra.lu) */
static int solve3(const double A_in[9], const double b_in[3], double x[3])
{
    double A[9], b[3];
    memcpy(A, A_in, sizeof A);
    memcpy(b, b_in, sizeof b);
    for (int k = 0; k < 3; k++) {
        int piv = k;
        double amax = fabs(A[k * 3 + k]);
        for (int i = k + 1; i < 3; i++)
            if (fabs(A[i * 3 + k]) > amax) { amax = fabs(A[i * 3 + k]); piv = i; }
        if (amax == 0.0) return -1;
        if (piv != k) {
            for (int j = 0; j < 3; j++) { double t = A[k * 3 + j]; A[k * 3 + j] = A[piv * 3 + j]; A[piv * 3 + j] = t; }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < 3; i++) {
            double l = A[i * 3 + k] / A[k * 3 + k];
            A[i * 3 + k] = l;
            for (int j = k + 1; j < 3; j++) A[i * 3 + j] -= l * A[k * 3 + j];
            b[i] -= l * b[k];
        }
    }
    for (int i = 2; i >= 0; i--) {
        double sacc = b[i];
        for (int j = i + 1; j < 3; j++) sacc -= A[i * 3 + j] * x[j];
        x[i] = sacc / A[i * 3 + i];
    }
    return 0;
}

/* ------------------------------------------------------------------ fits */

static void set_v(double *dst, v3 a) { dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; }

/* fit(::Type{FittedPlane}, ...): plane.jl:33-57 */
static int fit_plane(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3) return 0;
    v3 p1 = V(p), p2 = V(p + 3), p3 = V(p + 6);
    v3 crossv = vnormalize(vcross(vsub(p2, p1), vsub(p3, p1)));
    if (vnorm(crossv) < prm->collin_threshold) return 0; /* vacuous, Q9 */
    double thr = prm->cos_alpha[ORC_PLANE];
    int all_ok = 1, all_inv = 1;
    for (int i = 0; i < lp; i++) {
        double dotp = vdot(crossv, vnormalize(V(n + 3 * i)));
        if (!(dotp > thr)) all_ok = 0;
        if (!(dotp < -thr)) all_inv = 0;
    }
    memset(out, 0, sizeof *out);
    out->kind = ORC_PLANE;
    if (all_ok) { set_v(&out->v[0], p1); set_v(&out->v[3], crossv); return 1; }
    if (all_inv) { set_v(&out->v[0], p1); set_v(&out->v[3], vscale(crossv, -1.0)); return 1; }
    return 0;
}

/* fit2pointsphere: sphere.jl:29-75 */
int orc_fit2pointsphere(const double *vv, const double *nn, const orc_params *prm, orc_shape *out)
{
    v3 v1 = V(vv), v2 = V(vv + 3), n1 = V(nn), n2v = V(nn + 3);
    v3 n1n = vnormalize(n1), n2n = vnormalize(n2v);
    v3 center;
    double radius;
    if (fabs(vdot(n1n, n2n)) > prm->cos_parallelthr) {
        center = vdiv(vadd(v1, v2), 2);
        radius = vnorm(vsub(center, v1));
    } else {
        v3 g = vsub(v2, v1);
        v3 h = vcross(n2n, g);
        v3 k = vcross(n2n, n1n);
        double nk = vnorm(k), nh = vnorm(h);
        if (nk < prm->sphere_par || nh < prm->sphere_par) {
            v3 n2 = vcross(n2n, vcross(n1n, n2n));
            v3 n1_ = vcross(n1n, vcross(n2n, n1n));
            /* c1 = v[1] + dot((v[2]-v[1]), n2)/dot(n[1], n2) * n[1]  (raw normals, Q10) */
            v3 c1 = vadd(v1, vscale(n1, vdot(vsub(v2, v1), n2) / vdot(n1, n2)));
            v3 c2 = vadd(v2, vscale(n2v, vdot(vsub(v1, v2), n1_) / vdot(n2v, n1_)));
            center = vdiv(vadd(c1, c2), 2);
            radius = (vnorm(vsub(v1, center)) + vnorm(vsub(v1, center))) / 2;
        } else if (vdot(h, k) > 0) {
            center = vadd(v1, vscale(n1n, nh / nk));
            radius = vnorm(vsub(center, v1));
        } else {
            center = vsub(v1, vscale(n1n, nh / nk));
            radius = vnorm(vsub(center, v1));
        }
    }
    memset(out, 0, sizeof *out);
    out->kind = ORC_SPHERE;
    out->outwards = 0;
    set_v(&out->v[0], center);
    out->v[3] = radius;
    return 1;
}

/* fit(::Type{FittedSphere}, ...): sphere.jl:87-114 */
static int fit_sphere(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3) return 0;
    orc_shape sp;
    if (!orc_fit2pointsphere(p, n, prm, &sp)) return 0;
    v3 center = V(&sp.v[0]);
    double radius = sp.v[3], thr = prm->cos_alpha[ORC_SPHERE], eps = prm->eps[ORC_SPHERE];
    int vert = 1, ok = 1, inv = 1;
    for (int i = 0; i < lp; i++) {
        v3 pi = V(p + 3 * i);
        if (!(fabs(vnorm(vsub(pi, center)) - radius) < eps)) vert = 0;
        double dotp = vdot(vnormalize(vsub(pi, center)), vnormalize(V(n + 3 * i)));
        if (!(dotp > thr)) ok = 0;
        if (!(dotp < -thr)) inv = 0;
    }
    if (!vert) return 0;
    *out = sp;
    if (ok) { out->outwards = 1; return 1; }
    if (inv) { out->outwards = 0; return 1; }
    return 0;
}

/* helpers local to fit2pointcylinder: cylinder.jl:46-101 */
static v3 cyl_project2plane(v3 n, v3 w)
{
    /* w + n*lineplaneintersect(n,n,w); lineplaneintersect = dot(-n, w)/dot(n,u) */
    return vadd(w, vscale(n, vdot(vneg(n), w) / vdot(n, n)));
}

static void cyl_projectto2d(v3 xa, v3 ya, v3 za, v3 p1, double r[2])
{
    double xx = xa.x, xy = xa.y, xz = xa.z;
    double yx = ya.x, yy = ya.y, yz = ya.z;
    double zx = za.x, zy = za.y, zz = za.z;
    double px = p1.x, py = p1.y, pz = p1.z;
    /* cylinder.jl:80-81, operation order preserved */
    r[0] = -((-(pz * yy * zx) + py * yz * zx + pz * yx * zy - px * yz * zy - py * yx * zz + px * yy * zz) /
             (xz * yy * zx - xy * yz * zx - xz * yx * zy + xx * yz * zy + xy * yx * zz - xx * yy * zz));
    r[1] = -((pz * xy * zx - py * xz * zx - pz * xx * zy + px * xz * zy + py * xx * zz - px * xy * zz) /
             (xz * yy * zx - xy * yz * zx - xz * yx * zy + xx * yz * zy + xy * yx * zz - xx * yy * zz));
}

/* fit2pointcylinder: cylinder.jl:34-125 */
int orc_fit2pointcylinder(const double *p, const double *n, const orc_params *prm, orc_shape *out)
{
    v3 p1 = V(p), p2 = V(p + 3), n1 = V(n), n2 = V(n + 3);
    if (fabs(vdot(n1, n2)) > prm->cos_parallelthr) return 0; /* raw normals, Q18 */
    v3 an = vnormalize(vcross(n1, n2));
    v3 xax = vnormalize(cyl_project2plane(an, p1));
    v3 yax = vnormalize(vcross(an, xax));
    double p11[2], p12[2], p21[2], p22[2];
    cyl_projectto2d(xax, yax, an, cyl_project2plane(an, p1), p11);
    cyl_projectto2d(xax, yax, an, cyl_project2plane(an, vadd(p1, n1)), p12);
    cyl_projectto2d(xax, yax, an, cyl_project2plane(an, p2), p21);
    cyl_projectto2d(xax, yax, an, cyl_project2plane(an, vadd(p2, n2)), p22);
    /* lineintersectionpoint([p11,p12],[p21,p22]): cylinder.jl:87-101; det 2x2 = a1*b2 - a2*b1 */
    double amb[2] = { p11[0] - p12[0], p11[1] - p12[1] };
    double cmd[2] = { p21[0] - p22[0], p21[1] - p22[1] };
    double d1 = p11[0] * p12[1] - p11[1] * p12[0];
    double d2 = p21[0] * p22[1] - p21[1] * p22[0];
    double d3 = amb[0] * cmd[1] - amb[1] * cmd[0];
    double interc[2] = { (d1 * cmd[0] - d2 * amb[0]) / d3, (d1 * cmd[1] - d2 * amb[1]) / d3 };
    v3 c = vadd(vscale(xax, interc[0]), vscale(yax, interc[1]));
    /* nnormies = [norm(pt - c - an*dot(an, pt-c)) for pt in p[1:2]] */
    double nn1 = vnorm(vsub(vsub(p1, c), vscale(an, vdot(an, vsub(p1, c)))));
    double nn2 = vnorm(vsub(vsub(p2, c), vscale(an, vdot(an, vsub(p2, c)))));
    double R = (nn1 + nn2) / 2;
    /* outw = dot(p12proj-p11proj, p11proj-interc) > 0 */
    double outw = (p12[0] - p11[0]) * (p11[0] - interc[0]) + (p12[1] - p11[1]) * (p11[1] - interc[1]);
    memset(out, 0, sizeof *out);
    out->kind = ORC_CYLINDER;
    out->outwards = outw > 0;
    set_v(&out->v[0], an);
    set_v(&out->v[3], c);
    out->v[6] = R;
    return 1;
}

/* fit(::Type{FittedCylinder}, ...): cylinder.jl:135-168 */
static int fit_cylinder(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3) return 0;
    orc_shape fc;
    if (!orc_fit2pointcylinder(p, n, prm, &fc)) return 0;
    v3 axis = V(&fc.v[0]), center = V(&fc.v[3]);
    double radius = fc.v[6], thr = prm->cos_alpha[ORC_CYLINDER], eps = prm->eps[ORC_CYLINDER];
    int vert = 1, ok = 1, inv = 1;
    for (int i = 0; i < lp; i++) {
        v3 pi = V(p + 3 * i);
        v3 curr_norm = vsub(vsub(pi, vscale(axis, vdot(axis, vsub(pi, center)))), center);
        if (!(fabs(vnorm(curr_norm) - radius) < eps)) vert = 0;
        double dotp = vdot(vnormalize(curr_norm), V(n + 3 * i));
        if (!(dotp > thr)) ok = 0;
        if (!(dotp < -thr)) inv = 0;
    }
    if (!vert) return 0;
    *out = fc;
    if (ok) { out->outwards = 1; return 1; }
    if (inv) { out->outwards = 0; return 1; }
    return 0;
}

static double clamp1(double x) { return x < -1 ? -1 : (x > 1 ? 1 : x); }

/* fit3pointcone: cone.jl:39-61 */
int orc_fit3pointcone(const double *p, const double *n, orc_shape *out)
{
    double r[9], rv[12], ds[3], apx[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = n[3 * i + j];
    if (orc_rank(r, 3, 3) != 3) return 0;
    for (int i = 0; i < 3; i++) ds[i] = vdot(V(p + 3 * i), V(n + 3 * i));
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) rv[i * 4 + j] = r[i * 3 + j];
        rv[i * 4 + 3] = -1 * ds[i];
    }
    if (orc_rank(rv, 3, 4) != 3) return 0;
    if (solve3(r, ds, apx)) return 0;
    v3 ap = V(apx);
    v3 a3p[3];
    for (int i = 0; i < 3; i++) {
        v3 d = vsub(V(p + 3 * i), ap);
        a3p[i] = vadd(ap, vdiv(d, vnorm(d)));
    }
    v3 ax = vnormalize(vcross(vsub(a3p[1], a3p[0]), vsub(a3p[2], a3p[0])));
    v3 midp = vdiv(vadd(vadd(a3p[0], a3p[1]), a3p[2]), 3);
    v3 dirv = vnormalize(vsub(midp, ap));
    if (vdot(ax, dirv) < 0) ax = vscale(ax, -1.0);
    double angles[3];
    for (int i = 0; i < 3; i++) angles[i] = ORC_ACOS(clamp1(vdot(vnormalize(vsub(V(p + 3 * i), ap)), ax)));
    double opangle = 2 * ((angles[0] + angles[1]) + angles[2]) / 3;
    memset(out, 0, sizeof *out);
    out->kind = ORC_CONE;
    out->outwards = 1;
    set_v(&out->v[0], ap);
    set_v(&out->v[3], ax);
    out->v[6] = opangle;
    orc_shape_finalize(out);
    return 1;
}

/* validatecone: cone.jl:87-115; fit(::Type{FittedCone}): cone.jl:123-128 */
static int fit_cone(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3 || lp > 16) return 0;
    orc_shape cone;
    if (!orc_fit3pointcone(p, n, &cone)) return 0;
    double dist[16];
    v3 cn[16];
    for (int i = 0; i < lp; i++) project2cone(&cone, V(p + 3 * i), &dist[i], &cn[i]);
    for (int i = 0; i < lp; i++)
        if (dist[i] > prm->eps[ORC_CONE]) return 0; /* no abs: Q11, cone.jl:93 */
    if (cone.v[6] < prm->minconeopang) return 0;
    double thr = prm->cos_alpha[ORC_CONE];
    int ok = 1, inv = 1;
    for (int i = 0; i < lp; i++) {
        double dotp = vdot(cn[i], V(n + 3 * i));
        if (!(dotp > thr)) ok = 0;
        if (!(dotp < -thr)) inv = 0;
    }
    *out = cone;
    if (ok) { out->outwards = 1; return 1; }
    if (inv) { out->outwards = 0; return 1; }
    return 0;
}

int orc_fit(int kind, const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    switch (kind) {
    case ORC_PLANE: return fit_plane(p, n, lp, prm, out);
    case ORC_SPHERE: return fit_sphere(p, n, lp, prm, out);
    case ORC_CYLINDER: return fit_cylinder(p, n, lp, prm, out);
    case ORC_CONE: return fit_cone(p, n, lp, prm, out);
    }
    return 0;
}

/* ---------------------------------------------------------------- octree */

/* findAABB: utilities.jl:125-136 */
void orc_findAABB(const double *pts, int64_t n, int dim, double *minv, double *maxv)
{
    for (int j = 0; j < dim; j++) { minv[j] = pts[j]; maxv[j] = pts[j]; }
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < dim; j++) {
            double a = pts[i * dim + j];
            minv[j] = minv[j] > a ? a : minv[j];
            maxv[j] = maxv[j] < a ? a : maxv[j];
        }
}

/* iswithinrectangle: octree.jl:187-196; vs[1,1,1]=origin, vs[2,2,2]=origin+widths */
int orc_iswithinrectangle(const double origin[3], const double widths[3], const double p[3])
{
    for (int i = 0; i < 3; i++) {
        double vmin = origin[i], vmax = origin[i] + widths[i];
        if (!(vmin < p[i])) return 0;
        if (!(vmax >= p[i])) return 0;
    }
    return 1;
}

typedef struct {
    double origin[3], widths[3], div[3];
    int32_t depth, parent, child[8]; /* child[(i)+(2j)+(4k)], -1 = leaf */
    int64_t npts;
    int64_t *pts; /* 1-based */
} onode;

struct orc_octree {
    onode *nodes;
    int32_t n_nodes, cap;
    int overflow; /* refinement depth guard hit (Q17) */
};

static int32_t new_node(orc_octree *t)
{
    if (t->n_nodes == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 64;
        t->nodes = (onode *)realloc(t->nodes, sizeof(onode) * (size_t)t->cap);
    }
    onode *nd = &t->nodes[t->n_nodes];
    memset(nd, 0, sizeof *nd);
    for (int i = 0; i < 8; i++) nd->child[i] = -1;
    return t->n_nodes++;
}

/* buildoctree: octree.jl:237-244; adaptivesampling!/split!/child_boundary from
 * RegionTrees ^0.3 [recalled]: divisions = origin + widths/2; child origin is the
 * cell origin or the division, child width is division-origin or origin+width-division.
 * Root is Cell(minV, maxV): second argument is WIDTHS (Q2). */
orc_octree *orc_octree_build(const double *xyz, int64_t n)
{
    orc_octree *t = (orc_octree *)calloc(1, sizeof *t);
    double minv[3], maxv[3];
    orc_findAABB(xyz, n, 3, minv, maxv);
    int32_t root = new_node(t);
    for (int i = 0; i < 3; i++) { t->nodes[root].origin[i] = minv[i]; t->nodes[root].widths[i] = maxv[i]; }
    t->nodes[root].depth = 1;
    t->nodes[root].parent = -1;
    t->nodes[root].npts = n;
    t->nodes[root].pts = (int64_t *)malloc(8 * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) t->nodes[root].pts[i] = i + 1;
    for (int32_t cur = 0; cur < t->n_nodes; cur++) { /* queue order is irrelevant to the result */
        if (!(t->nodes[cur].npts > 8)) continue;      /* needs_refinement: octree.jl:163-165 */
        if (t->nodes[cur].depth >= 48) { t->overflow = 1; continue; }
        for (int i = 0; i < 3; i++) t->nodes[cur].div[i] = t->nodes[cur].origin[i] + t->nodes[cur].widths[i] / 2;
        for (int ci = 0; ci < 8; ci++) {
            int32_t ch = new_node(t);
            onode *par = &t->nodes[cur], *nd = &t->nodes[ch];
            int idx[3] = { ci & 1, (ci >> 1) & 1, (ci >> 2) & 1 };
            for (int i = 0; i < 3; i++) {
                nd->origin[i] = idx[i] == 0 ? par->origin[i] : par->div[i];
                nd->widths[i] = idx[i] == 0 ? par->div[i] - par->origin[i]
                                            : par->origin[i] + par->widths[i] - par->div[i];
            }
            nd->depth = par->depth + 1;
            nd->parent = cur;
            nd->pts = (int64_t *)malloc(8 * (size_t)(par->npts ? par->npts : 1));
            /* refine_data: octree.jl:167-177 -- re-test all parent points */
            for (int64_t k = 0; k < par->npts; k++) {
                int64_t id = par->pts[k];
                if (orc_iswithinrectangle(nd->origin, nd->widths, &xyz[3 * (id - 1)])) nd->pts[nd->npts++] = id;
            }
            par->child[ci] = ch;
        }
    }
    return t;
}

/* the same build on a Float32 cloud (octree.jl:102-109 converts the vertices; findAABB, RegionTrees' divisions and child
 * boundaries and iswithinrectangle then run on Float32 values): every operation below rounds to binary32; the nodes keep
 * the results widened to double */
orc_octree *orc_octree_build_f32(const float *xyz, int64_t n)
{
    orc_octree *t = (orc_octree *)calloc(1, sizeof *t);
    float minv[3] = { 0, 0, 0 }, maxv[3] = { 0, 0, 0 };
    for (int j = 0; j < 3 && n > 0; j++) minv[j] = maxv[j] = xyz[j];          /* findAABB: utilities.jl:125-136 */
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
            const float a = xyz[3 * i + j];
            minv[j] = minv[j] > a ? a : minv[j];
            maxv[j] = maxv[j] < a ? a : maxv[j];
        }
    int32_t root = new_node(t);
    for (int i = 0; i < 3; i++) { t->nodes[root].origin[i] = minv[i]; t->nodes[root].widths[i] = maxv[i]; }
    t->nodes[root].depth = 1;
    t->nodes[root].parent = -1;
    t->nodes[root].npts = n;
    t->nodes[root].pts = (int64_t *)malloc(8 * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) t->nodes[root].pts[i] = i + 1;
    for (int32_t cur = 0; cur < t->n_nodes; cur++) {
        if (!(t->nodes[cur].npts > 8)) continue;
        if (t->nodes[cur].depth >= 48) { t->overflow = 1; continue; }
        for (int i = 0; i < 3; i++) {
            const float o = (float)t->nodes[cur].origin[i], w = (float)t->nodes[cur].widths[i];
            const float half = w / 2.0f;
            t->nodes[cur].div[i] = (float)(o + half);
        }
        for (int ci = 0; ci < 8; ci++) {
            int32_t ch = new_node(t);
            onode *par = &t->nodes[cur], *nd = &t->nodes[ch];
            int idx[3] = { ci & 1, (ci >> 1) & 1, (ci >> 2) & 1 };
            for (int i = 0; i < 3; i++) {
                const float o = (float)par->origin[i], w = (float)par->widths[i], d = (float)par->div[i];
                const float top = o + w;
                nd->origin[i] = idx[i] == 0 ? o : d;
                nd->widths[i] = idx[i] == 0 ? (float)(d - o) : (float)(top - d);
            }
            nd->depth = par->depth + 1;
            nd->parent = cur;
            nd->pts = (int64_t *)malloc(8 * (size_t)(par->npts ? par->npts : 1));
            for (int64_t k = 0; k < par->npts; k++) {
                const int64_t id = par->pts[k];
                const float *p = &xyz[3 * (id - 1)];
                int in = 1;
                for (int i = 0; i < 3 && in; i++) {           /* iswithinrectangle: octree.jl:187-196, Float32 */
                    const float vmin = (float)nd->origin[i];
                    const float vmax = (float)nd->origin[i] + (float)nd->widths[i];
                    if (!(vmin < p[i]) || !(vmax >= p[i])) in = 0;
                }
                if (in) nd->pts[nd->npts++] = id;
            }
            par->child[ci] = ch;
        }
    }
    return t;
}

void orc_octree_destroy(orc_octree *t)
{
    if (!t) return;
    for (int32_t i = 0; i < t->n_nodes; i++) free(t->nodes[i].pts);
    free(t->nodes);
    free(t);
}

/* octreedepth: octree.jl:221-230 */
int orc_octree_depth(const orc_octree *t)
{
    int d = t->nodes[0].depth;
    for (int32_t i = 0; i < t->n_nodes; i++)
        if (t->nodes[i].child[0] < 0 && t->nodes[i].depth > d) d = t->nodes[i].depth;
    return d;
}

/* findleaf (RegionTrees): child index per axis = point[i] >= divisions[i] ? 2 : 1 */
int orc_octree_findleaf(const orc_octree *t, const double p[3], int32_t *path, int cap)
{
    int32_t cur = 0;
    int d = 0;
    for (;;) {
        if (d < cap) path[d] = cur;
        d++;
        const onode *nd = &t->nodes[cur];
        if (nd->child[0] < 0) return d;
        int ci = (p[0] >= nd->div[0] ? 1 : 0) | (p[1] >= nd->div[1] ? 2 : 0) | (p[2] >= nd->div[2] ? 4 : 0);
        cur = nd->child[ci];
    }
}

int64_t orc_octree_node_npoints(const orc_octree *t, int32_t node) { return t->nodes[node].npts; }
const int64_t *orc_octree_node_points(const orc_octree *t, int32_t node) { return t->nodes[node].pts; }

/* ------------------------------------------------------------------- RNG */
static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void orc_rng_seed(orc_rng *r, uint64_t seed)
{
    memset(r, 0, sizeof *r);
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

uint64_t orc_rng_next(orc_rng *r)
{
    r->draws++;
    if (r->stream && r->stream_pos < r->stream_len) return r->stream[r->stream_pos++];
    uint64_t *s = r->s;
    uint64_t result = rotl(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return result;
}

int64_t orc_rng_range(orc_rng *r, int64_t n)
{
    return 1 + (int64_t)(((unsigned __int128)orc_rng_next(r) * (unsigned __int128)(uint64_t)n) >> 64);
}

/* sampling_streams = 1: one splitmix64 stream per (iteration k, minimal set j) */
static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef struct { orc_rng *seq; uint64_t x; int per_set; } draw_src;

static int64_t draw_range(draw_src *d, int64_t n)
{
    if (!d->per_set) return orc_rng_range(d->seq, n);
    d->seq->draws++;
    d->x += 0x9E3779B97F4A7C15ULL;
    return 1 + (int64_t)(((unsigned __int128)mix64(d->x) * (unsigned __int128)(uint64_t)n) >> 64);
}

/* ---------------------------------------------------------------- driver */

/* ---- fixed-behaviour sampling: level-weighted cells of a linear (Morton) octree ------------
 * What docs/src/ransac.md:73-96 describes and the live reference never does: the
 * first point is uniform over the enabled points; a level l is drawn from the level distribution
 * P; the other points are drawn uniformly from the enabled points of the level-l cell that holds
 * the first point.  Cells are cubes of the cloud's bounding cube; with the points sorted by
 * Morton code a cell is a contiguous range, so "k-th enabled point of the cell" is a rank/select
 * on the Morton-ordered enabled bits.  This is OUR specification (nothing in the reference runs
 * it); the product must match it bit for bit. */
typedef struct {
    int64_t n, nwords;
    int depth;
    uint64_t *code;   /* sorted Morton codes */
    int32_t *perm;    /* Morton position -> original index0 */
    int32_t *pos;     /* original index0 -> Morton position */
    uint64_t *men;    /* enabled bits in Morton order */
    int32_t *prefix;  /* exclusive popcount prefix per word of men (+ total at [nwords]) */
} lin_octree;

static uint64_t spread21(uint64_t v)
{
    v &= 0x1FFFFFULL;
    v = (v | (v << 32)) & 0x1F00000000FFFFULL;
    v = (v | (v << 16)) & 0x1F0000FF0000FFULL;
    v = (v | (v << 8)) & 0x100F00F00F00F00FULL;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}

typedef struct { uint64_t code; int32_t idx; } code_idx;
static int cmp_code_idx(const void *a, const void *b)
{
    const code_idx *x = (const code_idx *)a, *y = (const code_idx *)b;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

static void lo_rebuild_prefix(lin_octree *o)
{
    int32_t acc = 0;
    for (int64_t w = 0; w < o->nwords; w++) { o->prefix[w] = acc; acc += __builtin_popcountll(o->men[w]); }
    o->prefix[o->nwords] = acc;
}

static lin_octree *lo_build(const orc_cloud *c, int max_depth)
{
    lin_octree *o = (lin_octree *)calloc(1, sizeof *o);
    int64_t n = c->n;
    o->n = n;
    o->nwords = (n + 63) / 64;
    o->code = (uint64_t *)malloc(8 * (size_t)(n ? n : 1));
    o->perm = (int32_t *)malloc(4 * (size_t)(n ? n : 1));
    o->pos = (int32_t *)malloc(4 * (size_t)(n ? n : 1));
    o->men = (uint64_t *)calloc((size_t)(o->nwords ? o->nwords : 1), 8);
    o->prefix = (int32_t *)calloc((size_t)(o->nwords + 1), 4);
    double lo[3], hi[3];
    orc_findAABB(c->xyz, n, 3, lo, hi);
    double size = 0;
    for (int k = 0; k < 3; k++) if (hi[k] - lo[k] > size) size = hi[k] - lo[k];
    size = size * (1 + 1e-9);
    if (!(size > 0)) size = 1;
    code_idx *ci = (code_idx *)malloc(sizeof(code_idx) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        uint64_t code = 0;
        for (int k = 0; k < 3; k++) {
            double t = (c->xyz[3 * i + k] - lo[k]) / size;
            double q = t * 2097152.0;
            uint64_t qi = !(q >= 0) ? 0 : (q >= 2097151.0 ? 2097151ULL : (uint64_t)q);
            code |= spread21(qi) << k;
        }
        ci[i].code = code;
        ci[i].idx = (int32_t)i;
    }
    qsort(ci, (size_t)n, sizeof(code_idx), cmp_code_idx);
    for (int64_t i = 0; i < n; i++) { o->code[i] = ci[i].code; o->perm[i] = ci[i].idx; o->pos[ci[i].idx] = (int32_t)i; }
    free(ci);
    /* depth: first level whose fullest cell holds <= 8 points (octree.jl:163-165), capped */
    if (max_depth < 1) max_depth = 1;
    if (max_depth > 21) max_depth = 21;
    o->depth = max_depth;
    for (int l = 1; l <= max_depth; l++) {
        int shift = 3 * (21 - (l - 1));
        int64_t run = 0, best = 0;
        uint64_t prev = 0;
        for (int64_t i = 0; i < n; i++) {
            uint64_t key = shift >= 63 ? 0 : (o->code[i] >> shift);
            if (i == 0 || key != prev) { run = 0; prev = key; }
            if (++run > best) best = run;
        }
        if (best <= 8) { o->depth = l; break; }
    }
    for (int64_t i = 0; i < n; i++)
        if (is_enabled(c, i)) o->men[o->pos[i] >> 6] |= 1ULL << (o->pos[i] & 63);
    lo_rebuild_prefix(o);
    return o;
}

static void lo_free(lin_octree *o)
{
    if (!o) return;
    free(o->code); free(o->perm); free(o->pos); free(o->men); free(o->prefix); free(o);
}

static int64_t lo_lower_bound(const lin_octree *o, uint64_t key)
{
    int64_t lo = 0, hi = o->n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (o->code[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static int64_t lo_rank(const lin_octree *o, int64_t i) /* enabled among Morton positions < i */
{
    if (i >= o->n) return o->prefix[o->nwords];
    return o->prefix[i >> 6] + __builtin_popcountll(o->men[i >> 6] & ((1ULL << (i & 63)) - 1ULL));
}

static int64_t lo_select(const lin_octree *o, int64_t r) /* Morton position of the r-th (1-based) enabled */
{
    int64_t lo = 0, hi = o->nwords;
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (o->prefix[mid] < r) lo = mid; else hi = mid;
    }
    uint64_t m = o->men[lo];
    for (int64_t t = 1; t < r - o->prefix[lo]; t++) m &= m - 1;
    return lo * 64 + __builtin_ctzll(m);
}

static uint64_t draw_raw(draw_src *d)
{
    if (!d->per_set) return orc_rng_next(d->seq);
    d->seq->draws++;
    d->x += 0x9E3779B97F4A7C15ULL;
    return mix64(d->x);
}

/* one minimal set, level-weighted; returns 1 and the level, or 0 */
static int sample_octree(const orc_cloud *c, const lin_octree *o, const double *P, const orc_params *p,
                         draw_src *rng, int64_t n_enabled, int64_t *sd, int *level_out)
{
    if (n_enabled <= 0) return 0;
    int64_t r1 = draw_range(rng, c->n);
    while (!is_enabled(c, r1 - 1)) r1 = draw_range(rng, c->n);
    double u = (double)(draw_raw(rng) >> 11) * (1.0 / 9007199254740992.0);
    int level = o->depth;
    double acc = 0;
    for (int l = 0; l < o->depth; l++) {
        acc += P[l];
        if (u < acc) { level = l + 1; break; }
    }
    *level_out = level;
    int shift = 3 * (21 - (level - 1));
    int64_t lo, hi;
    if (shift >= 63) { lo = 0; hi = o->n; }
    else {
        uint64_t key = o->code[o->pos[r1 - 1]] >> shift;
        lo = lo_lower_bound(o, key << shift);
        hi = (key + 1) << shift == 0 ? o->n : lo_lower_bound(o, (key + 1) << shift);
        if (((key + 1) << shift) >> shift != key + 1) hi = o->n; /* top cell: no overflow */
    }
    int64_t base = lo_rank(o, lo), ne = lo_rank(o, hi) - base;
    if (ne < p->drawN) return 0;
    sd[0] = r1;
    for (int q = 1; q < p->drawN; q++) {
        int64_t pick = (int64_t)o->perm[lo_select(o, base + draw_range(rng, ne))] + 1;
        if (pick == sd[0]) pick = (int64_t)o->perm[lo_select(o, base + draw_range(rng, ne))] + 1;
        sd[q] = pick;
    }
    for (int i = 1; i < p->drawN; i++)
        for (int j = 0; j < i; j++)
            if (sd[i] == sd[j]) return 0;
    return 1;
}

/* level distribution update (octree.jl:198-205 with the intended initialisation); kept unchanged
 * while no score has been collected or if the formula leaves the simplex */
static void update_level_probs(double *P, const double *sigma, int d)
{
    double w = 0, Pn[32];
    for (int i = 0; i < d; i++) w += sigma[i] / P[i];
    if (!(w > 0) || d > 32) return;
    double sum = 0;
    for (int i = 0; i < d; i++) {
        Pn[i] = 0.9 * sigma[i] / (w * P[i]) + (1 - 0.9) / d;
        if (!(Pn[i] >= 0)) return;
        sum += Pn[i];
    }
    if (!(sum > 0)) return;
    for (int i = 0; i < d; i++) P[i] = Pn[i];
}

/* Julia argmax over a Float64 vector: NaN is the maximum; first occurrence wins */
static int jl_argmax(const double *a, int n)
{
    int best = 0;
    for (int i = 0; i < n; i++) if (a[i] != a[i]) return i;
    for (int i = 1; i < n; i++) if (a[i] > a[best]) best = i;
    return best;
}

/* samplepointcloud4!: fitting.jl:383-430 with the root cell (the
 * argmax at :401 is always 1, asserted by the caller).  sd: 1-based indices. */
static int sample4(const orc_cloud *c, const orc_params *p, draw_src *rng, int64_t n_enabled, int64_t *sd)
{
    int64_t r1 = draw_range(rng, c->n);
    while (!is_enabled(c, r1 - 1)) r1 = draw_range(rng, c->n);
    if (n_enabled < p->drawN) return 0; /* (false, 0) */
    sd[0] = r1;
    for (int k = 1; k < p->drawN; k++) {
        int64_t nexti = draw_range(rng, n_enabled);
        int64_t cand = orc_select_enabled(c, nexti);
        if (sd[0] == cand) {
            nexti = draw_range(rng, n_enabled); /* try oncemore: fitting.jl:416-419 */
            cand = orc_select_enabled(c, nexti);
        }
        sd[k] = cand;
    }
    for (int i = 1; i < p->drawN; i++) /* allisdifferent: utilities.jl:285-295 */
        for (int j = 0; j < i; j++)
            if (sd[i] == sd[j]) return 0; /* (false, 1) */
    return 1;
}

typedef struct {
    orc_shape *shapes;
    orc_ci *scores;
    uint64_t **masks; /* inpoints as a bitset over subset positions */
    int64_t n, cap;
} cand_store;

static void store_push(cand_store *st, const orc_shape *s, orc_ci ci, uint64_t *mask)
{
    if (st->n == st->cap) {
        st->cap = st->cap ? st->cap * 2 : 256;
        st->shapes = (orc_shape *)realloc(st->shapes, sizeof(orc_shape) * (size_t)st->cap);
        st->scores = (orc_ci *)realloc(st->scores, sizeof(orc_ci) * (size_t)st->cap);
        st->masks = (uint64_t **)realloc(st->masks, sizeof(uint64_t *) * (size_t)st->cap);
    }
    st->shapes[st->n] = *s;
    st->scores[st->n] = ci;
    st->masks[st->n] = mask;
    st->n++;
}

static void store_delete(cand_store *st, int64_t i) /* deleteat!: fitting.jl:126-131 */
{
    free(st->masks[i]);
    for (int64_t k = i; k + 1 < st->n; k++) {
        st->shapes[k] = st->shapes[k + 1];
        st->scores[k] = st->scores[k + 1];
        st->masks[k] = st->masks[k + 1];
    }
    st->n--;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ransac(pc, params): iterations.jl:35-162 */
int orc_ransac(orc_cloud *c, const double *xyz, const double *nrm, const orc_params *p,
               orc_rng *rng, int octree_depth, orc_result *out)
{
    (void)xyz; (void)nrm; /* the cloud holds its own copy */
    double t0 = now_s();
    memset(out, 0, sizeof *out);
    if (octree_depth < 1) octree_depth = 1;
    if (p->drawN < 2 || p->drawN > 16) return -1;
    /* nomodRANSACCloud passes (levelscore, levelweight) into (levelweight, levelscore): octree.jl:82-84 */
    double *levelweight = (double *)calloc((size_t)octree_depth, 8);
    double *levelscore = (double *)malloc(8 * (size_t)octree_depth);
    for (int i = 0; i < octree_depth; i++) levelscore[i] = 1.0 / octree_depth;

    int64_t w = (c->s + 63) / 64;
    cand_store st = { 0 };
    orc_shape *cands = NULL;
    int32_t *levels = NULL;
    int64_t ncand = 0, capcand = 0;
    int64_t countcandidates[4] = { 0, 0, 0, 0 };
    int64_t ext_cap = 0;
    int64_t *sd = (int64_t *)malloc(8 * (size_t)p->drawN);
    double *fp = (double *)malloc(8 * 3 * (size_t)p->drawN), *fn = (double *)malloc(8 * 3 * (size_t)p->drawN);
    int64_t *refit_idx = (int64_t *)malloc(8 * (size_t)(c->n ? c->n : 1));
    int rc = 0;

    lin_octree *oct = NULL;
    double octP[32], octS[32];
    if (p->octree_sampling) {
        oct = lo_build(c, p->octree_max_depth);
        for (int i = 0; i < oct->depth; i++) { octP[i] = 1.0 / oct->depth; octS[i] = 0; }
    }

    int64_t k;
    for (k = 1; k <= p->itermax; k++) {
        int64_t n_enabled = orc_cloud_count_enabled(c);
        if (n_enabled < p->tau) break; /* iterations.jl:75 */
        for (int i = 0; i < p->minsubsetN; i++) {
            draw_src src = { rng, 0, p->sampling_streams };
            if (p->sampling_streams)
                src.x = mix64(rng->s[0] + (uint64_t)k * 0xD1B54A32D192ED03ULL) ^
                        mix64((uint64_t)i * 0x8CB92BA72F3D8DD7ULL + 0x2545F4914F6CDD1DULL);
            int lvl;
            if (oct) {
                if (!sample_octree(c, oct, octP, p, &src, n_enabled, sd, &lvl)) continue;
            } else {
                if (!sample4(c, p, &src, n_enabled, sd)) continue;
                /* fitting.jl:401: argmax(levelweight[1:max_depth]) must be 1 */
                lvl = jl_argmax(levelweight, octree_depth) + 1;
                if (lvl != 1) { rc = -2; goto done; }
            }
            for (int q = 0; q < p->drawN; q++) {
                memcpy(fp + 3 * q, &c->xyz[3 * (sd[q] - 1)], 24);
                memcpy(fn + 3 * q, &c->nrm[3 * (sd[q] - 1)], 24);
            }
            for (int t = 0; t < p->n_shape_types; t++) { /* forcefitshapes!: fitting.jl:165-173 */
                orc_shape fitted;
                if (!(c->f32 ? orc32_fit(p->shape_types[t], fp, fn, p->drawN, p, &fitted)
                             : orc_fit(p->shape_types[t], fp, fn, p->drawN, p, &fitted))) continue;
                if (ncand == capcand) {
                    capcand = capcand ? capcand * 2 : 64;
                    cands = (orc_shape *)realloc(cands, sizeof(orc_shape) * (size_t)capcand);
                    levels = (int32_t *)realloc(levels, 4 * (size_t)capcand);
                }
                cands[ncand] = fitted;
                levels[ncand] = lvl;
                ncand++;
            }
        }
        countcandidates[2] += ncand;
        /* scorecandidates!: fitting.jl:181-190 */
        for (int64_t i = 0; i < ncand; i++) {
            uint64_t *mask = (uint64_t *)malloc(8 * (size_t)(w ? w : 1));
            int64_t cnt = orc_scorecandidate(c, &cands[i], p, NULL, mask);
            orc_ci sc = orc_estimatescore(c->s, c->n, cnt, p->score_mode);
            if (oct) octS[levels[i] - 1] += sc.E;
            else levelscore[levels[i] - 1] += sc.E;
            store_push(&st, &cands[i], sc, mask);
        }
        ncand = 0;
        countcandidates[3] = k * p->minsubsetN;
        countcandidates[1] = st.n;
        if (st.n >= 1) {
            /* findhighestscore: fitting.jl:140-158 (overlap flag unused by ransac) */
            int64_t ind = 0;
            double highest = st.scores[0].E;
            for (int64_t i = 0; i < st.n; i++)
                if (st.scores[i].E > highest) { highest = st.scores[i].E; ind = i; }
            double scr = st.scores[ind].E;
            int64_t sN = countcandidates[p->extract_s];
            double ppp = orc_prob(scr, sN, c->n, p->drawN);
            if (g_trace_buf != NULL) {
                if (g_trace_n < g_trace_cap) {
                    double *q = g_trace_buf + 4 * g_trace_n;
                    q[0] = (double)k; q[1] = scr; q[2] = (double)sN; q[3] = ppp;
                }
                g_trace_n++;
            }
            if (ppp > p->prob_det) {
                orc_shape bestshape = st.shapes[ind];
                int64_t ne = orc_refit(c, &bestshape, p, refit_idx, c->n);
                orc_invalidate(c, refit_idx, ne);
                if (oct) {
                    for (int64_t q = 0; q < ne; q++) {
                        int32_t mp = oct->pos[refit_idx[q] - 1];
                        oct->men[mp >> 6] &= ~(1ULL << (mp & 63));
                    }
                    lo_rebuild_prefix(oct);
                }
                if (out->n_shapes == ext_cap) {
                    ext_cap = ext_cap ? ext_cap * 2 : 16;
                    out->shapes = (orc_extracted *)realloc(out->shapes, sizeof(orc_extracted) * (size_t)ext_cap);
                }
                orc_extracted *e = &out->shapes[out->n_shapes++];
                e->shape = bestshape;
                e->n_inpoints = ne;
                e->inpoints = (int64_t *)malloc(8 * (size_t)(ne ? ne : 1));
                memcpy(e->inpoints, refit_idx, 8 * (size_t)ne);
                e->score_E = scr;
                e->iteration = k;
                store_delete(&st, ind);
                /* removeinvalidshapes!: fitting.jl:209-221 */
                int64_t keep = 0;
                for (int64_t i = 0; i < st.n; i++) {
                    int invalid = 0;
                    for (int64_t wd = 0; wd < w && !invalid; wd++) {
                        uint64_t m = st.masks[i][wd];
                        while (m) {
                            int b = __builtin_ctzll(m);
                            m &= m - 1;
                            if (!is_enabled(c, c->subset1[wd * 64 + b] - 1)) { invalid = 1; break; }
                        }
                    }
                    if (invalid) { free(st.masks[i]); continue; }
                    st.shapes[keep] = st.shapes[i];
                    st.scores[keep] = st.scores[i];
                    st.masks[keep] = st.masks[i];
                    keep++;
                }
                st.n = keep;
            }
        }
        /* updatelevelweight: octree.jl:198-205, x = 9//10 */
        if (oct) update_level_probs(octP, octS, oct->depth);
        else {
            double wsum = 0;
            for (int i = 0; i < octree_depth; i++) wsum += levelscore[i] / levelweight[i];
            for (int i = 0; i < octree_depth; i++)
                levelweight[i] = 0.9 * levelscore[i] / (wsum * levelweight[i]) + (1 - 0.9) / octree_depth;
        }
        int64_t sT = countcandidates[p->terminate_s];
        if (orc_prob((double)p->tau, sT, c->n, p->drawN) > p->prob_det) { k++; break; }
    }
    out->iterations = k - 1;
    if (out->iterations > p->itermax) out->iterations = p->itermax;
done:
    out->candidates_scored = countcandidates[2];
    out->scored_left = st.n;
    for (int64_t i = 0; i < st.n; i++) free(st.masks[i]);
    free(st.shapes); free(st.scores); free(st.masks);
    free(cands); free(levels); free(sd); free(fp); free(fn); free(refit_idx);
    free(levelweight); free(levelscore);
    lo_free(oct);
    out->seconds = now_s() - t0;
    return rc;
}

void orc_result_free(orc_result *r)
{
    for (int64_t i = 0; i < r->n_shapes; i++) free(r->shapes[i].inpoints);
    free(r->shapes);
    memset(r, 0, sizeof *r);
}

/* ------------------------------- parameter-space bitmap + largest component */

/* Julia round(Int, x): RoundNearest, ties to even */
static int64_t jl_round(double x) { return (int64_t)nearbyint(x); }

/* bitmapparameters: parameterspacebitmap.jl:12-46 */
int orc_bitmapparameters(const double *prm2, const uint8_t *compat_, const int64_t *idsource,
                         int64_t n, double beta, int32_t *xs_, int32_t *ys_,
                         double *betax, double *betay, uint8_t *bitmap, int64_t *idxmap)
{
    double miv[2], mav[2];
    orc_findAABB(prm2, n, 2, miv, mav);
    double minv[2] = { miv[0] - 0.1, miv[1] - 0.1 }, maxv[2] = { mav[0] + 0.1, mav[1] + 0.1 };
    int64_t xs = jl_round((maxv[0] - minv[0]) / beta), ys = jl_round((maxv[1] - minv[1]) / beta);
    if (!(xs > 0 && ys > 0)) return -1;
    double bx = (maxv[0] - minv[0]) / (double)xs, by = (maxv[1] - minv[1]) / (double)ys;
    *xs_ = (int32_t)xs; *ys_ = (int32_t)ys; *betax = bx; *betay = by;
    if (!bitmap) return 0;
    memset(bitmap, 0, (size_t)(xs * ys));
    memset(idxmap, 0, 8 * (size_t)(xs * ys));
    for (int64_t i = 0; i < n; i++) {
        if (!compat_[i]) continue;
        int64_t xp = (int64_t)ceil((prm2[2 * i] - minv[0]) / bx);
        int64_t yp = (int64_t)ceil((prm2[2 * i + 1] - minv[1]) / by);
        if (xp != 0 && yp != 0 && xp != xs && yp != ys) {
            int64_t li = (xp - 1) + xs * (yp - 1);
            if (!bitmap[li]) { bitmap[li] = 1; idxmap[li] = idsource ? idsource[i] : i + 1; }
        }
    }
    return 0;
}

/* largestconncomp: parameterspacebitmap.jl:69-109.  Images.label_components
 * numbers components in column-major first-encounter order [recalled];
 * argmax picks the first largest; component_subscripts lists pixels in
 * column-major order. */
int64_t orc_largestconncomp(const uint8_t *bitmap, int32_t xs, int32_t ys, int conn8,
                            int64_t *out, int64_t cap)
{
    int64_t npx = (int64_t)xs * ys;
    int32_t *label = (int32_t *)calloc((size_t)(npx ? npx : 1), 4);
    int64_t *stack = (int64_t *)malloc(8 * (size_t)(npx ? npx : 1));
    int32_t nlab = 0, best = 0;
    int64_t bestsize = 0;
    for (int64_t li = 0; li < npx; li++) {
        if (!bitmap[li] || label[li]) continue;
        nlab++;
        int64_t sp = 0, size = 0;
        stack[sp++] = li;
        label[li] = nlab;
        while (sp) {
            int64_t cur = stack[--sp];
            size++;
            int32_t x = (int32_t)(cur % xs), y = (int32_t)(cur / xs);
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    if (!conn8 && dx && dy) continue;
                    int32_t nx = x + dx, ny = y + dy;
                    if (nx < 0 || ny < 0 || nx >= xs || ny >= ys) continue;
                    int64_t nl = nx + (int64_t)xs * ny;
                    if (bitmap[nl] && !label[nl]) { label[nl] = nlab; stack[sp++] = nl; }
                }
        }
        if (size > bestsize) { bestsize = size; best = nlab; }
    }
    int64_t k = 0;
    if (best)
        for (int64_t li = 0; li < npx; li++)
            if (label[li] == best) { if (k < cap) out[k] = li; k++; }
    free(label);
    free(stack);
    return k;
}
