// fit.cpp -- host-side pieces of the plugin API: minimal-set fits, score statistics,
// parameters, RNG.  O(1) per minimal set, so they stay on the host like in the reference.
// binary64, the reference's operation order, built with -ffp-contract=off.
// Paths in comments are under /root/reference/src.
#include <math.h>
#include <string.h>

#include "fit_shared.h"
#include "rh_internal.h"

namespace {

using namespace rhfit;

constexpr double kPi = 3.14159265358979323846;

// ---- LinearAlgebra stand-ins for cone.jl:44,48,50 ----
// singular values via one-sided Jacobi on the columns (rows x cols, rows >= cols)
void jacobi_svals(double *M, int rows, int cols, double *sv)
{
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p + 1 < cols; p++)
            for (int q = p + 1; q < cols; q++) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < rows; i++) {
                    a += M[i * cols + p] * M[i * cols + p];
                    b += M[i * cols + q] * M[i * cols + q];
                    g += M[i * cols + p] * M[i * cols + q];
                }
                if (g == 0.0 || fabs(g) <= 1e-300 + 2.2e-16 * sqrt(a * b)) continue;
                rotated = true;
                const double zeta = (b - a) / (2 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                const double cs = 1 / sqrt(1 + t * t), sn = cs * t;
                for (int i = 0; i < rows; i++) {
                    const double mp = M[i * cols + p], mq = M[i * cols + q];
                    M[i * cols + p] = cs * mp - sn * mq;
                    M[i * cols + q] = sn * mp + cs * mq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < cols; j++) {
        double a = 0;
        for (int i = 0; i < rows; i++) a += M[i * cols + j] * M[i * cols + j];
        sv[j] = sqrt(a);
    }
}

int matrix_rank(const double *A, int m, int n)   // rank(A): count(svdvals .> min(m,n)*eps*max)
{
    double M[16], sv[4];
    int rows, cols;
    if (m >= n) {
        rows = m; cols = n;
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[i * cols + j] = A[i * n + j];
    } else {
        rows = n; cols = m;
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[j * cols + i] = A[i * n + j];
    }
    jacobi_svals(M, rows, cols, sv);
    double smax = 0;
    for (int j = 0; j < cols; j++) smax = sv[j] > smax ? sv[j] : smax;
    const double tol = (double)(m < n ? m : n) * 2.220446049250313e-16 * smax;
    int r = 0;
    for (int j = 0; j < cols; j++) r += sv[j] > tol;
    return r;
}

bool lu_solve3(const double A0[9], const double b0[3], double x[3])   // A \ b, partial pivoting
{
    double A[9], b[3];
    memcpy(A, A0, sizeof A);
    memcpy(b, b0, sizeof b);
    for (int k = 0; k < 3; k++) {
        int piv = k;
        double best = fabs(A[k * 3 + k]);
        for (int i = k + 1; i < 3; i++)
            if (fabs(A[i * 3 + k]) > best) { best = fabs(A[i * 3 + k]); piv = i; }
        if (best == 0.0) return false;
        if (piv != k) {
            for (int j = 0; j < 3; j++) { const double t = A[k * 3 + j]; A[k * 3 + j] = A[piv * 3 + j]; A[piv * 3 + j] = t; }
            const double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < 3; i++) {
            const double l = A[i * 3 + k] / A[k * 3 + k];
            A[i * 3 + k] = l;
            for (int j = k + 1; j < 3; j++) A[i * 3 + j] -= l * A[k * 3 + j];
            b[i] -= l * b[k];
        }
    }
    for (int i = 2; i >= 0; i--) {
        double acc = b[i];
        for (int j = i + 1; j < 3; j++) acc -= A[i * 3 + j] * x[j];
        x[i] = acc / A[i * 3 + i];
    }
    return true;
}

// ---- cone.jl:68-85 (host twin of the device test, used by validatecone) ----
void project2cone(const rh_shape &cone, const Vec &p, double *dist, Vec *cn)
{
    const Vec apex(cone.v), axis(cone.v + 3);
    const Vec to_point = apex - p;
    const Vec to_pointn = normalize(to_point);
    const Vec rot_ax = normalize(cross(axis, to_pointn));
    const Vec comp_n = normalize(cross(axis, rot_ax));
    const Vec v = normalize(rot_ax);   // rodriguesrad re-normalizes (utilities.jl:62)
    const double c = cone.v[7], s = cone.v[8];
    const double e[3] = { v.x, v.y, v.z };
    double R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double nn = e[i] * e[j];
            R[i][j] = nn + c * ((i == j ? 1.0 : 0.0) - nn);
        }
    R[0][1] -= s * e[2]; R[0][2] += s * e[1];   // pluscrossprod!: utilities.jl:32-43
    R[1][0] += s * e[2]; R[1][2] -= s * e[0];
    R[2][0] -= s * e[1]; R[2][1] += s * e[0];
    const Vec rc((R[0][0] * comp_n.x + R[0][1] * comp_n.y) + R[0][2] * comp_n.z,
                 (R[1][0] * comp_n.x + R[1][1] * comp_n.y) + R[1][2] * comp_n.z,
                 (R[2][0] * comp_n.x + R[2][1] * comp_n.y) + R[2][2] * comp_n.z);
    *cn = normalize(rc);
    *dist = dot(-*cn, -to_point);
}

inline double clamp_unit(double x) { return x < -1 ? -1 : (x > 1 ? 1 : x); }

// ---- cone.jl:39-61 ----
bool fit3pointcone(const double *p, const double *n, rh_shape *cone)
{
    double r[9], rv[12], ds[3], ap[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = n[3 * i + j];
    if (matrix_rank(r, 3, 3) != 3) return false;
    for (int i = 0; i < 3; i++) ds[i] = dot(Vec(p + 3 * i), Vec(n + 3 * i));
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) rv[i * 4 + j] = r[i * 3 + j];
        rv[i * 4 + 3] = -1 * ds[i];
    }
    if (matrix_rank(rv, 3, 4) != 3) return false;
    if (!lu_solve3(r, ds, ap)) return false;
    const Vec apex(ap);
    Vec a3[3];
    for (int i = 0; i < 3; i++) {
        const Vec d = Vec(p + 3 * i) - apex;
        a3[i] = apex + d / norm(d);
    }
    Vec ax = normalize(cross(a3[1] - a3[0], a3[2] - a3[0]));
    const Vec midp = ((a3[0] + a3[1]) + a3[2]) / 3;
    const Vec dirv = normalize(midp - apex);
    if (dot(ax, dirv) < 0) ax = -1.0 * ax;
    double ang[3];
    for (int i = 0; i < 3; i++) ang[i] = acos(clamp_unit(dot(normalize(Vec(p + 3 * i) - apex), ax)));
    memset(cone, 0, sizeof *cone);
    cone->kind = RH_CONE;
    cone->outwards = 1;
    apex.store(cone->v);
    ax.store(cone->v + 3);
    cone->v[6] = 2 * ((ang[0] + ang[1]) + ang[2]) / 3;
    rh_shape_finalize(cone);
    return true;
}

// ---- cone.jl:87-115, 123-128 ----
bool fit_cone(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out)
{
    if (lp > 16) return false;
    rh_shape cone;
    if (!fit3pointcone(p, n, &cone)) return false;
    double dist[16];
    Vec cn[16];
    for (int i = 0; i < lp; i++) project2cone(cone, Vec(p + 3 * i), &dist[i], &cn[i]);
    for (int i = 0; i < lp; i++)
        if (dist[i] > prm.eps[RH_CONE]) return false;   // no abs in the reference (cone.jl:93)
    if (cone.v[6] < prm.minconeopang) return false;
    const double thr = prm.cos_alpha[RH_CONE];
    bool same = true, opposite = true;
    for (int i = 0; i < lp; i++) {
        const double dotp = dot(cn[i], Vec(n + 3 * i));
        same = same && (dotp > thr);
        opposite = opposite && (dotp < -thr);
    }
    if (!same && !opposite) return false;
    *out = cone;
    out->outwards = same ? 1 : 0;
    return true;
}

inline double julia_min(double x, double y) { return x != x ? x : (y != y ? y : (y < x ? y : x)); }
inline double julia_max(double x, double y) { return x != x ? x : (y != y ? y : (x < y ? y : x)); }

uint64_t splitmix(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

}  // namespace

// ------------------------------------------------------------------- ABI ----
extern "C" void rh_params_finalize(rh_params *p)
{
    for (int k = 0; k < 4; k++) p->cos_alpha[k] = cos(p->alpha[k]);
    p->cos_parallelthr = cos(p->parallelthrdeg * kPi / 180.0);   // cosd: sphere.jl:37, cylinder.jl:40
}

// utilities.jl:345,371; plane.jl:22; sphere.jl:25; cylinder.jl:27; cone.jl:30; RANSAC.jl:94
extern "C" void rh_default_params(rh_params *p)
{
    memset(p, 0, sizeof *p);
    for (int k = 0; k < 4; k++) { p->eps[k] = 0.3; p->alpha[k] = 5.0 * kPi / 180.0; }
    p->collin_threshold = 0.2;
    p->parallelthrdeg = 1.0;
    p->sphere_par = 0.02;
    p->minconeopang = 2.0 * kPi / 180.0;
    p->prob_det = 0.9;
    p->tau = 900;
    p->itermax = 1000;
    p->drawN = 3;
    p->minsubsetN = 15;
    p->extract_s = RH_S_NOFMINSET;
    p->terminate_s = RH_S_NOFMINSET;
    p->n_shape_types = 4;
    p->shape_types[0] = RH_PLANE;
    p->shape_types[1] = RH_CONE;
    p->shape_types[2] = RH_CYLINDER;
    p->shape_types[3] = RH_SPHERE;
    p->score_mode = RH_SCORE_INT64_WRAP;
    p->sphere_uses_enabled = 0;
    p->octree_max_depth = 10;
    rh_params_finalize(p);
}

extern "C" void rh_shape_finalize(rh_shape *s)
{
    if (s->kind == RH_CONE) {
        const double th = -s->v[6] / 2;   // rodriguesrad(rot_ax, -cone.opang/2): cone.jl:76
        s->v[7] = cos(th);
        s->v[8] = sin(th);
    }
}

extern "C" int rh_fit(int kind, const double *p, const double *n, int32_t lp, const rh_params *prm, rh_shape *out,
                      int32_t *fitted)
{
    if (!p || !n || !prm || !out || !fitted) { rh_set_error("rh_fit: NULL argument"); return RH_E_INVALID; }
    if (lp < 3) { rh_set_error("rh_fit: at least 3 points are needed (got %d)", lp); return RH_E_INVALID; }   // @assert
    rh_shape s;
    memset(&s, 0, sizeof s);
    bool ok;
    switch (kind) {
    case RH_PLANE: ok = fit_plane(p, n, lp, *prm, &s); break;
    case RH_SPHERE: ok = fit_sphere(p, n, lp, *prm, &s); break;
    case RH_CYLINDER: ok = fit_cylinder(p, n, lp, *prm, &s); break;
    case RH_CONE: ok = fit_cone(p, n, lp, *prm, &s); break;
    default: rh_set_error("rh_fit: unknown kind %d", kind); return RH_E_INVALID;
    }
    *fitted = ok ? 1 : 0;
    if (ok) *out = s;
    return RH_OK;
}

// estimatescore + hypergeomdev + notsoconfident: confidenceintervals.jl:71-74, 53-59, 20-22
extern "C" int rh_estimatescore(int64_t S1length, int64_t Plength, int64_t sigma, int32_t score_mode, double *ci_min,
                                double *ci_max, double *ci_E)
{
    const int64_t N = -2 - S1length, x = -2 - Plength, n = -1 - sigma;
    double sq_, xn;
    if (score_mode == RH_SCORE_INT64_WRAP) {
        // Julia Int64 arithmetic wraps silently; unsigned multiplication has the same bits
        const uint64_t xn_u = (uint64_t)x * (uint64_t)n;
        const uint64_t prod = xn_u * (uint64_t)(N - x) * (uint64_t)(N - n);
        sq_ = (double)(int64_t)prod / (double)(N - 1);
        xn = (double)(int64_t)xn_u;
    } else if (score_mode == RH_SCORE_F64) {
        const double xd = (double)x, nd = (double)n, Nd = (double)N;
        sq_ = (xd * nd * (Nd - xd) * (Nd - nd)) / (Nd - 1);
        xn = xd * nd;
    } else {
        rh_set_error("rh_estimatescore: unknown score_mode %d", score_mode);
        return RH_E_INVALID;
    }
    const double sq = sq_ < 0 ? 0.0 : sqrt(sq_);
    const double a = -1 - (xn + sq) / (double)N, b = -1 - (xn - sq) / (double)N;
    const double lo = julia_min(a, b), hi = julia_max(a, b);
    if (ci_min) *ci_min = lo;
    if (ci_max) *ci_max = hi;
    if (ci_E) *ci_E = (lo + hi) / 2;
    return RH_OK;
}

// prob(n, s, N, k) = 1-(1-(n/N)^k)^s: utilities.jl:262
extern "C" double rh_prob(double n, int64_t s, int64_t N, int64_t k)
{
    return 1 - pow(1 - pow(n / (double)N, (double)k), (double)s);
}

extern "C" void rh_rng_seed(rh_rng *r, uint64_t seed)
{
    memset(r, 0, sizeof *r);
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) r->s[i] = splitmix(&x);
}

static uint64_t rng_next(rh_rng *r)
{
    r->draws++;
    if (r->stream && r->stream_pos < r->stream_len) return r->stream[r->stream_pos++];
    uint64_t *s = r->s;
    const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

extern "C" int64_t rh_rng_range(rh_rng *r, int64_t n)
{
    return 1 + (int64_t)(((unsigned __int128)rng_next(r) * (unsigned __int128)(uint64_t)n) >> 64);
}
