/*
 * orc_f32.c -- CPU oracle for Float32 clouds (TEST INFRASTRUCTURE ONLY; see ransac_oracle.h).
 *
 * The reference builds a cloud of SVector{3,Float32} when asked to (RANSACCloud(...; force_eltype = Float32),
 * src/octree.jl:102-109); shapes fitted to such a cloud are Float32 as well, so every operation of the four
 * compatibles* (plane.jl:114-130 + project2plane :82-95, sphere.jl:144-172, cylinder.jl:194-221, cone.jl:132-153 +
 * project2cone :68-85 + rodrigues utilities.jl:19-24,32-43,61-64) is a binary32 operation in the same order as the
 * binary64 path.  The parameters stay what the caller made them: eps and cos(alpha) are Float64 unless
 * setfloattype (utilities.jl:488-504) converted them, and Julia compares a Float32 with a Float64 after promoting the
 * Float32 exactly -- here: the binary32 result is converted to double and compared with the double threshold (a
 * caller with Float32 parameters passes their values).  cos / sin of -opang/2 are host-computed (orc_shape v[7], v[8])
 * like in the binary64 path; orc32_shape_finalize rounds the oracle's binary64 values to binary32.
 *
 * This file is the binary32 twin of the per-point part of ransac_oracle.c: the same statements with `float`.
 * Shapes arrive as orc_shape (double fields); every field is converted to float on entry (exact for values that are
 * binary32 numbers, which is what a Float32 shape holds).  Build: -ffp-contract=off, no double promotion anywhere in
 * the arithmetic (every literal is a float literal).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orc_trig.h"
#include "ransac_oracle.h"

typedef struct { float x, y, z; } f3;

static inline f3 F(const float *p) { f3 r = { p[0], p[1], p[2] }; return r; }
static inline f3 Fd(const double *p) { f3 r = { (float)p[0], (float)p[1], (float)p[2] }; return r; }
static inline f3 fsub(f3 a, f3 b) { f3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline f3 fscale(f3 a, float s) { f3 r = { a.x * s, a.y * s, a.z * s }; return r; }
static inline f3 fneg(f3 a) { f3 r = { -a.x, -a.y, -a.z }; return r; }
static inline float fdot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float fnorm(f3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
static inline f3 fnormalize(f3 a) { float inv = 1.0f / fnorm(a); f3 r = { inv * a.x, inv * a.y, inv * a.z }; return r; }
static inline f3 fcross(f3 a, f3 b)
{
    f3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}

void orc32_shape_finalize(orc_shape *s)
{
    for (int i = 0; i < 7; i++) s->v[i] = (double)(float)s->v[i];   /* a Float32 shape holds binary32 numbers */
    if (s->kind == ORC_CONE) {
        /* rodriguesrad(rot_ax, -cone.opang/2) on a Float32 opang: cos / sin of a Float32 are Float32 */
        float th = -(float)s->v[6] / 2.0f;
        s->v[7] = (double)(float)orc_cos((double)th);
        s->v[8] = (double)(float)orc_sin((double)th);
    }
}

/* compatiblesPlane: plane.jl:114-130, project2plane :82-95, isparallel utilities.jl:115-117 */
static int compat_plane32(const orc_shape *s, f3 p, f3 n, double eps, double cosa)
{
    f3 point = Fd(&s->v[0]), normal = Fd(&s->v[3]);
    f3 o_z = fnormalize(normal);
    float d = fdot(o_z, fsub(p, point));
    return ((double)fdot(normal, n) > cosa) && ((double)fabsf(d) < eps);
}

/* compatiblesSphere: sphere.jl:144-172 */
static int compat_sphere32(const orc_shape *s, f3 p, f3 n, double eps, double cosa)
{
    f3 o = Fd(&s->v[0]);
    float R = (float)s->v[3];
    if (s->outwards)
        return ((double)fdot(fnormalize(fsub(p, o)), n) > cosa) && ((double)fabsf(fnorm(fsub(p, o)) - R) < eps);
    else
        return ((double)fdot(fnormalize(fsub(o, p)), n) > cosa) && ((double)fabsf(fnorm(fsub(p, o)) - R) < eps);
}

/* compatiblesCylinder: cylinder.jl:194-221 */
static int compat_cylinder32(const orc_shape *s, f3 p, f3 n, double eps, double cosa)
{
    f3 a = Fd(&s->v[0]), c = Fd(&s->v[3]);
    float R = (float)s->v[6];
    f3 curr_norm = fsub(fsub(p, fscale(a, fdot(a, fsub(p, c)))), c);
    if ((double)fabsf(fnorm(curr_norm) - R) < eps) {
        if (s->outwards)
            return (double)fdot(fnormalize(curr_norm), n) > cosa;
        else
            return (double)fdot(fneg(fnormalize(curr_norm)), n) > cosa;
    }
    return 0;
}

/* project2cone: cone.jl:68-85 with rodriguesrad / rodrigues / pluscrossprod! utilities.jl:61-64, 19-24, 32-43 */
static void project2cone32(const orc_shape *s, f3 p, float *dist, f3 *normal)
{
    f3 apex = Fd(&s->v[0]), axis = Fd(&s->v[3]);
    const float c = (float)s->v[7], sn = (float)s->v[8];
    f3 to_point = fsub(apex, p);
    f3 to_pointn = fnormalize(to_point);
    f3 rot_ax = fnormalize(fcross(axis, to_pointn));
    f3 comp_n = fnormalize(fcross(axis, rot_ax));
    f3 nvn = fnormalize(rot_ax);                       /* rodriguesrad re-normalizes the axis */
    float nv[3] = { nvn.x, nvn.y, nvn.z }, R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float nn = nv[i] * nv[j];
            float id = (i == j) ? 1.0f : 0.0f;
            R[i * 3 + j] = nn + c * (id - nn);
        }
    R[0 * 3 + 1] -= sn * nv[2];
    R[0 * 3 + 2] += sn * nv[1];
    R[1 * 3 + 0] += sn * nv[2];
    R[1 * 3 + 2] -= sn * nv[0];
    R[2 * 3 + 0] -= sn * nv[1];
    R[2 * 3 + 1] += sn * nv[0];
    f3 rc = {
        (R[0] * comp_n.x + R[1] * comp_n.y) + R[2] * comp_n.z,
        (R[3] * comp_n.x + R[4] * comp_n.y) + R[5] * comp_n.z,
        (R[6] * comp_n.x + R[7] * comp_n.y) + R[8] * comp_n.z,
    };
    f3 current_normal = fnormalize(rc);
    *dist = fdot(fneg(current_normal), fneg(to_point));
    *normal = current_normal;
}

/* compatiblesCone: cone.jl:132-153 */
static int compat_cone32(const orc_shape *s, f3 p, f3 n, double eps, double cosa)
{
    float dist;
    f3 cn;
    project2cone32(s, p, &dist, &cn);
    if (s->outwards)
        return ((double)fdot(cn, n) > cosa) && ((double)fabsf(dist) < eps);
    else
        return ((double)fdot(fneg(cn), n) > cosa) && ((double)fabsf(dist) < eps);
}

static inline int compat32(const orc_shape *s, f3 p, f3 n, double eps, double cosa)
{
    switch (s->kind) {
    case ORC_PLANE: return compat_plane32(s, p, n, eps, cosa);
    case ORC_SPHERE: return compat_sphere32(s, p, n, eps, cosa);
    case ORC_CYLINDER: return compat_cylinder32(s, p, n, eps, cosa);
    case ORC_CONE: return compat_cone32(s, p, n, eps, cosa);
    }
    return 0;
}

int orc32_compatible(const orc_shape *s, const float p[3], const float n[3], double eps, double cos_alpha)
{
    return compat32(s, F(p), F(n), eps, cos_alpha);
}

/* ------------------------------------------------------------------ cloud */
struct orc_cloud32 {
    int64_t n, s, nchunks;
    float *xyz, *nrm;    /* AoS, like Vector{SVector{3,Float32}} */
    int64_t *subset1;    /* 1-based */
    uint64_t *enabled;
};

orc_cloud32 *orc32_cloud_create(const float *xyz, const float *nrm, int64_t n, const int64_t *subset1, int64_t s)
{
    orc_cloud32 *c = (orc_cloud32 *)calloc(1, sizeof *c);
    c->n = n;
    c->s = s;
    c->nchunks = (n + 63) / 64;
    c->xyz = (float *)malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
    c->nrm = (float *)malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
    memcpy(c->xyz, xyz, sizeof(float) * 3 * (size_t)n);
    memcpy(c->nrm, nrm, sizeof(float) * 3 * (size_t)n);
    c->subset1 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s ? s : 1));
    memcpy(c->subset1, subset1, sizeof(int64_t) * (size_t)s);
    c->enabled = (uint64_t *)calloc((size_t)(c->nchunks ? c->nchunks : 1), 8);
    orc32_cloud_enable_all(c);
    return c;
}

void orc32_cloud_destroy(orc_cloud32 *c)
{
    if (!c) return;
    free(c->xyz); free(c->nrm); free(c->subset1); free(c->enabled);
    free(c);
}

void orc32_cloud_enable_all(orc_cloud32 *c)
{
    for (int64_t i = 0; i < c->nchunks; i++) c->enabled[i] = ~0ULL;
    if (c->n % 64) c->enabled[c->nchunks - 1] = (~0ULL) >> (64 - c->n % 64);
}

void orc32_cloud_set_enabled(orc_cloud32 *c, const uint64_t *chunks, int64_t nchunks)
{
    int64_t m = nchunks < c->nchunks ? nchunks : c->nchunks;
    memcpy(c->enabled, chunks, 8 * (size_t)m);
    if (c->n % 64 && m == c->nchunks) c->enabled[c->nchunks - 1] &= (~0ULL) >> (64 - c->n % 64);
}

void orc32_cloud_get_enabled(const orc_cloud32 *c, uint64_t *chunks, int64_t nchunks)
{
    int64_t m = nchunks < c->nchunks ? nchunks : c->nchunks;
    memcpy(chunks, c->enabled, 8 * (size_t)m);
}

static inline int is_enabled32(const orc_cloud32 *c, int64_t i0) { return (int)((c->enabled[i0 >> 6] >> (i0 & 63)) & 1); }

/* scorecandidate: plane.jl:61-71, sphere.jl:118-134 (Q4: ignores isenabled), cylinder.jl:172-183, cone.jl:155-167 */
int64_t orc32_scorecandidate(const orc_cloud32 *c, const orc_shape *s, const orc_params *p, int64_t *inpoints, uint64_t *mask)
{
    double eps = p->eps[s->kind], cosa = p->cos_alpha[s->kind];
    int use_en = (s->kind != ORC_SPHERE) || p->sphere_uses_enabled;
    int64_t cnt = 0;
    if (mask) memset(mask, 0, 8 * (size_t)((c->s + 63) / 64));
    for (int64_t j = 0; j < c->s; j++) {
        int64_t i0 = c->subset1[j] - 1;
        int ok = compat32(s, F(&c->xyz[3 * i0]), F(&c->nrm[3 * i0]), eps, cosa);
        if (use_en) ok = ok & is_enabled32(c, i0);
        if (ok) {
            if (inpoints) inpoints[cnt] = i0 + 1;
            if (mask) mask[j >> 6] |= 1ULL << (j & 63);
            cnt++;
        }
    }
    return cnt;
}

void orc32_score_masks_mt(const orc_cloud32 *c, const orc_shape *s, int32_t b, const orc_params *p, int32_t *counts,
                          uint64_t *masks, int32_t nthreads)
{
    int64_t w = (c->s + 63) / 64;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int32_t i = 0; i < b; i++)
        counts[i] = (int32_t)orc32_scorecandidate(c, &s[i], p, NULL, masks ? masks + (size_t)i * w : NULL);
}

/* refit: plane.jl:137-143, sphere.jl:179-190, cylinder.jl:228-234, cone.jl:176-182 */
int64_t orc32_refit(const orc_cloud32 *c, const orc_shape *s, const orc_params *p, int64_t *idx_out, int64_t cap)
{
    double eps = p->eps[s->kind], cosa = p->cos_alpha[s->kind];
    int64_t cnt = 0;
    for (int64_t i0 = 0; i0 < c->n; i0++) {
        if (!is_enabled32(c, i0)) continue;
        if (compat32(s, F(&c->xyz[3 * i0]), F(&c->nrm[3 * i0]), eps, cosa)) {
            if (cnt < cap) idx_out[cnt] = i0 + 1;
            cnt++;
        }
    }
    return cnt;
}

/* invalidate_indexes!: fitting.jl:197-202 */
void orc32_invalidate(orc_cloud32 *c, const int64_t *idx, int64_t n)
{
    for (int64_t k = 0; k < n; k++) {
        int64_t i0 = idx[k] - 1;
        c->enabled[i0 >> 6] &= ~(1ULL << (i0 & 63));
    }
}


/* ------------------------------------------------------------------------------------------------------------------
 * Minimal-set fits on a Float32 cloud: plane.jl:33-57, sphere.jl:29-75 + 87-114, cylinder.jl:34-125 + 135-168 with
 * SVector{3,Float32} points and normals -- every vector operation is a binary32 operation, integer literals of the
 * source ((v[1]+v[2])/2, -1*crossv) take the vectors' type, and the Float64 parameters only ever appear on one side of
 * a comparison (collin_threshold, cosd(parallelthrdeg), sphere_par, cos(alpha), eps: the Float32 side is promoted
 * exactly).  The statements follow ransac_oracle.c's binary64 fits one for one; p / n arrive as doubles holding the
 * Float32 values.  The cone's fit (round 5): cone.jl:39-61 + 87-128 in binary32 -- rank() and \ of a Matrix{Float32} are
 * LAPACK's single-precision SVD and LU upstream, restated here as a one-sided Jacobi SVD and an LU with partial pivoting
 * in float, acos as the oracle's double kernel rounded once (Julia's Float32 acos is accurate to < 1 ulp, not always
 * correctly rounded): UNPINNED beyond the restatement, like the Float64 cone fit -- the reference holds no cone fixture.
 */
static inline f3 fadd(f3 a, f3 b) { f3 r = { a.x + b.x, a.y + b.y, a.z + b.z }; return r; }
static inline f3 fdivs(f3 a, float s) { f3 r = { a.x / s, a.y / s, a.z / s }; return r; }
static void set_f(double *dst, f3 a) { dst[0] = (double)a.x; dst[1] = (double)a.y; dst[2] = (double)a.z; }

static int fit_plane32(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3) return 0;
    f3 p1 = Fd(p), p2 = Fd(p + 3), p3 = Fd(p + 6);
    f3 crossv = fnormalize(fcross(fsub(p2, p1), fsub(p3, p1)));
    if ((double)fnorm(crossv) < prm->collin_threshold) return 0;
    double thr = prm->cos_alpha[ORC_PLANE];
    int all_ok = 1, all_inv = 1;
    for (int i = 0; i < lp; i++) {
        float dotp = fdot(crossv, fnormalize(Fd(n + 3 * i)));
        if (!((double)dotp > thr)) all_ok = 0;
        if (!((double)dotp < -thr)) all_inv = 0;
    }
    memset(out, 0, sizeof *out);
    out->kind = ORC_PLANE;
    if (all_ok) { set_f(&out->v[0], p1); set_f(&out->v[3], crossv); return 1; }
    if (all_inv) { set_f(&out->v[0], p1); set_f(&out->v[3], fscale(crossv, -1.0f)); return 1; }
    return 0;
}

static int fit2pointsphere32(const double *vv, const double *nn, const orc_params *prm, f3 *center_out, float *radius_out)
{
    f3 v1 = Fd(vv), v2 = Fd(vv + 3), n1 = Fd(nn), n2v = Fd(nn + 3);
    f3 n1n = fnormalize(n1), n2n = fnormalize(n2v);
    f3 center;
    float radius;
    if ((double)fabsf(fdot(n1n, n2n)) > prm->cos_parallelthr) {
        center = fdivs(fadd(v1, v2), 2.0f);
        radius = fnorm(fsub(center, v1));
    } else {
        f3 g = fsub(v2, v1);
        f3 h = fcross(n2n, g);
        f3 k = fcross(n2n, n1n);
        float nk = fnorm(k), nh = fnorm(h);
        if ((double)nk < prm->sphere_par || (double)nh < prm->sphere_par) {
            f3 n2 = fcross(n2n, fcross(n1n, n2n));
            f3 n1_ = fcross(n1n, fcross(n2n, n1n));
            f3 c1 = fadd(v1, fscale(n1, fdot(fsub(v2, v1), n2) / fdot(n1, n2)));
            f3 c2 = fadd(v2, fscale(n2v, fdot(fsub(v1, v2), n1_) / fdot(n2v, n1_)));
            center = fdivs(fadd(c1, c2), 2.0f);
            radius = (fnorm(fsub(v1, center)) + fnorm(fsub(v1, center))) / 2.0f;
        } else if (fdot(h, k) > 0.0f) {
            center = fadd(v1, fscale(n1n, nh / nk));
            radius = fnorm(fsub(center, v1));
        } else {
            center = fsub(v1, fscale(n1n, nh / nk));
            radius = fnorm(fsub(center, v1));
        }
    }
    *center_out = center;
    *radius_out = radius;
    return 1;
}

static int fit_sphere32(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3) return 0;
    f3 center;
    float radius;
    if (!fit2pointsphere32(p, n, prm, &center, &radius)) return 0;
    double thr = prm->cos_alpha[ORC_SPHERE], eps = prm->eps[ORC_SPHERE];
    int vert = 1, ok = 1, inv = 1;
    for (int i = 0; i < lp; i++) {
        f3 pi = Fd(p + 3 * i);
        if (!((double)fabsf(fnorm(fsub(pi, center)) - radius) < eps)) vert = 0;
        float dotp = fdot(fnormalize(fsub(pi, center)), fnormalize(Fd(n + 3 * i)));
        if (!((double)dotp > thr)) ok = 0;
        if (!((double)dotp < -thr)) inv = 0;
    }
    if (!vert) return 0;
    memset(out, 0, sizeof *out);
    out->kind = ORC_SPHERE;
    set_f(&out->v[0], center);
    out->v[3] = (double)radius;
    if (ok) { out->outwards = 1; return 1; }
    if (inv) { out->outwards = 0; return 1; }
    return 0;
}

static f3 cyl_project2plane32(f3 n, f3 w) { return fadd(w, fscale(n, fdot(fneg(n), w) / fdot(n, n))); }

static void cyl_projectto2d32(f3 xa, f3 ya, f3 za, f3 p1, float r[2])
{
    float xx = xa.x, xy = xa.y, xz = xa.z;
    float yx = ya.x, yy = ya.y, yz = ya.z;
    float zx = za.x, zy = za.y, zz = za.z;
    float px = p1.x, py = p1.y, pz = p1.z;
    r[0] = -((-(pz * yy * zx) + py * yz * zx + pz * yx * zy - px * yz * zy - py * yx * zz + px * yy * zz) /
             (xz * yy * zx - xy * yz * zx - xz * yx * zy + xx * yz * zy + xy * yx * zz - xx * yy * zz));
    r[1] = -((pz * xy * zx - py * xz * zx - pz * xx * zy + px * xz * zy + py * xx * zz - px * xy * zz) /
             (xz * yy * zx - xy * yz * zx - xz * yx * zy + xx * yz * zy + xy * yx * zz - xx * yy * zz));
}

static int fit_cylinder32(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3) return 0;
    f3 p1 = Fd(p), p2 = Fd(p + 3), n1 = Fd(n), n2 = Fd(n + 3);
    if ((double)fabsf(fdot(n1, n2)) > prm->cos_parallelthr) return 0;
    f3 an = fnormalize(fcross(n1, n2));
    f3 xax = fnormalize(cyl_project2plane32(an, p1));
    f3 yax = fnormalize(fcross(an, xax));
    float p11[2], p12[2], p21[2], p22[2];
    cyl_projectto2d32(xax, yax, an, cyl_project2plane32(an, p1), p11);
    cyl_projectto2d32(xax, yax, an, cyl_project2plane32(an, fadd(p1, n1)), p12);
    cyl_projectto2d32(xax, yax, an, cyl_project2plane32(an, p2), p21);
    cyl_projectto2d32(xax, yax, an, cyl_project2plane32(an, fadd(p2, n2)), p22);
    float amb[2] = { p11[0] - p12[0], p11[1] - p12[1] };
    float cmd[2] = { p21[0] - p22[0], p21[1] - p22[1] };
    float d1 = p11[0] * p12[1] - p11[1] * p12[0];
    float d2 = p21[0] * p22[1] - p21[1] * p22[0];
    float d3 = amb[0] * cmd[1] - amb[1] * cmd[0];
    float interc[2] = { (d1 * cmd[0] - d2 * amb[0]) / d3, (d1 * cmd[1] - d2 * amb[1]) / d3 };
    f3 c = fadd(fscale(xax, interc[0]), fscale(yax, interc[1]));
    float nn1 = fnorm(fsub(fsub(p1, c), fscale(an, fdot(an, fsub(p1, c)))));
    float nn2 = fnorm(fsub(fsub(p2, c), fscale(an, fdot(an, fsub(p2, c)))));
    float R = (nn1 + nn2) / 2.0f;
    float outw = (p12[0] - p11[0]) * (p11[0] - interc[0]) + (p12[1] - p11[1]) * (p11[1] - interc[1]);
    (void)outw;   /* fit() overwrites the outerity (cylinder.jl:165-166) */
    double thr = prm->cos_alpha[ORC_CYLINDER], eps = prm->eps[ORC_CYLINDER];
    int vert = 1, ok = 1, inv = 1;
    for (int i = 0; i < lp; i++) {
        f3 pi = Fd(p + 3 * i);
        f3 curr_norm = fsub(fsub(pi, fscale(an, fdot(an, fsub(pi, c)))), c);
        if (!((double)fabsf(fnorm(curr_norm) - R) < eps)) vert = 0;
        float dotp = fdot(fnormalize(curr_norm), Fd(n + 3 * i));
        if (!((double)dotp > thr)) ok = 0;
        if (!((double)dotp < -thr)) inv = 0;
    }
    if (!vert) return 0;
    memset(out, 0, sizeof *out);
    out->kind = ORC_CYLINDER;
    set_f(&out->v[0], an);
    set_f(&out->v[3], c);
    out->v[6] = (double)R;
    if (ok) { out->outwards = 1; return 1; }
    if (inv) { out->outwards = 0; return 1; }
    return 0;
}

/* ---- cone.jl:39-61 in binary32 ---- */
static void svdvals_cols32(float *M, int r, int c, float *sv)   /* one-sided Jacobi on the columns, r >= c */
{
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < c - 1; p++)
            for (int q = p + 1; q < c; q++) {
                float a = 0, b = 0, g = 0;
                for (int i = 0; i < r; i++) {
                    a += M[i * c + p] * M[i * c + p];
                    b += M[i * c + q] * M[i * c + q];
                    g += M[i * c + p] * M[i * c + q];
                }
                if (g == 0.0f || fabsf(g) <= 1e-37f + 1.2e-7f * sqrtf(a * b)) continue;
                rotated = 1;
                float zeta = (b - a) / (2 * g);
                float t = (zeta >= 0 ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1 + zeta * zeta));
                float cs = 1 / sqrtf(1 + t * t), sn = cs * t;
                for (int i = 0; i < r; i++) {
                    float mp = M[i * c + p], mq = M[i * c + q];
                    M[i * c + p] = cs * mp - sn * mq;
                    M[i * c + q] = sn * mp + cs * mq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < c; j++) {
        float a = 0;
        for (int i = 0; i < r; i++) a += M[i * c + j] * M[i * c + j];
        sv[j] = sqrtf(a);
    }
}

static int rank32(const float *A, int m, int n)   /* count(svdvals .> min(m,n) * eps(Float32) * maximum(svdvals)) */
{
    float M[16], sv[4];
    int r, c;
    if (m >= n) { r = m; c = n; for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[i * c + j] = A[i * n + j]; }
    else { r = n; c = m; for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[j * c + i] = A[i * n + j]; }
    svdvals_cols32(M, r, c, sv);
    float smax = 0;
    for (int j = 0; j < c; j++) if (sv[j] > smax) smax = sv[j];
    float tol = (float)(m < n ? m : n) * 1.1920928955078125e-07f * smax;
    int cnt = 0;
    for (int j = 0; j < c; j++) if (sv[j] > tol) cnt++;
    return cnt;
}

static int solve3_32(const float A_in[9], const float b_in[3], float x[3])   /* r \ ds: LU with partial pivoting */
{
    float A[9], b[3];
    memcpy(A, A_in, sizeof A);
    memcpy(b, b_in, sizeof b);
    for (int k = 0; k < 3; k++) {
        int piv = k;
        float amax = fabsf(A[k * 3 + k]);
        for (int i = k + 1; i < 3; i++)
            if (fabsf(A[i * 3 + k]) > amax) { amax = fabsf(A[i * 3 + k]); piv = i; }
        if (amax == 0.0f) return -1;
        if (piv != k) {
            for (int j = 0; j < 3; j++) { float t = A[k * 3 + j]; A[k * 3 + j] = A[piv * 3 + j]; A[piv * 3 + j] = t; }
            float t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < 3; i++) {
            float l = A[i * 3 + k] / A[k * 3 + k];
            A[i * 3 + k] = l;
            for (int j = k + 1; j < 3; j++) A[i * 3 + j] -= l * A[k * 3 + j];
            b[i] -= l * b[k];
        }
    }
    for (int i = 2; i >= 0; i--) {
        float sacc = b[i];
        for (int j = i + 1; j < 3; j++) sacc -= A[i * 3 + j] * x[j];
        x[i] = sacc / A[i * 3 + i];
    }
    return 0;
}

static float clamp1f(float x) { return x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x); }

static int fit3pointcone32(const double *p, const double *n, orc_shape *out)
{
    float r[9], rv[12], ds[3], apx[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = (float)n[3 * i + j];
    if (rank32(r, 3, 3) != 3) return 0;
    for (int i = 0; i < 3; i++) ds[i] = fdot(Fd(p + 3 * i), Fd(n + 3 * i));
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) rv[i * 4 + j] = r[i * 3 + j];
        rv[i * 4 + 3] = -1 * ds[i];
    }
    if (rank32(rv, 3, 4) != 3) return 0;
    if (solve3_32(r, ds, apx)) return 0;
    f3 ap = { apx[0], apx[1], apx[2] };
    f3 a3p[3];
    for (int i = 0; i < 3; i++) {
        f3 d = fsub(Fd(p + 3 * i), ap);
        a3p[i] = fadd(ap, fdivs(d, fnorm(d)));
    }
    f3 ax = fnormalize(fcross(fsub(a3p[1], a3p[0]), fsub(a3p[2], a3p[0])));
    f3 midp = fdivs(fadd(fadd(a3p[0], a3p[1]), a3p[2]), 3.0f);
    f3 dirv = fnormalize(fsub(midp, ap));
    if (fdot(ax, dirv) < 0) ax = fscale(ax, -1.0f);
    float angles[3];
    for (int i = 0; i < 3; i++) angles[i] = (float)orc_acos((double)clamp1f(fdot(fnormalize(fsub(Fd(p + 3 * i), ap)), ax)));
    float opangle = 2 * ((angles[0] + angles[1]) + angles[2]) / 3;
    memset(out, 0, sizeof *out);
    out->kind = ORC_CONE;
    out->outwards = 1;
    set_f(&out->v[0], ap);
    set_f(&out->v[3], ax);
    out->v[6] = (double)opangle;
    orc32_shape_finalize(out);
    return 1;
}

/* validatecone: cone.jl:87-115; fit(::Type{FittedCone}): cone.jl:123-128 */
static int fit_cone32(const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    if (lp < 3 || lp > 16) return 0;
    orc_shape cone;
    if (!fit3pointcone32(p, n, &cone)) return 0;
    float dist[16];
    f3 cn[16];
    for (int i = 0; i < lp; i++) project2cone32(&cone, Fd(p + 3 * i), &dist[i], &cn[i]);
    for (int i = 0; i < lp; i++)
        if ((double)dist[i] > prm->eps[ORC_CONE]) return 0; /* no abs: Q11, cone.jl:93 */
    if (cone.v[6] < prm->minconeopang) return 0;
    double thr = prm->cos_alpha[ORC_CONE];
    int ok = 1, inv = 1;
    for (int i = 0; i < lp; i++) {
        double dotp = (double)fdot(cn[i], Fd(n + 3 * i));
        if (!(dotp > thr)) ok = 0;
        if (!(dotp < -thr)) inv = 0;
    }
    *out = cone;
    if (ok) { out->outwards = 1; return 1; }
    if (inv) { out->outwards = 0; return 1; }
    return 0;
}

/* fit(T, p, n, pc, params) on Float32 points */
int orc32_fit(int kind, const double *p, const double *n, int lp, const orc_params *prm, orc_shape *out)
{
    switch (kind) {
    case ORC_PLANE: return fit_plane32(p, n, lp, prm, out);
    case ORC_SPHERE: return fit_sphere32(p, n, lp, prm, out);
    case ORC_CYLINDER: return fit_cylinder32(p, n, lp, prm, out);
    case ORC_CONE: return fit_cone32(p, n, lp, prm, out);
    default: return 0;
    }
}
