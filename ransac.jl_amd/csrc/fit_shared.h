// fit_shared.h -- minimal-set fits for plane / sphere / cylinder / cone and the per-set random
// streams, compiled for BOTH host and device (hipcc, -ffp-contract=off): only + - * / sqrt fabs and
// comparisons are used (the cone's acos / cos / sin come from det_math.h, built from the same
// operations), so the two sides produce bit-identical candidates.
// Paths in comments are under /root/reference/src.
#pragma once

#if defined(RH_OCT_TIMING)
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdint.h>

#include "det_math.h"
#include "ransac_hip.h"

#if defined(__HIPCC__)
#define RH_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RH_HD inline
#endif

namespace rhfit {

// StaticArrays-style 3-vector: dot / norm sum left to right, normalize multiplies by 1/norm.
// T = double, or float for a Float32 cloud (RANSACCloud(...; force_eltype = Float32), octree.jl:102-109: the points, the
// normals and everything computed from them are Float32 there; the parameters stay Float64 unless setfloattype,
// utilities.jl:488-503, converted them, and Julia compares a Float32 with a Float64 after promoting the Float32 exactly
// -- here the float result is promoted by the comparison with the double threshold).  Integer literals of the Julia
// source ((v1 + v2) / 2, -1 * crossv) take the vector's type.
RH_HD double rh_sqrt_t(double x) { return sqrt(x); }
RH_HD float rh_sqrt_t(float x) { return sqrtf(x); }
RH_HD double rh_fabs_t(double x) { return fabs(x); }
RH_HD float rh_fabs_t(float x) { return fabsf(x); }

template <typename T>
struct VecT {
    T x, y, z;
    RH_HD VecT() : x(0), y(0), z(0) {}
    RH_HD VecT(T a, T b, T c) : x(a), y(b), z(c) {}
    RH_HD explicit VecT(const double *p) : x((T)p[0]), y((T)p[1]), z((T)p[2]) {}   // (exact: a Float32 cloud's values are binary32 numbers)
    RH_HD VecT operator+(const VecT &o) const { return VecT(x + o.x, y + o.y, z + o.z); }
    RH_HD VecT operator-(const VecT &o) const { return VecT(x - o.x, y - o.y, z - o.z); }
    RH_HD VecT operator-() const { return VecT(-x, -y, -z); }
    RH_HD VecT operator*(T s) const { return VecT(x * s, y * s, z * s); }
    RH_HD VecT operator/(T s) const { return VecT(x / s, y / s, z / s); }
    RH_HD void store(double *p) const { p[0] = (double)x; p[1] = (double)y; p[2] = (double)z; }
};
template <typename T> RH_HD VecT<T> operator*(T s, const VecT<T> &v) { return VecT<T>(s * v.x, s * v.y, s * v.z); }
template <typename T> RH_HD T dot(const VecT<T> &a, const VecT<T> &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
template <typename T> RH_HD T norm(const VecT<T> &a) { return rh_sqrt_t((a.x * a.x + a.y * a.y) + a.z * a.z); }
template <typename T> RH_HD VecT<T> normalize(const VecT<T> &a) { return (T(1) / norm(a)) * a; }
template <typename T> RH_HD VecT<T> cross(const VecT<T> &a, const VecT<T> &b)
{
    return VecT<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
typedef VecT<double> Vec;

template <typename T>
struct Vec2T {
    T x, y;
};
typedef Vec2T<double> Vec2;


// ---- plane.jl:33-57 ----
template <typename T>
RH_HD bool fit_plane_t(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out)
{
    typedef VecT<T> V;
    const V p1(p), p2(p + 3), p3(p + 6);
    const V crossv = normalize(cross(p2 - p1, p3 - p1));
    if (norm(crossv) < prm.collin_threshold) return false;
    const double thr = prm.cos_alpha[RH_PLANE];
    bool same = true, opposite = true;
    for (int i = 0; i < lp; i++) {
        const T dotp = dot(crossv, normalize(V(n + 3 * i)));
        same = same && (dotp > thr);
        opposite = opposite && (dotp < -thr);
    }
    if (!same && !opposite) return false;
    out->kind = RH_PLANE;
    p1.store(out->v);
    (same ? crossv : T(-1) * crossv).store(out->v + 3);
    return true;
}

// ---- sphere.jl:29-75 ----
template <typename T>
RH_HD void fit2pointsphere_t(const double *v, const double *n, const rh_params &prm, VecT<T> *center, T *radius)
{
    typedef VecT<T> V;
    const V v1(v), v2(v + 3), n1(n), n2raw(n + 3);
    const V n1n = normalize(n1), n2n = normalize(n2raw);
    if (rh_fabs_t(dot(n1n, n2n)) > prm.cos_parallelthr) {
        *center = (v1 + v2) / T(2);
        *radius = norm(*center - v1);
        return;
    }
    const V g = v2 - v1;
    const V h = cross(n2n, g);
    const V k = cross(n2n, n1n);
    const T nk = norm(k), nh = norm(h);
    if (nk < prm.sphere_par || nh < prm.sphere_par) {
        const V n2 = cross(n2n, cross(n1n, n2n));
        const V n1b = cross(n1n, cross(n2n, n1n));
        const V c1 = v1 + (dot(v2 - v1, n2) / dot(n1, n2)) * n1;
        const V c2 = v2 + (dot(v1 - v2, n1b) / dot(n2raw, n1b)) * n2raw;
        *center = (c1 + c2) / T(2);
        *radius = (norm(v1 - *center) + norm(v1 - *center)) / T(2);
    } else if (dot(h, k) > 0) {
        *center = v1 + (nh / nk) * n1n;
        *radius = norm(*center - v1);
    } else {
        *center = v1 - (nh / nk) * n1n;
        *radius = norm(*center - v1);
    }
}
RH_HD void fit2pointsphere(const double *v, const double *n, const rh_params &prm, Vec *center, double *radius)
{
    fit2pointsphere_t<double>(v, n, prm, center, radius);
}

// ---- sphere.jl:87-114 ----
template <typename T>
RH_HD bool fit_sphere_t(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out)
{
    typedef VecT<T> V;
    V center;
    T radius;
    fit2pointsphere_t<T>(p, n, prm, &center, &radius);
    const double thr = prm.cos_alpha[RH_SPHERE], eps = prm.eps[RH_SPHERE];
    bool vert = true, same = true, opposite = true;
    for (int i = 0; i < lp; i++) {
        const V pi(p + 3 * i);
        vert = vert && (rh_fabs_t(norm(pi - center) - radius) < eps);
        const T dotp = dot(normalize(pi - center), normalize(V(n + 3 * i)));
        same = same && (dotp > thr);
        opposite = opposite && (dotp < -thr);
    }
    if (!vert || (!same && !opposite)) return false;
    out->kind = RH_SPHERE;
    out->outwards = same ? 1 : 0;
    center.store(out->v);
    out->v[3] = (double)radius;
    return true;
}

// ---- cylinder.jl:46-59 (project2plane), :61-85 (projectto2d), :87-101 ----
template <typename T>
RH_HD VecT<T> cyl_project2plane(const VecT<T> &n, const VecT<T> &w) { return w + n * (dot(-n, w) / dot(n, n)); }

template <typename T>
RH_HD Vec2T<T> cyl_projectto2d(const VecT<T> &xa, const VecT<T> &ya, const VecT<T> &za, const VecT<T> &p1)
{
    const T xx = xa.x, xy = xa.y, xz = xa.z, yx = ya.x, yy = ya.y, yz = ya.z;
    const T zx = za.x, zy = za.y, zz = za.z, px = p1.x, py = p1.y, pz = p1.z;
    Vec2T<T> r;
    r.x = -((-(pz * yy * zx) + py * yz * zx + pz * yx * zy - px * yz * zy - py * yx * zz + px * yy * zz) /
            (xz * yy * zx - xy * yz * zx - xz * yx * zy + xx * yz * zy + xy * yx * zz - xx * yy * zz));
    r.y = -((pz * xy * zx - py * xz * zx - pz * xx * zy + px * xz * zy + py * xx * zz - px * xy * zz) /
            (xz * yy * zx - xy * yz * zx - xz * yx * zy + xx * yz * zy + xy * yx * zz - xx * yy * zz));
    return r;
}

// ---- cylinder.jl:34-125 ----
template <typename T>
RH_HD bool fit2pointcylinder_t(const double *p, const double *n, const rh_params &prm, VecT<T> *axis, VecT<T> *center, T *radius,
                               bool *outw)
{
    typedef VecT<T> V;
    typedef Vec2T<T> V2;
    const V p1(p), p2(p + 3), n1(n), n2(n + 3);
    if (rh_fabs_t(dot(n1, n2)) > prm.cos_parallelthr) return false;
    const V an = normalize(cross(n1, n2));
    const V xax = normalize(cyl_project2plane(an, p1));
    const V yax = normalize(cross(an, xax));
    const V2 a = cyl_projectto2d(xax, yax, an, cyl_project2plane(an, p1));
    const V2 b = cyl_projectto2d(xax, yax, an, cyl_project2plane(an, p1 + n1));
    const V2 c = cyl_projectto2d(xax, yax, an, cyl_project2plane(an, p2));
    const V2 d = cyl_projectto2d(xax, yax, an, cyl_project2plane(an, p2 + n2));
    const V2 amb = { a.x - b.x, a.y - b.y }, cmd = { c.x - d.x, c.y - d.y };
    const T d1 = a.x * b.y - a.y * b.x;       // det([a'; b'])
    const T d2 = c.x * d.y - c.y * d.x;
    const T d3 = amb.x * cmd.y - amb.y * cmd.x;
    const V2 ic = { (d1 * cmd.x - d2 * amb.x) / d3, (d1 * cmd.y - d2 * amb.y) / d3 };
    const V cc = ic.x * xax + ic.y * yax;
    const T r1 = norm((p1 - cc) - an * dot(an, p1 - cc));
    const T r2 = norm((p2 - cc) - an * dot(an, p2 - cc));
    *axis = an;
    *center = cc;
    *radius = (r1 + r2) / T(2);
    *outw = ((b.x - a.x) * (a.x - ic.x) + (b.y - a.y) * (a.y - ic.y)) > 0;
    return true;
}
RH_HD bool fit2pointcylinder(const double *p, const double *n, const rh_params &prm, Vec *axis, Vec *center, double *radius,
                       bool *outw)
{
    return fit2pointcylinder_t<double>(p, n, prm, axis, center, radius, outw);
}

// ---- cylinder.jl:135-168 ----
template <typename T>
RH_HD bool fit_cylinder_t(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out)
{
    typedef VecT<T> V;
    V axis, center;
    T radius;
    bool outw;
    if (!fit2pointcylinder_t<T>(p, n, prm, &axis, &center, &radius, &outw)) return false;
    const double thr = prm.cos_alpha[RH_CYLINDER], eps = prm.eps[RH_CYLINDER];
    bool vert = true, same = true, opposite = true;
    for (int i = 0; i < lp; i++) {
        const V pi(p + 3 * i);
        const V cn = (pi - axis * dot(axis, pi - center)) - center;
        vert = vert && (rh_fabs_t(norm(cn) - radius) < eps);
        const T dotp = dot(normalize(cn), V(n + 3 * i));
        same = same && (dotp > thr);
        opposite = opposite && (dotp < -thr);
    }
    if (!vert || (!same && !opposite)) return false;
    out->kind = RH_CYLINDER;
    out->outwards = same ? 1 : 0;
    axis.store(out->v);
    center.store(out->v + 3);
    out->v[6] = (double)radius;
    return true;
}

// the binary64 entry points (every caller of rounds 1-3) and the dispatch a cloud's element type makes
RH_HD bool fit_plane(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_plane_t<double>(p, n, lp, prm, out); }
RH_HD bool fit_sphere(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_sphere_t<double>(p, n, lp, prm, out); }
RH_HD bool fit_cylinder(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_cylinder_t<double>(p, n, lp, prm, out); }
RH_HD bool fit_plane32(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_plane_t<float>(p, n, lp, prm, out); }
RH_HD bool fit_sphere32(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_sphere_t<float>(p, n, lp, prm, out); }
RH_HD bool fit_cylinder32(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_cylinder_t<float>(p, n, lp, prm, out); }


// cos(-opang/2), sin(-opang/2) of a cone (rodriguesrad(rot_ax, -cone.opang/2): cone.jl:76,
// utilities.jl:21-22) with the deterministic det_math.h kernels: same bits on host and device
RH_HD void cone_finalize(rh_shape *s)
{
    const double th = -s->v[6] / 2;
    s->v[7] = rh_cos(th);
    s->v[8] = rh_sin(th);
}
// ... of a Float32 cone (opang a binary32 number): Float32 cos / sin of the Float32 angle.  Julia's Float32 kernels evaluate a
// double-precision polynomial of the widened argument and round once (base/special/trig.jl, after msun's k_cosf.c / k_sinf.c);
// here: the deterministic double kernels on the widened angle, rounded once -- the same bits but for the one-in-2^29 double
// rounding case
RH_HD void cone_finalize32(rh_shape *s)
{
    const float th = -(float)s->v[6] / 2.0f;
    s->v[7] = (double)(float)rh_cos((double)th);
    s->v[8] = (double)(float)rh_sin((double)th);
}
RH_HD double rh_acos_t(double x) { return rh_acos(x); }
RH_HD float rh_acos_t(float x) { return (float)rh_acos((double)x); }   // (rounded once from the deterministic double kernel)
template <typename T> struct FitEps;
template <> struct FitEps<double> { static RH_HD double eps() { return 2.220446049250313e-16; } static RH_HD double tiny() { return 1e-300; } static RH_HD double jac() { return 2.2e-16; } };
template <> struct FitEps<float> { static RH_HD float eps() { return 1.1920928955078125e-07f; } static RH_HD float tiny() { return 1e-37f; } static RH_HD float jac() { return 1.2e-7f; } };

// ---- LinearAlgebra stand-ins for cone.jl:44,48,50 (T = the cloud's element type: rank() and \ of a Matrix{Float32} on a
// Float32 cloud -- LAPACK's single-precision SVD and LU upstream; like the Float64 instantiation a restatement that nothing
// of the reference can pin beyond its accept / reject fixtures) ----
// singular values via one-sided Jacobi on the columns (rows x cols, rows >= cols)
template <typename T>
RH_HD void jacobi_svals(T *M, int rows, int cols, T *sv)
{
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p + 1 < cols; p++)
            for (int q = p + 1; q < cols; q++) {
                T a = 0, b = 0, g = 0;
                for (int i = 0; i < rows; i++) {
                    a += M[i * cols + p] * M[i * cols + p];
                    b += M[i * cols + q] * M[i * cols + q];
                    g += M[i * cols + p] * M[i * cols + q];
                }
                if (g == T(0) || rh_fabs_t(g) <= FitEps<T>::tiny() + FitEps<T>::jac() * rh_sqrt_t(a * b)) continue;
                rotated = true;
                const T zeta = (b - a) / (2 * g);
                const T t = (zeta >= 0 ? T(1) : T(-1)) / (rh_fabs_t(zeta) + rh_sqrt_t(1 + zeta * zeta));
                const T cs = 1 / rh_sqrt_t(1 + t * t), sn = cs * t;
                for (int i = 0; i < rows; i++) {
                    const T mp = M[i * cols + p], mq = M[i * cols + q];
                    M[i * cols + p] = cs * mp - sn * mq;
                    M[i * cols + q] = sn * mp + cs * mq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < cols; j++) {
        T a = 0;
        for (int i = 0; i < rows; i++) a += M[i * cols + j] * M[i * cols + j];
        sv[j] = rh_sqrt_t(a);
    }
}

template <typename T>
RH_HD int matrix_rank(const T *A, int m, int n)   // rank(A): count(svdvals .> min(m,n)*eps*max)
{
    T M[16], sv[4];
    int rows, cols;
    if (m >= n) {
        rows = m; cols = n;
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[i * cols + j] = A[i * n + j];
    } else {
        rows = n; cols = m;
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) M[j * cols + i] = A[i * n + j];
    }
    jacobi_svals<T>(M, rows, cols, sv);
    T smax = 0;
    for (int j = 0; j < cols; j++) smax = sv[j] > smax ? sv[j] : smax;
    const T tol = (T)(m < n ? m : n) * FitEps<T>::eps() * smax;
    int r = 0;
    for (int j = 0; j < cols; j++) r += sv[j] > tol;
    return r;
}

template <typename T>
RH_HD bool lu_solve3(const T A0[9], const T b0[3], T x[3])   // A \ b, partial pivoting
{
    T A[9], b[3];
    for (int i = 0; i < 9; i++) A[i] = A0[i];
    for (int i = 0; i < 3; i++) b[i] = b0[i];
    for (int k = 0; k < 3; k++) {
        int piv = k;
        T best = rh_fabs_t(A[k * 3 + k]);
        for (int i = k + 1; i < 3; i++)
            if (rh_fabs_t(A[i * 3 + k]) > best) { best = rh_fabs_t(A[i * 3 + k]); piv = i; }
        if (best == T(0)) return false;
        if (piv != k) {
            for (int j = 0; j < 3; j++) { const T t = A[k * 3 + j]; A[k * 3 + j] = A[piv * 3 + j]; A[piv * 3 + j] = t; }
            const T t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < 3; i++) {
            const T l = A[i * 3 + k] / A[k * 3 + k];
            A[i * 3 + k] = l;
            for (int j = k + 1; j < 3; j++) A[i * 3 + j] -= l * A[k * 3 + j];
            b[i] -= l * b[k];
        }
    }
    for (int i = 2; i >= 0; i--) {
        T acc = b[i];
        for (int j = i + 1; j < 3; j++) acc -= A[i * 3 + j] * x[j];
        x[i] = acc / A[i * 3 + i];
    }
    return true;
}

// ---- cone.jl:68-85 (host twin of the device test, used by validatecone) ----
template <typename T>
RH_HD void project2cone(const rh_shape &cone, const VecT<T> &p, T *dist, VecT<T> *cn)
{
    const VecT<T> apex(cone.v), axis(cone.v + 3);
    const VecT<T> to_point = apex - p;
    const VecT<T> to_pointn = normalize(to_point);
    const VecT<T> rot_ax = normalize(cross(axis, to_pointn));
    const VecT<T> comp_n = normalize(cross(axis, rot_ax));
    const VecT<T> v = normalize(rot_ax);   // rodriguesrad re-normalizes (utilities.jl:62)
    const T c = (T)cone.v[7], s = (T)cone.v[8];
    const T e[3] = { v.x, v.y, v.z };
    T R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const T nn = e[i] * e[j];
            R[i][j] = nn + c * ((i == j ? T(1) : T(0)) - nn);
        }
    R[0][1] -= s * e[2]; R[0][2] += s * e[1];   // pluscrossprod!: utilities.jl:32-43
    R[1][0] += s * e[2]; R[1][2] -= s * e[0];
    R[2][0] -= s * e[1]; R[2][1] += s * e[0];
    const VecT<T> rc((R[0][0] * comp_n.x + R[0][1] * comp_n.y) + R[0][2] * comp_n.z,
                     (R[1][0] * comp_n.x + R[1][1] * comp_n.y) + R[1][2] * comp_n.z,
                     (R[2][0] * comp_n.x + R[2][1] * comp_n.y) + R[2][2] * comp_n.z);
    *cn = normalize(rc);
    *dist = dot(-*cn, -to_point);
}

template <typename T> RH_HD T clamp_unit(T x) { return x < T(-1) ? T(-1) : (x > T(1) ? T(1) : x); }

// ---- cone.jl:39-61 ----
template <typename T>
RH_HD bool fit3pointcone_t(const double *p, const double *n, rh_shape *cone)
{
    typedef VecT<T> V;
    T r[9], rv[12], ds[3], ap[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i * 3 + j] = (T)n[3 * i + j];
    if (matrix_rank<T>(r, 3, 3) != 3) return false;
    for (int i = 0; i < 3; i++) ds[i] = dot(V(p + 3 * i), V(n + 3 * i));
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) rv[i * 4 + j] = r[i * 3 + j];
        rv[i * 4 + 3] = -1 * ds[i];
    }
    if (matrix_rank<T>(rv, 3, 4) != 3) return false;
    if (!lu_solve3<T>(r, ds, ap)) return false;
    const V apex(ap[0], ap[1], ap[2]);
    V a3[3];
    for (int i = 0; i < 3; i++) {
        const V d = V(p + 3 * i) - apex;
        a3[i] = apex + d / norm(d);
    }
    V ax = normalize(cross(a3[1] - a3[0], a3[2] - a3[0]));
    const V midp = ((a3[0] + a3[1]) + a3[2]) / T(3);
    const V dirv = normalize(midp - apex);
    if (dot(ax, dirv) < 0) ax = T(-1) * ax;
    T ang[3];
    for (int i = 0; i < 3; i++) ang[i] = rh_acos_t(clamp_unit<T>(dot(normalize(V(p + 3 * i) - apex), ax)));
    for (int i = 0; i < 10; i++) cone->v[i] = 0.0;
    cone->kind = RH_CONE;
    cone->outwards = 1;
    apex.store(cone->v);
    ax.store(cone->v + 3);
    cone->v[6] = (double)(2 * ((ang[0] + ang[1]) + ang[2]) / 3);
    if (sizeof(T) == sizeof(float)) cone_finalize32(cone); else cone_finalize(cone);
    return true;
}
RH_HD bool fit3pointcone(const double *p, const double *n, rh_shape *cone) { return fit3pointcone_t<double>(p, n, cone); }

// ---- cone.jl:87-115, 123-128 ----
template <typename T>
RH_HD bool fit_cone_t(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out)
{
    typedef VecT<T> V;
    if (lp > 16) return false;
    rh_shape cone;
    if (!fit3pointcone_t<T>(p, n, &cone)) return false;
    T dist[16];
    V cn[16];
    for (int i = 0; i < lp; i++) project2cone<T>(cone, V(p + 3 * i), &dist[i], &cn[i]);
    for (int i = 0; i < lp; i++)
        if (dist[i] > prm.eps[RH_CONE]) return false;   // no abs in the reference (cone.jl:93); a Float32 distance is promoted exactly
    if (cone.v[6] < prm.minconeopang) return false;
    const double thr = prm.cos_alpha[RH_CONE];
    bool same = true, opposite = true;
    for (int i = 0; i < lp; i++) {
        const T dotp = dot(cn[i], V(n + 3 * i));
        same = same && (dotp > thr);
        opposite = opposite && (dotp < -thr);
    }
    if (!same && !opposite) return false;
    *out = cone;
    out->outwards = same ? 1 : 0;
    return true;
}
RH_HD bool fit_cone(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_cone_t<double>(p, n, lp, prm, out); }
RH_HD bool fit_cone32(const double *p, const double *n, int lp, const rh_params &prm, rh_shape *out) { return fit_cone_t<float>(p, n, lp, prm, out); }


// ---- per-minimal-set random stream (sampling_streams = 1) ------------------------------------
// One splitmix64 stream per (iteration k, minimal set j), a pure function of (seed, k, j): every
// set can be drawn independently -- on the device, in any order -- with the same result.
RH_HD uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
RH_HD uint64_t set_stream_init(uint64_t seed, uint64_t k, uint64_t j)
{
    return mix64(seed + k * 0xD1B54A32D192ED03ULL) ^ mix64(j * 0x8CB92BA72F3D8DD7ULL + 0x2545F4914F6CDD1DULL);
}
RH_HD uint64_t set_stream_next(uint64_t *x)
{
    *x += 0x9E3779B97F4A7C15ULL;
    return mix64(*x);
}
RH_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
}
// rand(1:n) = 1 + floor(u * n / 2^64)
RH_HD int64_t set_stream_range(uint64_t *x, int64_t n) { return 1 + (int64_t)mulhi64(set_stream_next(x), (uint64_t)n); }

}  // namespace rhfit

namespace rhfit {

// samplepointcloud4! (src/fitting.jl:383-430) on the root cell, drawing from a per-set stream.
// En provides test(i0) and select(rank) over the enabled bits.  Returns false for the
// reference's (false, 0) / (false, 1) outcomes.  *ndraws counts rand() calls.
// DN > 0: compile-time drawN (loops unroll, the small arrays stay in registers on the device)
template <int DN, class En>
RH_HD bool sample_minimal_set(En &en, int64_t n, int64_t n_enabled, int drawN_rt, uint64_t *x, int64_t *sd,
                              uint32_t *ndraws, bool *gave_up)
{
    const int drawN = DN > 0 ? DN : drawN_rt;
    if (n_enabled <= 0) return false;   // the reference would spin forever at fitting.jl:393
    int64_t r1 = set_stream_range(x, n);
    uint32_t nd = 1;
    while (!en.test(r1 - 1)) {
        r1 = set_stream_range(x, n);
        if (++nd > (1u << 24)) { *gave_up = true; *ndraws += nd; return false; }
    }
    *ndraws += nd;
    if (n_enabled < drawN) return false;
    sd[0] = r1;
#pragma unroll
    for (int q = 1; q < drawN; q++) {
        int64_t pick = en.select(set_stream_range(x, n_enabled));
        ++*ndraws;
        if (pick == sd[0]) {   // one redraw: fitting.jl:416-419
            pick = en.select(set_stream_range(x, n_enabled));
            ++*ndraws;
        }
        sd[q] = pick;
    }
    bool distinct = true;   // allisdifferent: utilities.jl:285-295
#pragma unroll
    for (int a = 1; a < drawN; a++)
#pragma unroll
        for (int b = 0; b < a; b++) distinct = distinct && (sd[a] != sd[b]);
    return distinct;
}

}  // namespace rhfit

namespace rhfit {

// ---- level-weighted sampling on a linear (Morton) octree: octree_sampling = 1 -------------------
// View over the octree arrays (host or device pointers).  Points are sorted by 63-bit Morton code
// in the cloud's bounding cube; the level-l cell of a point is the range of codes sharing its top
// 3(l-1) bits, so cell populations are rank differences on the Morton-ordered enabled bits.
struct OctView {
    const uint64_t *code;     // [n] sorted codes
    const int32_t *perm;      // [n] Morton position -> original index0
    const int32_t *pos;       // [n] original index0 -> Morton position
    const uint64_t *men;      // enabled bits in Morton order
    const int32_t *prefix;    // exclusive popcount prefix per word of men; [nwords] = total
    int64_t n, nwords;
    int depth;
    // Optional (device sampler): tab[k] = lower_bound(k << tab_shift), k = 0 .. 8^(tab_level - 1) -- the first Morton
    // position of every level-tab_level cell.  A level-l cell, l <= tab_level, is a run of such cells, so its bounds are
    // two table entries; a deeper cell is searched for inside its level-tab_level ancestor.  The results are those of the
    // plain binary searches (a lower bound is unique), found in 2-4 dependent loads instead of ~50.
    const int32_t *tab = nullptr;
    int tab_level = 0;
    const uint64_t *code_o = nullptr;   // optional (device sampler): the codes again, in ORIGINAL point order (code[pos[i]] in one load)

    // lower_bound(key) on [lo, hi]: the answer is known to lie in that range.  8-ary rounds (seven independent loads)
    RH_HD int64_t lower_bound_in(uint64_t key, int64_t lo, int64_t hi) const
    {
        while (hi - lo > 7) {
            const int64_t step = (hi - lo + 7) >> 3;   // pivots lo + i * step - 1, i = 1..7: code[pv] < key  <=>  answer > pv
            // (the seven loads are unconditional, on clamped addresses, so that they are in flight together)
            uint64_t cv[7];
#pragma unroll
            for (int i = 1; i < 8; i++) {
                const int64_t pv = lo + i * step - 1;
                cv[i - 1] = code[pv < hi ? pv : hi - 1];
            }
            int k = 0;   // codes ascend: the pivots below the answer come first
#pragma unroll
            for (int i = 1; i < 8; i++) k += ((lo + i * step - 1 < hi) & (cv[i - 1] < key)) ? 1 : 0;
            const int64_t nlo = lo + k * step;
            hi = (k < 7 && nlo + step - 1 < hi) ? nlo + step - 1 : hi;
            lo = nlo < hi ? nlo : hi;
        }
        // at most seven candidates left: the answer is lo + the number of them below the key
        uint64_t cv[7];
#pragma unroll
        for (int i = 0; i < 7; i++) cv[i] = code[lo + i < hi ? lo + i : (hi > 0 ? hi - 1 : 0)];
        int k = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) k += ((lo + i < hi) & (cv[i] < key)) ? 1 : 0;
        return lo + k;
    }
    // [lo, hi) of the level-`level` cell of the point with code cd (needs the directory): a cell deeper than the
    // directory's level is looked for inside its ancestor there, both ends at once
    RH_HD void cell_bounds_code(int level, uint64_t cd, int64_t *lo_out, int64_t *hi_out) const
    {
        const int shift = 3 * (21 - (level - 1));
        if (shift >= 63) { *lo_out = 0; *hi_out = n; return; }
        const uint64_t key = cd >> shift;
        const bool last = (((key + 1) << shift) >> shift) != key + 1;
        const int tshift = 3 * (21 - (tab_level - 1));
        if (level <= tab_level) {
            const int up = shift - tshift;
            const int32_t a = tab[key << up], b = tab[last ? (key << up) : ((key + 1) << up)];
            *lo_out = a;
            *hi_out = last ? n : b;
            return;
        }
        const uint64_t anc = cd >> tshift;
        const int64_t a = tab[anc], b = tab[anc + 1];
        *lo_out = lower_bound_in(key << shift, a, b);
        *hi_out = last ? n : lower_bound_in((key + 1) << shift, *lo_out, b);
    }
    // [lo, hi) of the level-`level` cell that holds Morton position q0
    RH_HD void cell_bounds(int level, int64_t q0, int64_t *lo_out, int64_t *hi_out) const
    {
        const int shift = 3 * (21 - (level - 1));
        if (shift >= 63) { *lo_out = 0; *hi_out = n; return; }
        const uint64_t key = code[q0] >> shift;
        const bool last = (((key + 1) << shift) >> shift) != key + 1;   // the cell reaches the end of the code space
        if (tab == nullptr) {
            *lo_out = lower_bound(key << shift);
            *hi_out = last ? n : lower_bound((key + 1) << shift);
            return;
        }
        const int tshift = 3 * (21 - (tab_level - 1));
        if (level <= tab_level) {
            const int up = shift - tshift;
            *lo_out = tab[key << up];
            *hi_out = last ? n : tab[(key + 1) << up];
            return;
        }
        // deeper than the table: inside the ancestor cell [a, b); the cell starts at or before q0 and ends after it
        const uint64_t anc = code[q0] >> tshift;
        const int64_t a = tab[anc], b = tab[anc + 1];
        *lo_out = lower_bound_in(key << shift, a, q0);
        *hi_out = last ? n : lower_bound_in((key + 1) << shift, q0 + 1, b);
    }

    RH_HD int64_t lower_bound(uint64_t key) const
    {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (code[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
    RH_HD int64_t rank(int64_t i) const
    {
        if (i >= n) return prefix[nwords];
        uint64_t m = men[i >> 6] & ((1ULL << (i & 63)) - 1ULL);
        int c = 0;
        while (m) { m &= m - 1; c++; }
        return prefix[i >> 6] + c;
    }
    RH_HD int64_t select(int64_t r) const
    {
        int64_t lo = 0, hi = nwords;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (prefix[mid] < r) lo = mid; else hi = mid;
        }
        uint64_t m = men[lo];
        for (int64_t t = 1; t < r - prefix[lo]; t++) m &= m - 1;
        int b = 0;
        while (!((m >> b) & 1ULL)) b++;
        return lo * 64 + b;
    }
    // select(r) when the r-th enabled point is known to lie in the words [wlo, whi] and prefix[wlo] < r: the same word
    // (the last one whose exclusive prefix is below r), found by 8-ary rounds inside the bracket
    RH_HD int64_t select_in(int64_t r, int64_t wlo, int64_t whi) const
    {
        int64_t lo = wlo, hi = whi;   // answer in [lo, hi]
        while (hi - lo >= 8) {
            const int64_t step = (hi - lo + 8) >> 3;
            int32_t pv7[7];
#pragma unroll
            for (int i = 1; i < 8; i++) {
                const int64_t pv = lo + i * step;
                pv7[i - 1] = prefix[pv <= hi ? pv : hi];
            }
            int k = 0;   // prefix is non-decreasing: the pivots with prefix < r are the first k
#pragma unroll
            for (int i = 1; i < 8; i++) k += ((lo + i * step <= hi) & ((int64_t)pv7[i - 1] < r)) ? 1 : 0;
            lo += k * step;
            hi = hi < lo + step - 1 ? hi : lo + step - 1;
        }
        {
            int32_t pv7[7];
#pragma unroll
            for (int i = 1; i < 8; i++) pv7[i - 1] = prefix[lo + i <= hi ? lo + i : hi];
            int k = 0;
#pragma unroll
            for (int i = 1; i < 8; i++) k += ((lo + i <= hi) & ((int64_t)pv7[i - 1] < r)) ? 1 : 0;
            lo += k;
        }
        uint64_t m = men[lo];
        for (int64_t t = 1; t < r - prefix[lo]; t++) m &= m - 1;
#if defined(__HIP_DEVICE_COMPILE__)
        return lo * 64 + __builtin_ctzll(m);
#else
        int b = 0;
        while (!((m >> b) & 1ULL)) b++;
        return lo * 64 + b;
#endif
    }
    // position of the (k + 1)-th set bit of m (k < popcount): five halving steps instead of k clearings
    static RH_HD int select_bit(uint64_t m, int k)
    {
        int pos = 0;
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) {
            const uint64_t lowmask = w == 32 ? 0xffffffffULL : ((1ULL << w) - 1ULL);
            const uint64_t lowpart = (m >> pos) & lowmask;
            int c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            c = __builtin_popcountll(lowpart);
#else
            for (uint64_t t = lowpart; t; t &= t - 1) c++;
#endif
            if (k >= c) { k -= c; pos += w; }
        }
        return pos;
    }
    // select_in for K ranks at once, in lockstep: the K searches' loads are in flight together (a round is the same
    // 8-ary step whatever the width of the bracket, down to a single word)
    template <int K>
    RH_HD void select_in_many(int64_t (&r)[K], int64_t wlo, int64_t whi) const
    {
        int64_t lo[K], hi[K];
#pragma unroll
        for (int q = 0; q < K; q++) { lo[q] = wlo; hi[q] = whi; }
        for (;;) {
            bool more = false;
#pragma unroll
            for (int q = 0; q < K; q++) more = more || hi[q] > lo[q];
            if (!more) break;
            int32_t pv[K][7];
#pragma unroll
            for (int q = 0; q < K; q++) {
                const int64_t step = (hi[q] - lo[q] + 8) >> 3;
#pragma unroll
                for (int i = 1; i < 8; i++) {
                    const int64_t p = lo[q] + i * step;
                    pv[q][i - 1] = prefix[p <= hi[q] ? p : hi[q]];
                }
            }
#pragma unroll
            for (int q = 0; q < K; q++) {
                const int64_t step = (hi[q] - lo[q] + 8) >> 3;
                int k = 0;
#pragma unroll
                for (int i = 1; i < 8; i++) k += ((lo[q] + i * step <= hi[q]) & ((int64_t)pv[q][i - 1] < r[q])) ? 1 : 0;
                lo[q] += k * step;
                hi[q] = hi[q] < lo[q] + step - 1 ? hi[q] : lo[q] + step - 1;
            }
        }
        uint64_t m[K];
        int32_t pf[K];
#pragma unroll
        for (int q = 0; q < K; q++) { m[q] = men[lo[q]]; pf[q] = prefix[lo[q]]; }
#pragma unroll
        for (int q = 0; q < K; q++) r[q] = lo[q] * 64 + select_bit(m[q], (int)(r[q] - 1 - pf[q]));
    }
};

#if defined(RH_OCT_TIMING)
static __device__ unsigned long long rh_oct_t[16];
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(RH_OCT_TIMING)
#define RH_OCT_T(i) do { const unsigned long long t_ = wall_clock64(); atomicAdd(&rhfit::rh_oct_t[i], t_ - t0_); t0_ = t_; } while (0)
#define RH_OCT_T0 unsigned long long t0_ = wall_clock64()
#else
#define RH_OCT_T(i) do { } while (0)
#define RH_OCT_T0 do { } while (0)
#endif
template <int DN, class En>
RH_HD bool sample_minimal_set_octree(En &en, const OctView &oc, const double *P, int64_t n, int64_t n_enabled,
                                     int drawN_rt, uint64_t *x, int64_t *sd, uint32_t *ndraws, bool *gave_up, int *level_out)
{
    const int drawN = DN > 0 ? DN : drawN_rt;
    *level_out = 1;
    if (n_enabled <= 0) return false;
    RH_OCT_T0;
    int64_t r1 = 0;
    uint32_t nd = 0;
    uint64_t cd = 0;
    if (oc.code_o != nullptr) {
        // Device form of the loop below: the stream advances by a constant per draw, so draw k is a pure function of
        // (x0, k) -- four draws are tested per round, each with its Morton code fetched alongside (independent loads
        // instead of a chain); the accepted index and the draws consumed are those of the one-at-a-time loop.
        const uint64_t G = 0x9E3779B97F4A7C15ULL, x0 = *x;
        for (;;) {
            int64_t r[4];
            bool e[4];
            uint64_t c4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r[i] = 1 + (int64_t)mulhi64(mix64(x0 + (uint64_t)(nd + 1 + i) * G), (uint64_t)n);
                e[i] = en.test(r[i] - 1);
                c4[i] = oc.code_o[r[i] - 1];
            }
            int hit = -1;
#pragma unroll
            for (int i = 3; i >= 0; i--) hit = e[i] ? i : hit;
            if (hit >= 0) {
                nd += (uint32_t)hit + 1;
#pragma unroll
                for (int i = 0; i < 4; i++) if (i == hit) { r1 = r[i]; cd = c4[i]; }
                break;
            }
            nd += 4;
            if (nd > (1u << 24)) { *gave_up = true; *x = x0 + (uint64_t)nd * G; *ndraws += nd; return false; }
        }
        *x = x0 + (uint64_t)nd * G;
    } else {
        r1 = set_stream_range(x, n);
        nd = 1;
        while (!en.test(r1 - 1)) {
            r1 = set_stream_range(x, n);
            if (++nd > (1u << 24)) { *gave_up = true; *ndraws += nd; return false; }
        }
    }
    RH_OCT_T(0);
    const double u = (double)(set_stream_next(x) >> 11) * (1.0 / 9007199254740992.0);
    *ndraws += nd + 1;
    int level = oc.depth;
    double acc = 0;
    for (int l = 0; l < oc.depth; l++) {
        acc += P[l];
        if (u < acc) { level = l + 1; break; }
    }
    *level_out = level;
    int64_t lo, hi;
    if (oc.code_o != nullptr && oc.tab != nullptr) oc.cell_bounds_code(level, cd, &lo, &hi);
    else oc.cell_bounds(level, (int64_t)oc.pos[r1 - 1], &lo, &hi);
    RH_OCT_T(1);
    const int64_t base = oc.rank(lo), ne = oc.rank(hi) - base;
    RH_OCT_T(2);
    if (ne < drawN) return false;
    sd[0] = r1;
    if (oc.tab != nullptr) {
        // Device form.  The ranks of all the other points are drawn first, as if no redraw happened, and looked up
        // together (independent searches, bracketed by the cell's words); a pick that hits the first point -- the one
        // case that changes the stream (fitting.jl:416-419) -- sends the set through the one-at-a-time loop below.
        const uint64_t xs = *x;
        const int64_t wlo = lo >> 6, whi = (hi - 1) >> 6;
        bool redo = false;
#pragma unroll
        for (int q = 1; q < drawN; q++) sd[q] = base + set_stream_range(x, ne);
        if (DN > 1) {
            int64_t rk[DN > 1 ? DN - 1 : 1];
#pragma unroll
            for (int q = 1; q < DN; q++) rk[q - 1] = sd[q];
            oc.template select_in_many<(DN > 1 ? DN - 1 : 1)>(rk, wlo, whi);
#pragma unroll
            for (int q = 1; q < DN; q++) sd[q] = rk[q - 1];
        } else {
            for (int q = 1; q < drawN; q++) sd[q] = oc.select_in(sd[q], wlo, whi);
        }
        RH_OCT_T(3);
#pragma unroll
        for (int q = 1; q < drawN; q++) { sd[q] = (int64_t)oc.perm[sd[q]] + 1; redo = redo || sd[q] == sd[0]; }
        RH_OCT_T(4);
        if (!redo) *ndraws += (uint32_t)(drawN - 1);
        else {
            *x = xs;
            for (int q = 1; q < drawN; q++) {
                int64_t pick = (int64_t)oc.perm[oc.select_in(base + set_stream_range(x, ne), wlo, whi)] + 1;
                ++*ndraws;
                if (pick == sd[0]) {
                    pick = (int64_t)oc.perm[oc.select_in(base + set_stream_range(x, ne), wlo, whi)] + 1;
                    ++*ndraws;
                }
                sd[q] = pick;
            }
        }
    } else {
#pragma unroll
        for (int q = 1; q < drawN; q++) {
            int64_t pick = (int64_t)oc.perm[oc.select(base + set_stream_range(x, ne))] + 1;
            ++*ndraws;
            if (pick == sd[0]) {
                pick = (int64_t)oc.perm[oc.select(base + set_stream_range(x, ne))] + 1;
                ++*ndraws;
            }
            sd[q] = pick;
        }
    }
    bool distinct = true;
#pragma unroll
    for (int a = 1; a < drawN; a++)
#pragma unroll
        for (int b = 0; b < a; b++) distinct = distinct && (sd[a] != sd[b]);
    return distinct;
}

// level distribution update (src/octree.jl:198-205, x = 9/10) with the initialisation the docs
// describe; left unchanged while no score has been collected or if the formula leaves the simplex
RH_HD void update_level_probs(double *P, const double *sigma, int d)
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

}  // namespace rhfit
