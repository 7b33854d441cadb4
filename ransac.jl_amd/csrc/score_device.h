// score_device.h -- the per-point tests, the conservative group-box tests and the band prefilter of the score
// kernels (shared by kernels.hip and score3.hip; moved here unchanged from kernels.hip).
//
// Numerics contract: IEEE binary64, the reference's operation order, NO fused multiply-add (built with
// -ffp-contract=off), correctly rounded sqrt and divide.  Each per-point test cites the reference function it
// restates (paths under /root/reference/src).
#pragma once

#include "rh_internal.h"

namespace rhdev {

typedef double rh_f64x2 __attribute__((ext_vector_type(2)));

// Wave-uniform, read-only inputs (candidate records written by an EARLIER kernel of the stream) are read through the
// CONSTANT address space: a uniform load from it is a scalar load (s_load_dword*, data in SGPRs, one request per wave)
// that the compiler itself tracks -- it allocates the registers, keeps them live and places the s_waitcnt.
#define RH_CONST_AS __attribute__((address_space(4)))
static __device__ __forceinline__ rh_prep rh_ld_prep_const(const rh_prep *p)
{
    const RH_CONST_AS rh_prep *q = (const RH_CONST_AS rh_prep *)(uintptr_t)p;
    rh_prep o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.f[i] = q->f[i];
    return o;
}

// Each test returns the WAVE's 64-bit result mask (bit l = lane l's point is compatible): the
// two comparisons are balloted separately and ANDed on the scalar unit.
#define WB(cond) __builtin_amdgcn_ballot_w64(cond)

// ------------------------------------------------------------------ tests ----
// plane: compatiblesPlane shapes/plane.jl:114-130 (+ project2plane :82-95), isparallel utilities.jl:115-117
static __device__ __forceinline__ uint64_t test_plane(const rh_prep &P, double px, double py, double pz, double nx,
                                           double ny, double nz, double eps, double cosa)
{
    // The normal half first: a point whose normal is not within alpha of the plane's fails whatever its
    // distance, and in most groups that is every point of the wave (outliers and other primitives' points),
    // so the distance half is skipped with one scalar branch.  Same bits as evaluating both.
    const double dn = (P.f[3] * nx + P.f[4] * ny) + P.f[5] * nz;
    const uint64_t mn = WB(dn > cosa);
    if (mn == 0) return 0;
    const double vx = px - P.f[0], vy = py - P.f[1], vz = pz - P.f[2];
    const double d = (P.f[6] * vx + P.f[7] * vy) + P.f[8] * vz;
    return mn & WB(fabs(d) < eps);
}

// sphere: compatiblesSphere shapes/sphere.jl:144-172.  Inward case: normalize(o-p) = -normalize(p-o)
// and dot(-u, n) = -dot(u, n) exactly (round-to-nearest is odd-symmetric), hence sgn * dot.
static __device__ __forceinline__ uint64_t test_sphere(const rh_prep &P, double px, double py, double pz, double nx,
                                            double ny, double nz, double eps, double cosa)
{
    const double dx = px - P.f[0], dy = py - P.f[1], dz = pz - P.f[2];
    const double nr = sqrt((dx * dx + dy * dy) + dz * dz);
    const uint64_t md = WB(fabs(nr - P.f[3]) < eps);
    if (md == 0) return 0;     // no lane of the wave inside the band: the normal half cannot change that
    const double inv = 1.0 / nr;
    const double ux = inv * dx, uy = inv * dy, uz = inv * dz;
    const double dt = (ux * nx + uy * ny) + uz * nz;
    return WB(P.f[4] * dt > cosa) & md;
}

// cylinder: compatiblesCylinder shapes/cylinder.jl:194-221
static __device__ __forceinline__ uint64_t test_cylinder(const rh_prep &P, double px, double py, double pz, double nx,
                                              double ny, double nz, double eps, double cosa)
{
    const double ax = P.f[0], ay = P.f[1], az = P.f[2];
    const double cx = P.f[3], cy = P.f[4], cz = P.f[5];
    const double tx = px - cx, ty = py - cy, tz = pz - cz;
    const double sd = (ax * tx + ay * ty) + az * tz;
    // curr_norm = p - a*dot(a, p-c) - c
    const double qx = (px - ax * sd) - cx, qy = (py - ay * sd) - cy, qz = (pz - az * sd) - cz;
    const double nr = sqrt((qx * qx + qy * qy) + qz * qz);
    // the reference nests the two tests (cylinder.jl:209-214); their conjunction is the same bit
    const uint64_t md = WB(fabs(nr - P.f[6]) < eps);
    if (md == 0) return 0;
    const double inv = 1.0 / nr;
    const double ux = inv * qx, uy = inv * qy, uz = inv * qz;
    const double dt = (ux * nx + uy * ny) + uz * nz;
    return md & WB(P.f[7] * dt > cosa);
}

// cone: compatiblesCone shapes/cone.jl:132-153, project2cone :68-85,
// rodriguesrad/rodrigues/pluscrossprod! utilities.jl:61-64,19-24,32-43
// (the frame: dist = dot(-current_normal, -to_point), dt = dot(current_normal, n); the audit kernels read them too)
static __device__ __forceinline__ void cone_frame(const rh_prep &P, double px, double py, double pz, double nx,
                                           double ny, double nz, double &dist_out, double &dt_out)
{
    const double ax = P.f[3], ay = P.f[4], az = P.f[5];
    const double c = P.f[6], s = P.f[7];
    // to_point = apex - p; to_pointn = normalize(to_point)
    const double tx = P.f[0] - px, ty = P.f[1] - py, tz = P.f[2] - pz;
    double inv = 1.0 / sqrt((tx * tx + ty * ty) + tz * tz);
    const double tnx = inv * tx, tny = inv * ty, tnz = inv * tz;
    // rot_ax = normalize(cross(axis, to_pointn))
    double kx = ay * tnz - az * tny, ky = az * tnx - ax * tnz, kz = ax * tny - ay * tnx;
    inv = 1.0 / sqrt((kx * kx + ky * ky) + kz * kz);
    const double rx = inv * kx, ry = inv * ky, rz = inv * kz;
    // comp_n = normalize(cross(axis, rot_ax))
    kx = ay * rz - az * ry; ky = az * rx - ax * rz; kz = ax * ry - ay * rx;
    inv = 1.0 / sqrt((kx * kx + ky * ky) + kz * kz);
    const double mx = inv * kx, my = inv * ky, mz = inv * kz;
    // rodriguesrad re-normalizes the axis
    inv = 1.0 / sqrt((rx * rx + ry * ry) + rz * rz);
    const double vx = inv * rx, vy = inv * ry, vz = inv * rz;
    // R = v v' + cos .* (I - v v'), then pluscrossprod!(R, sin, v)
    const double nxx = vx * vx, nxy = vx * vy, nxz = vx * vz, nyy = vy * vy, nyz = vy * vz, nzz = vz * vz;
    const double R00 = nxx + c * (1.0 - nxx);
    double R01 = nxy + c * (0.0 - nxy);
    double R02 = nxz + c * (0.0 - nxz);
    double R10 = R01;
    const double R11 = nyy + c * (1.0 - nyy);
    double R12 = nyz + c * (0.0 - nyz);
    double R20 = R02;
    double R21 = R12;
    const double R22 = nzz + c * (1.0 - nzz);
    R01 -= s * vz; R02 += s * vy;
    R10 += s * vz; R12 -= s * vx;
    R20 -= s * vy; R21 += s * vx;
    // current_normal = normalize(R * comp_n)
    kx = (R00 * mx + R01 * my) + R02 * mz;
    ky = (R10 * mx + R11 * my) + R12 * mz;
    kz = (R20 * mx + R21 * my) + R22 * mz;
    inv = 1.0 / sqrt((kx * kx + ky * ky) + kz * kz);
    const double gx = inv * kx, gy = inv * ky, gz = inv * kz;
    // dist = dot(-current_normal, -to_point)
    dist_out = ((-gx) * (-tx) + (-gy) * (-ty)) + (-gz) * (-tz);
    dt_out = (gx * nx + gy * ny) + gz * nz;
}

static __device__ __forceinline__ uint64_t test_cone(const rh_prep &P, double px, double py, double pz, double nx,
                                          double ny, double nz, double eps, double cosa)
{
    double dist, dt;
    cone_frame(P, px, py, pz, nx, ny, nz, dist, dt);
    return WB(P.f[8] * dt > cosa) & WB(fabs(dist) < eps);
}

template <int KIND>
static __device__ __forceinline__ uint64_t test_point(const rh_prep &P, double px, double py, double pz, double nx,
                                           double ny, double nz, double eps, double cosa)
{
    if (KIND == RH_PLANE) return test_plane(P, px, py, pz, nx, ny, nz, eps, cosa);
    if (KIND == RH_SPHERE) return test_sphere(P, px, py, pz, nx, ny, nz, eps, cosa);
    if (KIND == RH_CYLINDER) return test_cylinder(P, px, py, pz, nx, ny, nz, eps, cosa);
    return test_cone(P, px, py, pz, nx, ny, nz, eps, cosa);
}

static __device__ __forceinline__ uint64_t valid_mask(int64_t base, int64_t s)
{
    const int64_t left = s - base;
    return left >= 64 ? ~0ULL : (left <= 0 ? 0ULL : ((1ULL << left) - 1ULL));
}

// ------------------------------------------------- culled score (groups) ----
// Subset 1 is stored in k-d leaf order (cloud.hip); every 64 consecutive points form a group with an
// axis-aligned box (centre c, half extents h, radius hr = |h|).  A block stages a tile of
// RH_G2_TG groups in LDS.  Per 64-candidate chunk a wave runs
//   stage 1 (lane = candidate): one conservative box test per (candidate, group) -> survivor bits;
//   stage 2 (lane = point):     the exact per-point test only for surviving pairs.
// A pair is skipped ONLY when the box proves that no point of the group can pass the distance
// half of the test, with a slack (1e-9 x coordinate magnitude) that is >= 10^5 x the rounding
// error of the per-point distance, so counts and masks are bit-identical to the brute-force
// kernel.  Every comparison is written so that NaN means "do not skip".
static __device__ __forceinline__ double box_slack(const rh_prep &P, double coord_mag)
{
    return 1e-9 * ((1.0 + coord_mag) + P.f[11]);   // f[11] = sum |f[0..6]| (prep_derived, kernels.hip)
}

// Float32 clouds: the conservative stages still run in binary64 on the exactly converted values (they bound the
// REAL-valued distance), the exact test runs in binary32 (score_device32.h).  Its result differs from the real value by
// the binary32 rounding of ~10-30 operations on terms of the size of the coordinates and parameters -- and, for a
// cylinder, of |a|^2 (p - c) (the axis is used as stored, cylinder.jl:207) -- so the slack is 2^-16 of that magnitude:
// 256 binary32 ulps, ~10x the worst chain (the Float64 slack is 1e-9 of it: 4.5e6 ulps).
template <int KIND>
static __device__ __forceinline__ double box_slack32(const rh_prep &P, double coord_mag)
{
    double m = (1.0 + coord_mag) + P.f[11];
    if (KIND == RH_CYLINDER) m *= fmax(1.0, 2.0 - P.f[9]);   // |a|^2 = 2 - k (prep_derived)
    return 1.52587890625e-05 * m;
}

// The conservative stages (box tests, band prefilter) may fuse: they only have to BOUND the exact test's distance, with
// slacks 10^5 x any rounding, and v_fma_f64 issues at the rate of a multiply or an add -- a dot product is 3
// instructions instead of 5.  (The exact tests never fuse: the library is built with -ffp-contract=off.)
static __device__ __forceinline__ double dot3f(double ax, double ay, double az, double bx, double by, double bz)
{
    return __builtin_fma(az, bz, __builtin_fma(ay, by, ax * bx));
}

template <int KIND, bool F32 = false>
static __device__ __forceinline__ bool box_skip(const rh_prep &P, double cx, double cy, double cz, double hx, double hy,
                                         double hz, double hr, double eps, double slack)
{
    if (KIND == RH_PLANE) {
        // d(p) = dot(o_z, p - point) is affine: over the box it stays within d(c) +- sum |o_z_i| h_i
        const double d = dot3f(P.f[6], P.f[7], P.f[8], cx - P.f[0], cy - P.f[1], cz - P.f[2]);
        const double ext = dot3f(fabs(P.f[6]), fabs(P.f[7]), fabs(P.f[8]), hx, hy, hz);
        return fabs(d) > (ext + eps) + slack;
    }
    if (KIND == RH_SPHERE) {
        // |p - o| lies between the min and max distance from o to the box; compared as squares (no sqrt:
        // the thresholds carry `slack`, 10^6 x the rounding of the squares)
        const double ax = fabs(cx - P.f[0]), ay = fabs(cy - P.f[1]), az = fabs(cz - P.f[2]);
        const double nx = fmax(ax - hx, 0.0), ny = fmax(ay - hy, 0.0), nz = fmax(az - hz, 0.0);
        const double fx = ax + hx, fy = ay + hy, fz = az + hz;
        const double dmin2 = dot3f(nx, ny, nz, nx, ny, nz), dmax2 = dot3f(fx, fy, fz, fx, fy, fz);
        const double A = (P.f[3] + eps) + slack, B = (P.f[3] - eps) - slack;
        const double A2 = A > 0.0 ? A * A : (A <= 0.0 ? 0.0 : A);   // A <= 0: any positive distance is outside; NaN stays NaN
        return (dmin2 > A2) | ((B > 0.0) & (dmax2 < B * B));
    }
    if (KIND == RH_CYLINDER) {
        // q(p) = (I - a a')(p - c0) is linear: |q(p) - q(c)| <= max(1, |1 - |a|^2|) * |p - c|; squares as above
        const double ax = P.f[0], ay = P.f[1], az = P.f[2];
        const double tx = cx - P.f[3], ty = cy - P.f[4], tz = cz - P.f[5];
        const double sd = dot3f(ax, ay, az, tx, ty, tz);
        const double qx = __builtin_fma(-ax, sd, tx), qy = __builtin_fma(-ay, sd, ty), qz = __builtin_fma(-az, sd, tz);
        const double rho2 = dot3f(qx, qy, qz, qx, qy, qz);
        const double lip = fmax(1.0, fabs(P.f[9] - 1.0)) * hr;   // |1 - |a|^2| = |k - 1| (prep_derived)
        const double X = ((P.f[6] + eps) + slack) + lip, Y = ((P.f[6] - eps) - slack) - lip;
        const double X2 = X > 0.0 ? X * X : (X <= 0.0 ? 0.0 : X);
        return (rho2 > X2) | ((Y > 0.0) & (rho2 < Y * Y));
    }
    // cone: dist(p) = cos(w/2) rho(p) +- sin(w/2) h(p) (rho, h = radial / axial coordinate of p - apex;
    // the axis only enters through normalized cross products) is 1-Lipschitz in p
    {
        const double ax = P.f[3], ay = P.f[4], az = P.f[5];
        const double c = P.f[6], s = P.f[7];
        const double tx = P.f[0] - cx, ty = P.f[1] - cy, tz = P.f[2] - cz;
        const double ta2 = dot3f(tx, ty, tz, tx, ty, tz);
        double inv = 1.0 / sqrt(ta2);
        const double tnx = inv * tx, tny = inv * ty, tnz = inv * tz;
        // (a x t^) -- cross products as fma(a, b, -(c d)): two instructions per component
        const double k0x = __builtin_fma(ay, tnz, -(az * tny)), k0y = __builtin_fma(az, tnx, -(ax * tnz)), k0z = __builtin_fma(ax, tny, -(ay * tnx));
        const double k02 = dot3f(k0x, k0y, k0z, k0x, k0y, k0z);
        inv = 1.0 / sqrt(k02);
        const double rx = inv * k0x, ry = inv * k0y, rz = inv * k0z;
        double kx = __builtin_fma(ay, rz, -(az * ry)), ky = __builtin_fma(az, rx, -(ax * rz)), kz = __builtin_fma(ax, ry, -(ay * rx));
        inv = 1.0 / sqrt(dot3f(kx, ky, kz, kx, ky, kz));
        const double mx = inv * kx, my = inv * ky, mz = inv * kz;
        // Rodrigues rotation of m about r by the angle whose cos/sin are (c, s): m c + (r x m) s + r (r.m)(1-c)
        const double ux = __builtin_fma(ry, mz, -(rz * my)), uy = __builtin_fma(rz, mx, -(rx * mz)), uz = __builtin_fma(rx, my, -(ry * mx));
        const double rm = dot3f(rx, ry, rz, mx, my, mz) * (1.0 - c);
        const double gx = __builtin_fma(rx, rm, __builtin_fma(ux, s, mx * c));
        const double gy = __builtin_fma(ry, rm, __builtin_fma(uy, s, my * c));
        const double gz = __builtin_fma(rz, rm, __builtin_fma(uz, s, mz * c));
        const double gn = sqrt(dot3f(gx, gy, gz, gx, gy, gz));
        const double dist = dot3f(gx, gy, gz, tx, ty, tz) / gn;
        // a centre on / next to the axis makes the frame ill-conditioned: never skip there
        const double ta = sqrt(ta2);
        const double sinang = sqrt(k02);
        const double an = sqrt(dot3f(ax, ay, az, ax, ay, az));
        // (binary32 exact test: its frame degrades as 2^-24 / sin(angle to the axis): only skip well away from the axis)
        const bool well = (sinang > (F32 ? 3e-2 : 1e-6) * an) & (ta > 0.0);
        return well & (fabs(dist) > ((hr + eps) + slack) + (F32 ? 1e-4 : 1e-6) * (hr + fabs(dist)));
    }
}

}  // namespace rhdev
