// score_device32.h -- the binary32 twins of the four per-point tests of score_device.h, statement by statement (Float32
// clouds: RANSACCloud(...; force_eltype = Float32), src/octree.jl:102-109).  Every operation is a float operation in the
// reference's order, no FMA (-ffp-contract=off), correctly rounded sqrtf and division; eps and cos(alpha) stay doubles
// and are compared with the float result after exact promotion, like Julia compares a Float32 with a Float64.
// The oracle's twin is oracle/orc_f32.c.
#pragma once

#include "rh_internal.h"

namespace rhdev32 {

#define WB32(cond) __builtin_amdgcn_ballot_w64(cond)

struct rh_prepf {
    float f[12];
};

// float record of a candidate: the fields of the shape rounded to binary32 (exact for a Float32 shape), per-candidate
// constants in binary32 with the reference's operations (normalize(plane.normal), plane.jl:85)
__host__ __device__ inline void prep_one32(const rh_shape &s, rh_prepf &o)
{
    for (int i = 0; i < 12; i++) o.f[i] = 0.0f;
    const float sgn = s.outwards ? 1.0f : -1.0f;
    switch (s.kind) {
    case RH_PLANE: {
        for (int i = 0; i < 6; i++) o.f[i] = (float)s.v[i];
        const float a = o.f[3], b = o.f[4], c = o.f[5];
        const float inv = 1.0f / sqrtf((a * a + b * b) + c * c);
        o.f[6] = inv * a; o.f[7] = inv * b; o.f[8] = inv * c;
        break;
    }
    case RH_SPHERE:
        for (int i = 0; i < 4; i++) o.f[i] = (float)s.v[i];
        o.f[4] = sgn;
        break;
    case RH_CYLINDER:
        for (int i = 0; i < 7; i++) o.f[i] = (float)s.v[i];
        o.f[7] = sgn;
        break;
    default:
        for (int i = 0; i < 6; i++) o.f[i] = (float)s.v[i];
        o.f[6] = (float)s.v[7];   // cos(-opang/2), a binary32 number on a Float32 shape
        o.f[7] = (float)s.v[8];
        o.f[8] = sgn;
        break;
    }
}

// The float record of a candidate from its binary64 record (rh_prep, kernels.hip prep_one): the record's leading fields are the
// shape's own numbers (+ sgn), so the casts reproduce prep_one32 -- exactly for a Float32 shape, whose fields are binary32
// numbers, and with the same single rounding for any other; a plane's normalised normal (f[6..8]) is recomputed in
// binary32 (plane.jl:85 on a Float32 plane).  Lets every kernel that holds only rh_prep records (the candidate store of
// rh_ransac, the liveness passes) run the binary32 test of a Float32 cloud.
template <int KIND>
__host__ __device__ inline rh_prepf prepf_of(const rh_prep &P)
{
    rh_prepf o;
    for (int i = 0; i < 9; i++) o.f[i] = (float)P.f[i];
    for (int i = 9; i < 12; i++) o.f[i] = 0.0f;
    if (KIND == RH_PLANE) {
        const float a = o.f[3], b = o.f[4], c = o.f[5];
        const float inv = 1.0f / sqrtf((a * a + b * b) + c * c);
        o.f[6] = inv * a; o.f[7] = inv * b; o.f[8] = inv * c;
    }
    if (KIND == RH_SPHERE) { o.f[5] = o.f[6] = o.f[7] = o.f[8] = 0.0f; }
    if (KIND == RH_CYLINDER) { o.f[8] = 0.0f; }
    return o;
}
__host__ __device__ inline rh_prepf prepf_of_kind(const rh_prep &P, int kind)
{
    switch (kind) {
    case RH_PLANE: return prepf_of<RH_PLANE>(P);
    case RH_SPHERE: return prepf_of<RH_SPHERE>(P);
    case RH_CYLINDER: return prepf_of<RH_CYLINDER>(P);
    default: return prepf_of<RH_CONE>(P);
    }
}

#ifdef __HIPCC__
#define RH_CONST32 __attribute__((address_space(4)))
static __device__ __forceinline__ rh_prepf ld_prepf(const rh_prepf *p)
{
    const RH_CONST32 rh_prepf *q = (const RH_CONST32 rh_prepf *)(uintptr_t)p;   // wave-uniform: scalar loads
    rh_prepf o;
#pragma unroll
    for (int i = 0; i < 12; i++) o.f[i] = q->f[i];
    return o;
}

// plane: compatiblesPlane plane.jl:114-130 (+ project2plane :82-95), isparallel utilities.jl:115-117
static __device__ __forceinline__ uint64_t test_plane32(const rh_prepf &P, float px, float py, float pz, float nx, float ny, float nz,
                                                 double eps, double cosa)
{
    const float dn = (P.f[3] * nx + P.f[4] * ny) + P.f[5] * nz;
    const uint64_t mn = WB32((double)dn > cosa);
    if (mn == 0) return 0;
    const float vx = px - P.f[0], vy = py - P.f[1], vz = pz - P.f[2];
    const float d = (P.f[6] * vx + P.f[7] * vy) + P.f[8] * vz;
    return mn & WB32((double)fabsf(d) < eps);
}

// sphere: compatiblesSphere sphere.jl:144-172 (inward: sgn * dot, exact)
static __device__ __forceinline__ uint64_t test_sphere32(const rh_prepf &P, float px, float py, float pz, float nx, float ny, float nz,
                                                  double eps, double cosa)
{
    const float dx = px - P.f[0], dy = py - P.f[1], dz = pz - P.f[2];
    const float nr = sqrtf((dx * dx + dy * dy) + dz * dz);
    const uint64_t md = WB32((double)fabsf(nr - P.f[3]) < eps);
    if (md == 0) return 0;
    const float inv = 1.0f / nr;
    const float ux = inv * dx, uy = inv * dy, uz = inv * dz;
    const float dt = (ux * nx + uy * ny) + uz * nz;
    return WB32((double)(P.f[4] * dt) > cosa) & md;
}

// cylinder: compatiblesCylinder cylinder.jl:194-221
static __device__ __forceinline__ uint64_t test_cylinder32(const rh_prepf &P, float px, float py, float pz, float nx, float ny, float nz,
                                                    double eps, double cosa)
{
    const float ax = P.f[0], ay = P.f[1], az = P.f[2];
    const float cx = P.f[3], cy = P.f[4], cz = P.f[5];
    const float tx = px - cx, ty = py - cy, tz = pz - cz;
    const float sd = (ax * tx + ay * ty) + az * tz;
    const float qx = (px - ax * sd) - cx, qy = (py - ay * sd) - cy, qz = (pz - az * sd) - cz;
    const float nr = sqrtf((qx * qx + qy * qy) + qz * qz);
    const uint64_t md = WB32((double)fabsf(nr - P.f[6]) < eps);
    if (md == 0) return 0;
    const float inv = 1.0f / nr;
    const float ux = inv * qx, uy = inv * qy, uz = inv * qz;
    const float dt = (ux * nx + uy * ny) + uz * nz;
    return md & WB32((double)(P.f[7] * dt) > cosa);
}

// cone: compatiblesCone cone.jl:132-153, project2cone :68-85, rodriguesrad / rodrigues / pluscrossprod! utilities.jl:61-64,19-24,32-43
static __device__ __forceinline__ uint64_t test_cone32(const rh_prepf &P, float px, float py, float pz, float nx, float ny, float nz,
                                                double eps, double cosa)
{
    const float ax = P.f[3], ay = P.f[4], az = P.f[5];
    const float c = P.f[6], s = P.f[7];
    const float tx = P.f[0] - px, ty = P.f[1] - py, tz = P.f[2] - pz;
    float inv = 1.0f / sqrtf((tx * tx + ty * ty) + tz * tz);
    const float tnx = inv * tx, tny = inv * ty, tnz = inv * tz;
    float kx = ay * tnz - az * tny, ky = az * tnx - ax * tnz, kz = ax * tny - ay * tnx;
    inv = 1.0f / sqrtf((kx * kx + ky * ky) + kz * kz);
    const float rx = inv * kx, ry = inv * ky, rz = inv * kz;
    kx = ay * rz - az * ry; ky = az * rx - ax * rz; kz = ax * ry - ay * rx;
    inv = 1.0f / sqrtf((kx * kx + ky * ky) + kz * kz);
    const float mx = inv * kx, my = inv * ky, mz = inv * kz;
    inv = 1.0f / sqrtf((rx * rx + ry * ry) + rz * rz);
    const float vx = inv * rx, vy = inv * ry, vz = inv * rz;
    const float nxx = vx * vx, nxy = vx * vy, nxz = vx * vz, nyy = vy * vy, nyz = vy * vz, nzz = vz * vz;
    const float R00 = nxx + c * (1.0f - nxx);
    float R01 = nxy + c * (0.0f - nxy);
    float R02 = nxz + c * (0.0f - nxz);
    float R10 = R01;
    const float R11 = nyy + c * (1.0f - nyy);
    float R12 = nyz + c * (0.0f - nyz);
    float R20 = R02;
    float R21 = R12;
    const float R22 = nzz + c * (1.0f - nzz);
    R01 -= s * vz; R02 += s * vy;
    R10 += s * vz; R12 -= s * vx;
    R20 -= s * vy; R21 += s * vx;
    kx = (R00 * mx + R01 * my) + R02 * mz;
    ky = (R10 * mx + R11 * my) + R12 * mz;
    kz = (R20 * mx + R21 * my) + R22 * mz;
    inv = 1.0f / sqrtf((kx * kx + ky * ky) + kz * kz);
    const float gx = inv * kx, gy = inv * ky, gz = inv * kz;
    const float dist = ((-gx) * (-tx) + (-gy) * (-ty)) + (-gz) * (-tz);
    const float dt = (gx * nx + gy * ny) + gz * nz;
    return WB32((double)(P.f[8] * dt) > cosa) & WB32((double)fabsf(dist) < eps);
}

template <int KIND>
static __device__ __forceinline__ uint64_t test_point32(const rh_prepf &P, float px, float py, float pz, float nx, float ny, float nz,
                                                 double eps, double cosa)
{
    if (KIND == RH_PLANE) return test_plane32(P, px, py, pz, nx, ny, nz, eps, cosa);
    if (KIND == RH_SPHERE) return test_sphere32(P, px, py, pz, nx, ny, nz, eps, cosa);
    if (KIND == RH_CYLINDER) return test_cylinder32(P, px, py, pz, nx, ny, nz, eps, cosa);
    return test_cone32(P, px, py, pz, nx, ny, nz, eps, cosa);
}


static __device__ __forceinline__ uint64_t valid_mask32(int64_t base, int64_t s)
{
    const int64_t left = s - base;
    return left >= 64 ? ~0ULL : (left <= 0 ? 0ULL : ((1ULL << left) - 1ULL));
}

#endif   // __HIPCC__

}  // namespace rhdev32
