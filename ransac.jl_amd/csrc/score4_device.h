// score4_device.h -- the two-sided binary32 CLASSIFIER of the v4 score kernel (score4.hip), its per-candidate records and
// the binary32 box tests of its culling stage.
//
// Idea.  The reference's per-point tests (compatibles{Plane,Sphere,Cylinder,Cone}, /root/reference/src/shapes/*.jl) are
// binary64 and the score must reproduce their bits.  Almost every (candidate, point) pair, however, is nowhere near a
// threshold: a binary32 evaluation of the same quantities (fused multiply-adds allowed; ~2.3 SIMD cycles per wave64
// instruction against ~4.4 for binary64, 8 against 16+ for sqrt / rcp: tools/ubench/valu_rates.hip) already DECIDES it.
// The classifier evaluates each compared quantity q (distance, normal deviation) in binary32 as q32 and compares it with
// TWO thresholds, t - m and t + m, where m bounds |q32 - q64| rigorously (below).  Then
//     q32 inside by more than m   ->  the binary64 test passes:  "sure"
//     q32 outside by more than m  ->  the binary64 test fails
//     otherwise                   ->  "ambiguous": the pair goes to the exact binary64 test (score_device.h), unchanged.
// A point is an inlier iff every comparison passes, so  sure = AND of the sure bits,  maybe = AND of the "not surely
// outside" bits,  ambiguous = maybe & ~sure.  Counts are therefore bit-identical to the exact test whatever the margins'
// tightness; tightness only decides how many pairs take the slow path.
//
// Margins.  u = 2^-24.  Inputs are converted to binary32 (relative error u each), every binary32 operation adds (1 + d),
// |d| <= u.  With M = max |coordinate| of the point set, Nm = max |normal component|, T = M + max |centre coordinate|:
//   plane     dn = n . np - c             |.32 - .|   <= u (5.05 |n|_1 Nm + 4.1 |c|)      (c = cos alpha, added first)
//             d  = oz . p - oz . P0       |d32 - d|   <= 6.1 u (|oz|_1 M + |oz . P0|)
//   sphere    dx = p - o (per component e = 2.01 u T);  nr = |dx| = n2 rsq(n2):  |nr32 - nr| <= sqrt(3) e + 4.5 u nr
//             L = sgn dx . np - c nr  (c = cos alpha; the test is L > 0):
//                                         |L32 - L| <= Nm (3 e + 8.7 u nr) + |c| (sqrt(3) e + 6 u nr)
//   cylinder  q = t - a (a . t), t = p - c0: per component e_q = u T (3.01 + 8.04 |a|_inf |a|_1);  then as the sphere
//             with e_q for e
//   cone      the reference's frame in closed form (cone.jl:68-85 reduces to it, derivation at cls_cone_u): with w = p - apex,
//             h = w . a^, q = w - h a^, rho = |q| (e_q as the cylinder's with the unit axis a^), c', s' = cos / sin of
//             -opang/2 normalised:  D = c' rho + s' h  (the test is |D| < eps)
//                                         |D32 - D|   <= |c'| (sqrt(3) e_q + 5.5 u rho) + |s'| (6 u |a^|_1 T + 2 u |h|) + 4 u eps
//             L = sgn (c' q . np + s' rho a^ . np) - cos(alpha) rho  (the test is L > 0; no division by rho):
//                                         |L32 - L|   <= e_q (3 Nm |c'| + sqrt(3) G) + u rho (8.7 Nm |c'| + 6.5 G + 7 |s'| |a^|_1 Nm + 2 |c|)
//             with G = sqrt(3) Nm (|c'| + |s'|) + |c| + 1/4 and rho, |h| <= sqrt(3) T.  Points next to the axis
//             (rho^2 <= alpha |w|^2 + beta), where the reference's own frame is ill-conditioned, are always undecided.
// Every margin is the first-order bound x RH_CLS_SAFETY (2) plus the conversion error of the threshold itself; the
// binary64 side's own rounding (~1e-15 relative) disappears in that factor.  rh_dbg_cls_audit (score4.hip) evaluates
// max |q32 - q64| / (width of the ambiguity band = 2 m) over real batches with these very functions: sound below 1/2,
// by construction below 1/4 (measured worst 0.21 = 0.84 of the first-order bound); tests/test_parity_gpu.py and
// tools/cls_audit.py hold it below 0.3.  (Round 3 started with a safety factor of 4 and widths rounded up to powers of
// two: 8.6 % of the cylinder pairs of the cfg3 batch were redone in binary64 then.)
//
// The records are SCALED: with wN, wD = 2 x margin, the kernel computes ua = (cN_hi - dn) / wN and ub = (|d| - eD_lo) / wD
// directly (the scaling is folded into the coefficients before they are rounded to binary32), so that with u = max(ua, ub)
//     sure <=> u < 0,   maybe <=> u <= 1                        (read off the bit pattern of u: cls_plane_u below).
//
// Guards.  Non-finite or huge inputs would make binary32 overflow where binary64 does not, so a candidate is marked
// EXACT-ONLY (field RH_CLS_FLAG = NaN) unless all its parameters are finite and below 2^20 (cylinder axis components
// below 16) and M, Nm <= 2^20; a tile with an infinite or NaN value in an enabled point treats every candidate as
// exact-only.  Exact-only means: every enabled point is ambiguous.  Disabled points are staged as zeros and masked out
// (plane: their zero normal fails the angle test by itself, which needs cos(alpha) - margin > 0, else exact-only).
#pragma once

#include "rh_internal.h"

namespace rh4 {

constexpr double RH_CLS_U = 5.9604644775390625e-08;   // 2^-24
constexpr double RH_CLS_SAFETY = 2.0;
constexpr double RH_CLS_BIG = 1048576.0;               // 2^20

// 64-byte record, gathered per lane (lane = (candidate, group) pair or (candidate, point) pair)
struct rh_cls {
    float f[16];
};
constexpr int RH_CLS_FLAG = 15;   // NaN = exact-only
// plane:    0-2 -n / wN | 3 cN_hi / wN | 4-6 oz / wD | 7 -(oz . P0) / wD | 8 eD_lo / wD
// sphere:   0-2 o | 3 R | 4 1 / wD | 5 -eD_lo / wD | 6 -sgn / wN | 7 cN_hi / wN
// cylinder: 0-2 a | 3-5 c0 | 6 R | 7 1 / wD | 8 -eD_lo / wD | 9 -sgn / wN | 10 cN_hi / wN
// cone:     0-2 apex | 3-5 a^ | 6 c' / wD | 7 s' / wD | 8 eD_lo / wD | 9 -sgn c' / wL | 10 -sgn s' / wL | 11 cos(alpha) / wL | 12 beta

// culling record of a candidate, structure-of-arrays over the batch (field f of slot i at box[f * stride + i]): what
// the box tests of stage 1 read, coalesced, with lane = candidate
constexpr int RH_BOX_FIELDS = 11;
// plane:    0-2 oz | 3 -(oz . P0) | 4 eps + slack
// sphere:   0-2 o | 3 A2 | 4 B2
// cylinder: 0-2 a | 3-5 c0 | 6 (R + eps) + slack | 7 (R - eps) - slack | 8 max(1, |1 - |a|^2|)
// cone:     0-2 apex | 3-5 a^ | 6 kk | 7 1 / cn | 8 (eps + slack) / cn | 9 beta | 10 alpha
constexpr float RH_CONE_ALPHA = (float)((RH_CLS_SAFETY * 49.0 + 64.0) * RH_CLS_U);

__host__ __device__ inline bool cls_fin(double v) { return v - v == 0.0; }

// round a threshold to binary32 towards the SAFE side (lo thresholds down, hi thresholds up) by widening with its own ulp
__host__ __device__ inline float cls_dn(double v) { return (float)(v - fabs(v) * (2.0 * RH_CLS_U) - 1e-37); }
__host__ __device__ inline float cls_up(double v) { return (float)(v + fabs(v) * (2.0 * RH_CLS_U) + 1e-37); }

// the old kernel's slack of the conservative stages: covers the exact test's own binary64 rounding (score_device.h)
__host__ __device__ inline double cls_slack64(const rh_prep &P, double M) { return 1e-9 * ((1.0 + M) + P.f[11]); }

// P: the binary64 record of the candidate (rh_prep, kernels.hip prep_one, with prep_derived); eps, cosa: the kind's
// thresholds; M, Nm: max |coordinate| / max |normal component| of the point set.  o: classifier record; box: the culling
// record's fields go to box[f * bstride] (f < RH_BOX_FIELDS).
// dbg4 (audit): the unscaled thresholds and widths behind a scaled record: cN_hi, wN, eD_lo, wD
__host__ __device__ inline void cls_make(const rh_prep &P, int kind, double eps, double cosa, double M, double Nm, rh_cls &o,
                                         float *box, int64_t bstride, double *dbg4 = nullptr, bool f32cloud = false)
{
    // f32cloud: the EXACT test of this cloud is itself a binary32 chain without fused operations (Float32 cloud,
    // score_device32.h: 1.3 - 2 x the rounding steps of the classifier's chain on the same magnitudes).  Every margin
    // is tripled: one part for the classifier's own error, two for the exact chain's.
    const double u = RH_CLS_U, S = f32cloud ? 3.0 * RH_CLS_SAFETY : RH_CLS_SAFETY;
    const float fnan = __builtin_nanf("");
    for (int i = 0; i < 16; i++) o.f[i] = 0.0f;
    float bx[RH_BOX_FIELDS];
    for (int i = 0; i < RH_BOX_FIELDS; i++) bx[i] = fnan;   // NaN fields: the box test never skips
    bool ok = cls_fin(eps) && cls_fin(cosa) && cls_fin(M) && cls_fin(Nm) && M <= RH_CLS_BIG && Nm <= RH_CLS_BIG && fabs(eps) <= RH_CLS_BIG;
    const double slack64 = cls_slack64(P, M);
    if (kind == RH_PLANE) {
        bool fin = true;
        for (int i = 0; i < 9; i++) fin = fin && cls_fin(P.f[i]) && fabs(P.f[i]) <= RH_CLS_BIG;
        ok = ok && fin;
        const double n1 = (fabs(P.f[3]) + fabs(P.f[4])) + fabs(P.f[5]);
        const double z1 = (fabs(P.f[6]) + fabs(P.f[7])) + fabs(P.f[8]);
        const double zp = (P.f[6] * P.f[0] + P.f[7] * P.f[1]) + P.f[8] * P.f[2];
        // (the threshold is the first addend of the fused chain: every partial sum is as large as it is, hence the
        // 4.1 |cos alpha| and 4 |eps| beside the products' own 5.05 / 6.1)
        const double mN = S * u * (5.05 * (n1 * Nm) + 4.1 * (fabs(cosa) + 1e-3)) + 1e-30;
        // Float32 cloud: the test to be bracketed is the reference's OWN binary32 chain oz32 . (p - p0), whose error does not
        // shrink when oz . p0 cancels: p - p0 is rounded per component (u |p_i - p0_i|), oz32 = normalize in binary32 is
        // off by u per component, then three products and two sums -- u (6.2) sum_i |oz_i| (M + |p0_i|) to first order.  A
        // plane through the cloud given by a point 1e6 away is a legal candidate (test_f32_gpu.py).
        const double zabs = (fabs(P.f[6] * P.f[0]) + fabs(P.f[7] * P.f[1])) + fabs(P.f[8] * P.f[2]);
        const double mRef = f32cloud ? RH_CLS_SAFETY * u * 6.2 * (z1 * M + zabs) : 0.0;
        const double mD = S * u * (6.1 * (z1 * M + fabs(zp)) + 4.0 * fabs(eps)) + mRef + 1e-30;
        ok = ok && cls_fin(mN) && cls_fin(mD) && cls_fin(zp) && mN < 1e30 && mD < 1e30;
        if (ok) {
            const double wN = 2.0 * mN, wD = 2.0 * mD;
            const double iN = 1.0 / wN, iD = 1.0 / wD;
            const double cNhi = cosa + 0.5 * wN, eDlo = eps - 0.5 * wD;
            ok = cosa - 0.5 * wN > 0.0;   // a zero normal (a disabled point) must fail the angle test surely
            o.f[0] = (float)(-P.f[3] * iN); o.f[1] = (float)(-P.f[4] * iN); o.f[2] = (float)(-P.f[5] * iN);
            o.f[3] = (float)(cNhi * iN);
            o.f[4] = (float)(P.f[6] * iD); o.f[5] = (float)(P.f[7] * iD); o.f[6] = (float)(P.f[8] * iD);
            o.f[7] = (float)(-zp * iD);
            o.f[8] = (float)(eDlo * iD);
            if (dbg4 != nullptr) { dbg4[0] = cNhi; dbg4[1] = wN; dbg4[2] = eDlo; dbg4[3] = wD; }
            for (int i = 0; i < 9; i++) ok = ok && fabs((double)o.f[i]) < 1e30;
        }
        if (fin && cls_fin(zp) && cls_fin(eps) && cls_fin(M)) {
            // box: d(centre) against eps + sum |oz_i| h_i; binary32 error of both sides + the binary64 test's own slack
            const double sB = S * 10.1 * u * (z1 * M + fabs(zp)) + mRef + slack64;
            bx[0] = (float)P.f[6]; bx[1] = (float)P.f[7]; bx[2] = (float)P.f[8];
            bx[3] = (float)(-zp);
            bx[4] = cls_up(eps + sB);
        }
    } else if (kind == RH_SPHERE || kind == RH_CYLINDER) {
        const bool sph = kind == RH_SPHERE;
        bool fin = true;
        for (int i = 0; i < 7; i++) fin = fin && ((sph && i >= 4) || (cls_fin(P.f[i]) && fabs(P.f[i]) <= RH_CLS_BIG));   // (no run-time indices)
        ok = ok && fin;
        const double R = sph ? P.f[3] : P.f[6], sgn = sph ? P.f[4] : P.f[7];
        const double c0 = sph ? P.f[0] : P.f[3], c1 = sph ? P.f[1] : P.f[4], c2 = sph ? P.f[2] : P.f[5];
        const double T = M + fmax(fabs(c0), fmax(fabs(c1), fabs(c2)));
        double e, lipk = 1.0;   // e: error bound per component of the vector whose norm is taken
        bool axis_ok = true;
        if (sph) {
            e = 2.01 * u * T;
        } else {
            const double a1 = (fabs(P.f[0]) + fabs(P.f[1])) + fabs(P.f[2]);
            const double ai = fmax(fabs(P.f[0]), fmax(fabs(P.f[1]), fabs(P.f[2])));
            axis_ok = ai <= 16.0;
            ok = ok && axis_ok;
            e = u * T * (3.01 + 8.04 * ai * a1);
            lipk = fmax(1.0, fabs(P.f[9] - 1.0));   // |1 - |a|^2| = |k - 1| (prep_derived)
        }
        const double s3 = 1.7320508075688774;
        const double mD = S * (s3 * e + 7.0 * u * (fabs(R) + fabs(eps))) + 1e-30;   // (nr = n2 * rsq(n2): 4.5 u; R, eps and the scaled constant: 2 u)
        // a point the distance half may accept (|nr32 - R| < eps + wD / 2, wD < 4 mD) lies at least this far from the
        // centre / axis
        const double nrmin = R - eps - 3.0 * mD;
        // margin of the normal half as a margin on cos(alpha): |L32 - L| / nr.  Without a positive lower bound on nr (the
        // centre lies inside the band) or with a margin that is not small the candidate is exact-only.
        double mN = nrmin > 0.0 ? S * ((3.0 * Nm + s3 * (fabs(cosa) + 0.25)) * e / nrmin + (8.7 * Nm + 6.0 * (fabs(cosa) + 0.25)) * u) + 1e-30
                                : __builtin_nan("");
        if (!(mN <= 0.25)) mN = __builtin_nan("");
        // scaled like the plane record: ua = (|nr - R| - eD_lo) / wD, ub = (cN_hi - sgn (q . np) / nr) / wN, u = max(ua, ub);
        // a point whose distance half fails surely has ua > 1 whatever ub is
        ok = ok && cls_fin(mD) && cls_fin(mN) && (sgn == 1.0 || sgn == -1.0) && mD < 1e30;
        // (constant indices only: a run-time index would move the record to scratch memory)
        if (sph) { o.f[0] = (float)P.f[0]; o.f[1] = (float)P.f[1]; o.f[2] = (float)P.f[2]; o.f[3] = (float)R; }
        else { o.f[0] = (float)P.f[0]; o.f[1] = (float)P.f[1]; o.f[2] = (float)P.f[2]; o.f[3] = (float)P.f[3]; o.f[4] = (float)P.f[4]; o.f[5] = (float)P.f[5]; o.f[6] = (float)R; }
        if (ok) {
            const double mN2 = mN + 8.0 * u * Nm;   // + the rsq / multiply in place of the comparison against c * nr
            const double wD = 2.0 * mD, wN = 2.0 * mN2;
            const double eDlo = eps - 0.5 * wD, cNhi = cosa + 0.5 * wN;
            ok = cosa - 0.5 * wN > 0.0;   // a zero normal (a disabled point) must fail the angle test surely
            const double iN = 1.0 / wN, iD = 1.0 / wD;
            const float s0 = (float)iD, s1 = (float)(-eDlo * iD), s2 = (float)(-sgn * iN), s3f = (float)(cNhi * iN);
            if (dbg4 != nullptr) { dbg4[0] = cNhi; dbg4[1] = wN; dbg4[2] = eDlo; dbg4[3] = wD; }
            if (sph) { o.f[4] = s0; o.f[5] = s1; o.f[6] = s2; o.f[7] = s3f; }
            else { o.f[7] = s0; o.f[8] = s1; o.f[9] = s2; o.f[10] = s3f; }
            ok = ok && fabs((double)s0) < 1e30 && fabs((double)s1) < 1e30 && fabs((double)s2) < 1e30 && fabs((double)s3f) < 1e30;
        }
        if (fin && axis_ok && cls_fin(eps) && cls_fin(M)) {
            // box: |p - o| (sphere) or the distance from the axis (cylinder, at the centre of the box +- Lipschitz) against
            // R +- eps; binary32 error of the norm at the centre + of the squares + the binary64 test's own slack
            const double sB = S * (s3 * e + 4.0 * u * (fabs(R) + fabs(eps)) + 4.0 * u * lipk * s3 * M) + slack64;
            const double A = (R + eps) + sB, B = (R - eps) - sB;
            if (cls_fin(A) && cls_fin(B)) {
                if (sph) {
                    bx[0] = (float)P.f[0]; bx[1] = (float)P.f[1]; bx[2] = (float)P.f[2];
                    bx[3] = A > 0.0 ? cls_up(A * A * (1.0 + 16.0 * u)) : 0.0f;   // A <= 0: any positive distance is outside
                    bx[4] = B > 0.0 ? cls_dn(B * B * (1.0 - 16.0 * u)) : -1.0f;  // B <= 0: never "inside the inner ball"
                } else {
                    for (int i = 0; i < 6; i++) bx[i] = (float)P.f[i];
                    bx[6] = cls_up(A);
                    bx[7] = cls_dn(B);
                    bx[8] = cls_up(lipk);
                }
            }
        }
    } else {
        // cone: the two-sided classifier works on the closed form of the reference's frame (cls_cone_u below); the culling
        // record keeps the band form: with t = p - apex, h = t . a^, rho^2 = |t|^2 - h^2 the reference's distance is
        // -(c rho + s h) / sqrt(c^2 + s^2) (c, s = cos / sin of -opang/2): |dist| < eps <=> rho in (k h - e, k h + e),
        // k = -s / c, e = eps sqrt(c^2 + s^2) / c   (c > 0)
        for (int i = 0; i < 8; i++) ok = ok && cls_fin(P.f[i]) && fabs(P.f[i]) <= RH_CLS_BIG;
        const double ax = P.f[3], ay = P.f[4], az = P.f[5], c = P.f[6], sn = P.f[7], sgn = P.f[8];
        const double an = sqrt((ax * ax + ay * ay) + az * az), cs = sqrt(c * c + sn * sn);
        ok = ok && (an > 1e-300) && cls_fin(an) && (cs > 1e-300) && cls_fin(cs) && (sgn == 1.0 || sgn == -1.0);
        const bool band_ok = ok && (c > 1e-6 * cs);     // the band form (culling record; Float32 clouds: the point record too)
        const double ia = ok ? 1.0 / an : 0.0;
        const double T = M + fmax(fabs(P.f[0]), fmax(fabs(P.f[1]), fabs(P.f[2])));
        const double kk = band_ok ? -sn / c : 0.0;
        const double cn = band_ok ? c / cs : 1.0;
        // band half width: exact form + the f64 prefilter's slack + binary32 error of k h (|t| <= sqrt(3) T)
        // (Float32 cloud: the exact test's frame is a long binary32 chain, ~1e-5 relative near its band: score_device.h F32 margins)
        const double ek = S * 19.1 * fabs(kk) * u * T + 1e-8 * (1.0 + T) * (1.0 + fabs(kk)) + (f32cloud ? 2e-5 * (1.0 + T) * (1.0 + fabs(kk)) : 0.0);
        const double beta = S * 1.75 * u * T * T + 1e-30;
        // ---- the point record
        const double a0 = ax * ia, a1v = ay * ia, a2v = az * ia;          // a^
        const double cp = ok ? c / cs : 0.0, sp = ok ? sn / cs : 0.0;     // c', s'
        const double a1 = (fabs(a0) + fabs(a1v)) + fabs(a2v), ai = fmax(fabs(a0), fmax(fabs(a1v), fabs(a2v)));
        const double s3 = 1.7320508075688774;
        const double eq = u * T * (3.01 + 8.04 * ai * a1);
        const double rmax = s3 * T;
        // (classifier's own binary32 error, first order x RH_CLS_SAFETY -- never the tripled S: a Float32 cloud's exact
        // chain is bracketed by its own term below)
        const double mDc = RH_CLS_SAFETY * (fabs(cp) * (s3 * eq + 5.5 * u * rmax) + fabs(sp) * (6.0 * u * a1 * T + 2.0 * u * rmax) + 4.0 * u * fabs(eps));
        const double G = s3 * Nm * (fabs(cp) + fabs(sp)) + fabs(cosa) + 0.25;
        const double mLc = RH_CLS_SAFETY * (eq * (3.0 * Nm * fabs(cp) + s3 * G) +
                                            u * rmax * (8.7 * Nm * fabs(cp) + 6.5 * G + 7.0 * fabs(sp) * a1 * Nm + 2.0 * fabs(cosa)));
        double mD, mL;
        if (!f32cloud) {
            // the binary64 chain of the reference differs from the closed form by ~1e-15 |t| / sin(angle to the axis); the
            // axis guard (alpha, beta) keeps that sine above ~3e-3: 1e-11 (1 + T) covers it 30 times over
            mD = mDc + 1e-11 * (1.0 + T) + 1e-30;
            mL = mLc + 1e-11 * (1.0 + T) * (1.0 + Nm) + 1e-30;
        } else {
            // Float32 cloud: the exact test is a ~60-operation binary32 chain.  Its distance is bracketed like round 3's
            // prefilter did (band form, needs c > 0; 2e-5 relative: in ek); its normal half is never decided here -- every
            // point inside the widened band goes to the exact test.
            ok = ok && band_ok;
            mD = mDc + ek * cn + 1e-9 * fabs(eps) + 1e-30;
            mL = __builtin_inf();
        }
        o.f[0] = (float)P.f[0]; o.f[1] = (float)P.f[1]; o.f[2] = (float)P.f[2];
        o.f[3] = (float)a0; o.f[4] = (float)a1v; o.f[5] = (float)a2v;
        ok = ok && cls_fin(mD) && mD < 1e30 && (f32cloud || (cls_fin(mL) && mL < 1e30)) && cls_fin(beta);
        if (ok) {
            const double wD = 2.0 * mD, iD = 1.0 / wD, eDlo = eps - 0.5 * wD;
            const double wL = 2.0 * mL, iL = f32cloud ? 0.0 : 1.0 / wL;
            o.f[6] = (float)(cp * iD); o.f[7] = (float)(sp * iD);
            o.f[8] = (float)(eDlo * iD);
            o.f[9] = (float)(-sgn * cp * iL); o.f[10] = (float)(-sgn * sp * iL); o.f[11] = (float)(cosa * iL);
            if (dbg4 != nullptr) { dbg4[0] = 0.0; dbg4[1] = wL; dbg4[2] = eDlo; dbg4[3] = wD; }
            for (int i = 6; i < 12; i++) ok = ok && fabs((double)o.f[i]) < 1e30;
        }
        o.f[12] = cls_up(beta);     // next to the axis: rho^2 <= alpha |w|^2 + beta -> undecided (alpha = RH_CONE_ALPHA)
        const double e = band_ok ? (eps / cn) * (1.0 + 1e-9) + ek : 0.0;
        const bool box_ok = band_ok && cls_fin(e) && cls_fin(kk) && fabs(kk) <= 1e6 && cls_fin(beta);
        if (box_ok) {
            // box: the same closed form at the centre of the box, the band widened by the radius of the box (the
            // distance is 1-Lipschitz in p): half width (hr + eps + slack) / cn
            bx[0] = o.f[0]; bx[1] = o.f[1]; bx[2] = o.f[2];
            bx[3] = o.f[3]; bx[4] = o.f[4]; bx[5] = o.f[5];
            bx[6] = (float)kk;
            bx[7] = cls_up(1.0 / cn);
            bx[8] = cls_up((eps + slack64) / cn * (1.0 + 1e-9) + ek);
            bx[9] = cls_up(beta);
            // Float32 cloud: the exact test's frame degrades as 2^-24 / sin(angle to the axis) -- never skip a box whose
            // centre lies within ~0.03 rad of the axis (rho^2 <= alpha |t|^2)
            bx[10] = f32cloud ? 1e-3f : RH_CONE_ALPHA;
        }
    }
    if (!ok) o.f[RH_CLS_FLAG] = fnan;
    if (box != nullptr)
        for (int i = 0; i < RH_BOX_FIELDS; i++) box[(int64_t)i * bstride] = bx[i];
}

// binary32 box of a 64-point group from its binary64 box (centre c, half extents h): the half extents absorb the
// conversion of the centre and are rounded up; o = cx cy cz hx hy hz hr 0
__host__ __device__ inline void box_to_f32(const double c[3], const double h[3], float o[8])
{
    double h2 = 0;
    for (int k = 0; k < 3; k++) {
        const float cf = (float)c[k];
        const double hh = (h[k] + fabs(c[k] - (double)cf)) * (1.0 + 4.0 * RH_CLS_U) + 1e-37;
        const float hf = cls_up(hh);
        o[k] = cf;
        o[3 + k] = hf;
        h2 += (double)hf * (double)hf;
    }
    o[6] = cls_up(sqrt(h2));
    o[7] = 0.0f;
}

#ifdef __HIPCC__
#define WB4(cond) __builtin_amdgcn_ballot_w64(cond)

// ---- stage 1: may the group's box be skipped for this candidate?  lane = candidate (B = its culling record),
// the box is wave-uniform.  Every comparison is written so that NaN means "do not skip".
struct rh_box32 { float cx, cy, cz, hx, hy, hz, hr; };

template <int KIND>
static __device__ __forceinline__ bool box_skip32(const float (&B)[RH_BOX_FIELDS], const rh_box32 &G)
{
    if (KIND == RH_PLANE) {
        const float d = __builtin_fmaf(B[2], G.cz, __builtin_fmaf(B[1], G.cy, __builtin_fmaf(B[0], G.cx, B[3])));
        const float ext = __builtin_fmaf(__builtin_fabsf(B[2]), G.hz, __builtin_fmaf(__builtin_fabsf(B[1]), G.hy, __builtin_fmaf(__builtin_fabsf(B[0]), G.hx, B[4])));
        return __builtin_fabsf(d) > ext;
    }
    if (KIND == RH_SPHERE) {
        const float ax = __builtin_fabsf(G.cx - B[0]), ay = __builtin_fabsf(G.cy - B[1]), az = __builtin_fabsf(G.cz - B[2]);
        const float nx = fmaxf(ax - G.hx, 0.0f), ny = fmaxf(ay - G.hy, 0.0f), nz = fmaxf(az - G.hz, 0.0f);
        const float fx = ax + G.hx, fy = ay + G.hy, fz = az + G.hz;
        const float dmin2 = __builtin_fmaf(nz, nz, __builtin_fmaf(ny, ny, nx * nx));
        const float dmax2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx)) * 1.000001f;
        return (dmin2 * 0.999999f > B[3]) | (dmax2 < B[4]);
    }
    if (KIND == RH_CYLINDER) {
        const float tx = G.cx - B[3], ty = G.cy - B[4], tz = G.cz - B[5];
        const float sd = __builtin_fmaf(B[2], tz, __builtin_fmaf(B[1], ty, B[0] * tx));
        const float qx = __builtin_fmaf(-B[0], sd, tx), qy = __builtin_fmaf(-B[1], sd, ty), qz = __builtin_fmaf(-B[2], sd, tz);
        const float rho2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
        const float lip = B[8] * G.hr * 1.000001f;
        const float X = B[6] + lip, Y = B[7] - lip;
        const float X2 = X > 0.0f ? X * X * 1.000001f : (X <= 0.0f ? 0.0f : X);   // NaN stays NaN
        return (rho2 * 0.999999f > X2) | ((Y > 0.0f) & (rho2 * 1.000001f < Y * Y));
    }
    // cone: "surely outside the widened band", positive comparisons only
    const float tx = G.cx - B[0], ty = G.cy - B[1], tz = G.cz - B[2];
    const float tt = __builtin_fmaf(tz, tz, __builtin_fmaf(ty, ty, tx * tx));
    const float h = __builtin_fmaf(tz, B[5], __builtin_fmaf(ty, B[4], tx * B[3]));
    const float rho2 = __builtin_fmaf(-h, h, tt);
    const float s2 = __builtin_fmaf(B[10], tt, B[9]);
    const float e = __builtin_fmaf(G.hr, B[7], B[8]) * 1.000001f;
    const float uu = B[6] * h;
    const float lo = uu - e, hi = uu + e;
    const float hi2 = __builtin_fmaf(hi, hi, s2), lo2 = __builtin_fmaf(lo, lo, -s2);
    return (rho2 > s2) & ((hi <= 0.0f) | (rho2 > hi2) | ((lo > 0.0f) & (rho2 < lo2)));
}

// ---- The classifier's value of a (candidate, point) pair: u = max(ua, ub), the scaled "outsideness" of the two tests --
// ua = (threshold_lo - quantity) / width for the test that wants its quantity large (the normal's cosine), (quantity -
// threshold_lo) / width for the one that wants it small (the distance), widths = 2 x margin:
//     u < 0  (sign bit set)   both tests pass by more than their margins: SURELY an inlier
//     0 <= u <= 1             some test is within its margin and none fails surely: UNDECIDED -> the exact test
//     u > 1                   some test fails by more than its margin: surely not
// so that the 64-point loop of the score kernel (score4.hip) keeps the books with two instructions per point and no
// compare: the sign bit of u is shifted into the pair's inlier word (v_alignbit_b32), and a running UNSIGNED minimum over
// the bit patterns of the u (v_min3_u32, two points per instruction) ends <= bits(1.0f) iff some point was undecided --
// non-negative floats order like their bit patterns and every negative one lies above them.  (Measured on this GPU,
// tools/ubench/count_seq.hip: v_cmp / v_cndmask / v_addc / v_min_f32 hold the issue port 4.2 cycles each, fma / add / mul
// 2.3-2.9; round 3's t = min(a, b) with `count += t > 1/2` and a running minimum of |t| cost 14.6 cycles per point beside the
// arithmetic, this form 10.5.)  u = -0 counts as surely-in: it stands for a test passed by exactly its margin, and the
// margins carry a factor RH_CLS_SAFETY over the first-order bound.
static __device__ __forceinline__ void cls_plane_ab(const rh_cls &C, float x, float y, float z, float nx, float ny, float nz, float &ua, float &ub)
{
    ua = __builtin_fmaf(C.f[2], nz, __builtin_fmaf(C.f[1], ny, __builtin_fmaf(C.f[0], nx, C.f[3])));
    const float d = __builtin_fmaf(C.f[6], z, __builtin_fmaf(C.f[5], y, __builtin_fmaf(C.f[4], x, C.f[7])));
    ub = __builtin_fabsf(d) - C.f[8];
}
static __device__ __forceinline__ float cls_plane_u(const rh_cls &C, float x, float y, float z, float nx, float ny, float nz)
{
    float a, b;
    cls_plane_ab(C, x, y, z, nx, ny, nz, a, b);
    return fmaxf(a, b);
}

// ---- sphere / cylinder: the same u = max(ua, ub) from the scaled record
template <int KIND>
static __device__ __forceinline__ void cls_round_ab(const rh_cls &C, float x, float y, float z, float nx, float ny, float nz, float &ua, float &ub)
{
    float qx, qy, qz;
    constexpr int b = KIND == RH_SPHERE ? 3 : 6;
    if (KIND == RH_SPHERE) {
        qx = x - C.f[0]; qy = y - C.f[1]; qz = z - C.f[2];
    } else {
        const float tx = x - C.f[3], ty = y - C.f[4], tz = z - C.f[5];
        const float sd = __builtin_fmaf(C.f[2], tz, __builtin_fmaf(C.f[1], ty, C.f[0] * tx));
        qx = __builtin_fmaf(-C.f[0], sd, tx); qy = __builtin_fmaf(-C.f[1], sd, ty); qz = __builtin_fmaf(-C.f[2], sd, tz);
    }
    // (+ 1e-37: a point ON the centre / axis gives nr = 0 and a zero cosine -- surely out, as the exact test's NaN says --
    // instead of 0 x inf = NaN, whose sign bit the books below would read; any other n2 absorbs it unchanged)
    const float n2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, __builtin_fmaf(qx, qx, 1e-37f)));
    const float inr = __builtin_amdgcn_rsqf(n2);
    const float nr = n2 * inr;
    const float dt = __builtin_fmaf(qz, nz, __builtin_fmaf(qy, ny, qx * nx)) * inr;
    ua = __builtin_fmaf(__builtin_fabsf(nr - C.f[b]), C.f[b + 1], C.f[b + 2]);
    ub = __builtin_fmaf(dt, C.f[b + 3], C.f[b + 4]);
}
template <int KIND>
static __device__ __forceinline__ float cls_round_u(const rh_cls &C, float x, float y, float z, float nx, float ny, float nz)
{
    float a, b;
    cls_round_ab<KIND>(C, x, y, z, nx, ny, nz, a, b);
    return fmaxf(a, b);
}

// ---- cone: the same u = max(ua, ub).  The reference builds a frame per point (project2cone, cone.jl:68-85): with
// w = p - apex = h a^ + rho e_rho (a^ the unit axis, e_rho the unit radial direction, phi^ = a^ x e_rho)
//     rot_ax = normalize(axis x normalize(-w)) = -phi^,   comp_n = normalize(axis x rot_ax) = e_rho,
//     current_normal = normalize(R(rot_ax, -opang/2) comp_n) = (c e_rho + s (rot_ax x comp_n)) / |(c, s)| = c' e_rho + s' a^
// so dist = dot(-current_normal, apex - p) = -(c' rho + s' h) and the normal's cosine is c' (q . n) / rho + s' (a^ . n).
// ua = (|D| - eD_lo) / wD with D = c' rho + s' h;  ub = 1/2 - L / wL with L = sgn (c' q . n + s' rho a^ . n) - cos(alpha) rho
// (the angle test multiplied through by rho > 0: its margin is absolute, nothing is divided by a small rho).  Next to the
// axis the reference's frame is ill-conditioned (and NaN on it): u = 1/2, undecided, the exact test answers.
static __device__ __forceinline__ void cls_cone_ab(const rh_cls &C, float x, float y, float z, float nx, float ny, float nz, float &ua, float &ub,
                                                   bool &near_axis)
{
    const float wx = x - C.f[0], wy = y - C.f[1], wz = z - C.f[2];
    const float h = __builtin_fmaf(wz, C.f[5], __builtin_fmaf(wy, C.f[4], wx * C.f[3]));
    const float qx = __builtin_fmaf(-C.f[3], h, wx), qy = __builtin_fmaf(-C.f[4], h, wy), qz = __builtin_fmaf(-C.f[5], h, wz);
    const float n2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
    const float rho = n2 * __builtin_amdgcn_rsqf(n2);
    const float D = __builtin_fmaf(C.f[6], rho, C.f[7] * h);
    ua = __builtin_fabsf(D) - C.f[8];
    const float qn = __builtin_fmaf(qz, nz, __builtin_fmaf(qy, ny, qx * nx));
    const float an = __builtin_fmaf(C.f[5], nz, __builtin_fmaf(C.f[4], ny, C.f[3] * nx));
    ub = __builtin_fmaf(qn, C.f[9], __builtin_fmaf(rho, __builtin_fmaf(an, C.f[10], C.f[11]), 0.5f));
    const float tt = __builtin_fmaf(h, h, n2);
    near_axis = !(n2 > __builtin_fmaf(RH_CONE_ALPHA, tt, C.f[12]));   // (NaN -> near; alpha is the same for every cone)
}
static __device__ __forceinline__ float cls_cone_u(const rh_cls &C, float x, float y, float z, float nx, float ny, float nz)
{
    float a, b;
    bool near_axis;
    cls_cone_ab(C, x, y, z, nx, ny, nz, a, b, near_axis);
    return near_axis ? 0.5f : fmaxf(a, b);
}
// what the books say of a u (score kernel, audits)
static __device__ __forceinline__ bool cls_sure(float u) { return (__builtin_bit_cast(uint32_t, u) >> 31) != 0; }
static __device__ __forceinline__ bool cls_undecided(float u) { return __builtin_bit_cast(uint32_t, u) <= 0x3f800000u; }   // 0 <= u <= 1
#endif   // __HIPCC__

}  // namespace rh4
