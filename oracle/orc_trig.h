/*
 * orc_trig.h -- the oracle's own acos / sin / cos (TEST INFRASTRUCTURE ONLY, see ransac_oracle.h).
 *
 * The cone code needs acos (fit3pointcone, cone.jl:58) and cos / sin of -opang/2 (rodrigues,
 * utilities.jl:21-22).  Julia's Base.acos / sin / cos are ports of the fdlibm family; this file restates
 * the fdlibm 5.3 algorithms (e_acos.c, k_sin.c, k_cos.c, s_sin.c, s_cos.c and the medium-size argument
 * reduction of e_rem_pio2.c) for the oracle, written here independently of the product's
 * ransac.jl_amd/csrc/det_math.h: the two are held against each other bit for bit by
 * tests/test_abi.py::test_deterministic_trig_is_within_one_ulp_of_libm and
 * tests/test_oracle_golden.py::test_oracle_trig_vs_libm, and both against the platform libm (<= 1 ulp).
 * ORC_VARIANT=5 swaps these for the platform libm to measure what that last ulp can change.
 *
 *   ====================================================
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *   Developed at SunSoft, a Sun Microsystems, Inc. business.
 *   Permission to use, copy, modify, and distribute this
 *   software is freely granted, provided that this notice
 *   is preserved.
 *   ====================================================
 */
#ifndef ORC_TRIG_H
#define ORC_TRIG_H

#include <math.h>
#include <stdint.h>

typedef union { double f; struct { uint32_t lo, hi; } w; } orc_dw;   /* little-endian binary64 */

static inline int32_t orc_hiword(double x) { orc_dw u; u.f = x; return (int32_t)u.w.hi; }
static inline double orc_with_words(uint32_t hi, uint32_t lo) { orc_dw u; u.w.hi = hi; u.w.lo = lo; return u.f; }

/* rational approximation of (asin(sqrt(z)) - sqrt(z)) / sqrt(z)^3 used by e_acos.c: P(z) / Q(z) */
static inline double orc_acos_R(double z)
{
    static const double P[6] = { 1.66666666666666657415e-01, -3.25565818622400915405e-01, 2.01212532134862925881e-01,
                                 -4.00555345006794114027e-02, 7.91534994289814532176e-04, 3.47933107596021167570e-05 };
    static const double Q[5] = { 1.0, -2.40339491173441421878e+00, 2.02094576023350569471e+00,
                                 -6.88283971605453293030e-01, 7.70381505559019352791e-02 };
    double p = P[5], q = Q[4];
    for (int i = 4; i >= 0; i--) p = P[i] + z * p;
    p = z * p;
    for (int i = 3; i >= 0; i--) q = Q[i] + z * q;
    return p / q;
}

/* __ieee754_acos */
static inline double orc_acos(double x)
{
    const double pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00,
                 pio2_lo = 6.12323399573676603587e-17;
    const int32_t hx = orc_hiword(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x3ff00000) {                      /* |x| >= 1 (or NaN) */
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);
    }
    if (ix < 0x3fe00000) {                       /* |x| < 1/2 */
        if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
        return pio2_hi - (x - (pio2_lo - x * orc_acos_R(x * x)));
    }
    if (hx < 0) {                                /* -1 < x <= -1/2 */
        const double z = (1.0 + x) * 0.5;
        const double s = sqrt(z);
        const double w = orc_acos_R(z) * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    /* 1/2 <= x < 1 */
    const double z = (1.0 - x) * 0.5;
    const double s = sqrt(z);
    orc_dw d; d.f = s; d.w.lo = 0;               /* df = s with the low word cleared */
    const double df = d.f;
    const double c = (z - df * df) / (s + df);
    const double w = orc_acos_R(z) * s + c;
    return 2.0 * (df + w);
}

/* __kernel_sin(x, y, iy): |x| <= pi/4, y the tail of x, iy = 0 when y is exactly 0 */
static inline double orc_ksin(double x, double y, int iy)
{
    static const double S[7] = { 0.0, -1.66666666666666324348e-01, 8.33333333332248946124e-03,
                                 -1.98412698298579493134e-04, 2.75573137070700676789e-06,
                                 -2.50507602534068634195e-08, 1.58969099521155010221e-10 };
    const double z = x * x;
    const double v = z * x;
    double r = S[6];
    for (int i = 5; i >= 2; i--) r = S[i] + z * r;
    if (iy == 0) return x + v * (S[1] + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S[1]);
}

/* __kernel_cos(x, y) */
static inline double orc_kcos(double x, double y)
{
    static const double Cc[7] = { 0.0, 4.16666666666666019037e-02, -1.38888888888741095749e-03,
                                  2.48015872894767294178e-05, -2.75573143513906633035e-07,
                                  2.08757232129817482790e-09, -1.13596475577881948265e-11 };
    const int32_t ix = orc_hiword(x) & 0x7fffffff;
    if (ix < 0x3e400000) return 1.0;             /* |x| < 2^-27: cos(x) rounds to 1 */
    const double z = x * x;
    double r = Cc[6];
    for (int i = 5; i >= 1; i--) r = Cc[i] + z * r;
    r = z * r;
    const double tail = z * r - x * y;
    if (ix < 0x3FD33333) return 1.0 - (0.5 * z - tail);   /* |x| < 0.3 */
    const double qx = ix > 0x3fe90000 ? 0.28125 : orc_with_words((uint32_t)(ix - 0x00200000), 0u);   /* ~ x/4 */
    const double hz = 0.5 * z - qx;
    return (1.0 - qx) - (hz - tail);
}

/* x - n*pi/2 for 3pi/4 < |x| < 2^19 pi/2 as head + tail; returns n (sign follows x) */
static inline int orc_rem_pio2_medium(double x, double *head, double *tail)
{
    /* pi/2 in three 33-bit pieces with their remainders (e_rem_pio2.c) */
    static const double piece[3] = { 1.57079632673412561417e+00, 6.07710050630396597660e-11, 2.02226624871116645580e-21 };
    static const double rest[3] = { 6.07710050650619224932e-11, 2.02226624879595063154e-21, 8.47842766036889956997e-32 };
    static const int need[3] = { 16, 49, 1 << 30 };     /* lost-bit thresholds that ask for the next piece */
    const double invpio2 = 6.36619772367581382433e-01;
    const int32_t hx = orc_hiword(x);
    const int e0 = (hx & 0x7fffffff) >> 20;
    const double ax = fabs(x);
    const int n = (int)(ax * invpio2 + 0.5);
    const double fn = (double)n;
    double r = ax - fn * piece[0];
    double w = fn * rest[0];
    double y0 = r - w;
    for (int k = 0; k < 2; k++) {
        const int lost = e0 - ((orc_hiword(y0) >> 20) & 0x7ff);
        if (lost <= need[k]) break;
        const double t = r;
        w = fn * piece[k + 1];
        r = t - w;
        w = fn * rest[k + 1] - ((t - r) - w);
        y0 = r - w;
    }
    const double y1 = (r - y0) - w;
    if (hx < 0) { *head = -y0; *tail = -y1; return -n; }
    *head = y0; *tail = y1;
    return n;
}

static inline double orc_sin(double x)
{
    const int32_t ix = orc_hiword(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return orc_ksin(x, 0.0, 0);
    if (ix >= 0x413921fb) return sin(x);         /* outside the medium reduction, inf, NaN: platform libm */
    double y0, y1;
    switch (orc_rem_pio2_medium(x, &y0, &y1) & 3) {
    case 0: return orc_ksin(y0, y1, 1);
    case 1: return orc_kcos(y0, y1);
    case 2: return -orc_ksin(y0, y1, 1);
    default: return -orc_kcos(y0, y1);
    }
}

static inline double orc_cos(double x)
{
    const int32_t ix = orc_hiword(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) return orc_kcos(x, 0.0);
    if (ix >= 0x413921fb) return cos(x);
    double y0, y1;
    switch (orc_rem_pio2_medium(x, &y0, &y1) & 3) {
    case 0: return orc_kcos(y0, y1);
    case 1: return -orc_ksin(y0, y1, 1);
    case 2: return -orc_kcos(y0, y1);
    default: return orc_ksin(y0, y1, 1);
    }
}

#endif
