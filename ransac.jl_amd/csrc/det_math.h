// det_math.h -- acos / sin / cos built from + - * / sqrt only, so that host code, device code and
// the oracle's C twin return the SAME bits (libm's results differ between glibc and the device
// library in the last place, which would make a cone fitted on the device differ from the same cone
// fitted on the host).  The algorithms and constants are those of FreeBSD msun / fdlibm
// (e_acos.c, k_sin.c, k_cos.c, e_rem_pio2.c medium path), which is also what Julia's own
// Base.acos / sin / cos are ports of; error < 1 ulp.
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//   Developed at SunSoft, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RH_DM __host__ __device__ inline
#else
#define RH_DM static inline
#endif

RH_DM uint32_t rh_dm_hi(double x)
{
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return (uint32_t)(u >> 32);
}
RH_DM double rh_dm_clear_lo(double x)
{
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    u &= 0xFFFFFFFF00000000ULL;
    double r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}
RH_DM double rh_dm_from_hi(uint32_t hi)
{
    uint64_t u = (uint64_t)hi << 32;
    double r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}

// ---- e_acos.c ----
RH_DM double rh_acos(double x)
{
    const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00,
                 pio2_lo = 6.12323399573676603587e-17, pS0 = 1.66666666666666657415e-01,
                 pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04,
                 pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00,
                 qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    const uint32_t hx = rh_dm_hi(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {   // |x| >= 1
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x);   // NaN
    }
    if (ix < 0x3fe00000u) {   // |x| < 0.5
        if (ix <= 0x3c600000u) return pio2_hi + pio2_lo;
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx & 0x80000000u) {   // x < -0.5
        const double z = (one + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    {   // x > 0.5
        const double z = (one - x) * 0.5;
        const double s = sqrt(z);
        const double df = rh_dm_clear_lo(s);
        const double c = (z - df * df) / (s + df);
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        const double w = r * s + c;
        return 2.0 * (df + w);
    }
}

// ---- k_sin.c / k_cos.c : |x| <= pi/4, y = tail of x ----
RH_DM double rh_dm_ksin(double x, double y, int iy)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

RH_DM double rh_dm_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const uint32_t ix = rh_dm_hi(x) & 0x7fffffffu;
    if (ix < 0x3e400000u) return 1.0;   // |x| < 2^-27
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
    const double qx = ix > 0x3fe90000u ? 0.28125 : rh_dm_from_hi(ix - 0x00200000u);
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

// ---- e_rem_pio2.c, medium path: valid for |x| < 2^19 * pi/2; returns n mod 4, y0 + y1 = x - n*pi/2 ----
RH_DM int rh_dm_rem_pio2(double x, double *y0, double *y1)
{
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                 pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                 pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                 pio2_3t = 8.47842766036889956997e-32;
    const uint32_t hx = rh_dm_hi(x), ix = hx & 0x7fffffffu;
    const double t0 = fabs(x);
    const int n = (int)(t0 * invpio2 + 0.5);
    const double fn = (double)n;
    double r = t0 - fn * pio2_1;
    double w = fn * pio2_1t;
    const int j = (int)(ix >> 20);
    double a = r - w;
    int i = j - (int)((rh_dm_hi(a) >> 20) & 0x7ff);
    if (i > 16) {   // 2nd iteration, good to 118 bits
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        a = r - w;
        i = j - (int)((rh_dm_hi(a) >> 20) & 0x7ff);
        if (i > 49) {   // 3rd iteration, 151 bits
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            a = r - w;
        }
    }
    const double b = (r - a) - w;
    if (hx & 0x80000000u) { *y0 = -a; *y1 = -b; return (-n) & 3; }
    *y0 = a; *y1 = b;
    return n & 3;
}

RH_DM double rh_sin(double x)
{
    const uint32_t ix = rh_dm_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return rh_dm_ksin(x, 0.0, 0);   // |x| <= pi/4
    if (ix >= 0x413921fbu) return sin(x);                  // beyond the medium reduction (or inf / NaN): libm
    double y0, y1;
    switch (rh_dm_rem_pio2(x, &y0, &y1)) {
    case 0: return rh_dm_ksin(y0, y1, 1);
    case 1: return rh_dm_kcos(y0, y1);
    case 2: return -rh_dm_ksin(y0, y1, 1);
    default: return -rh_dm_kcos(y0, y1);
    }
}

RH_DM double rh_cos(double x)
{
    const uint32_t ix = rh_dm_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return rh_dm_kcos(x, 0.0);
    if (ix >= 0x413921fbu) return cos(x);
    double y0, y1;
    switch (rh_dm_rem_pio2(x, &y0, &y1)) {
    case 0: return rh_dm_kcos(y0, y1);
    case 1: return -rh_dm_ksin(y0, y1, 1);
    case 2: return -rh_dm_kcos(y0, y1);
    default: return rh_dm_ksin(y0, y1, 1);
    }
}
