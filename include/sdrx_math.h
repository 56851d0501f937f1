/*
 * sdrx_math.h -- portable, bit-reproducible fp64 sincos / atan2.
 *
 * Why this exists: the reference's PSK carrier loop calls libm cos/sin per
 * sample (RadioNode/src/libraries/streams/BPSK.cpp:403-404, QPSK.cpp:365-366)
 * and the phase-difference FM discriminator calls libm atan2
 * (RadioNode/third-party/RTLSDR/rtl_fm.c:428-435).  glibc on the host and
 * ocml on the GPU do not return bit-identical results for those, and the
 * loops feed hard decisions, so the device path and its CPU checker must use
 * ONE definition.  Everything below is built only from IEEE-754 +,-,*,/ ,
 * fma (written out, in sincos), rint and comparisons, evaluated in source
 * order; compile it with -ffp-contract=off on every target (gcc for the
 * host, hipcc for gfx950) and the two sides agree to the last bit.  Accuracy against libm: <= 2 ulp on
 * the reduced range, |x| < 1.6e6 for sincos (the oracle's libm mode measures
 * the distance; see tests/test_oracle_golden.py).
 *
 * Polynomial coefficients and the split-pi constants are the classic
 * fdlibm / FreeBSD msun ones (Copyright (C) 1993 by Sun Microsystems, Inc.
 * "Permission to use, copy, modify, and distribute this software is freely
 * granted, provided that this notice is preserved.").  The evaluation code is
 * written for this project.
 */
#ifndef SDRX_MATH_H
#define SDRX_MATH_H

#include <math.h>

#if defined(__HIPCC__)
#define SDRX_HD __host__ __device__ static inline
#else
#define SDRX_HD static inline
#endif

/* ---- sincos ------------------------------------------------------------- */

/* fused multiply-add, one rounding: an IEEE-754 operation like the others (v_fma_f64 on the GPU, libm's correctly
 * rounded fma() or a hardware FMA on a host): the same bits everywhere.  Written out where it is meant; the rest of
 * the file stays uncontracted (-ffp-contract=off). */
#define SDRX_FMA(a, b, c) __builtin_fma((a), (b), (c))

/* sin(r) on |r| <= pi/4 */
SDRX_HD double sdrx_ksin(double r)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = r * r;
    double p = SDRX_FMA(z, S6, S5);
    p = SDRX_FMA(z, p, S4);
    p = SDRX_FMA(z, p, S3);
    p = SDRX_FMA(z, p, S2);
    p = SDRX_FMA(z, p, S1);
    return SDRX_FMA(z * r, p, r);
}

/* cos(r) on |r| <= pi/4 */
SDRX_HD double sdrx_kcos(double r)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = r * r;
    double p = SDRX_FMA(z, C6, C5);
    p = SDRX_FMA(z, p, C4);
    p = SDRX_FMA(z, p, C3);
    p = SDRX_FMA(z, p, C2);
    p = SDRX_FMA(z, p, C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    /* (1 - w) - hz is the rounding error of w, recovered exactly */
    return w + SDRX_FMA(z * z, p, (1.0 - w) - hz);
}

/* The common case, straight-line: s = sin(x), c = cos(x) for |x| <= 1.5e6.  Returns nonzero -- and leaves
 * unspecified finite-or-not values in *s, *c -- for every other x (huge, infinite, NaN): the caller then takes
 * sdrx_sincos_rare.  The PSK carrier loop, where this is the longest part of a serial chain, runs a block of samples
 * on this alone, ORs the flags, and repeats the block through sdrx_sincos if any was set (it never is unless a loop
 * has run away). */
SDRX_HD int sdrx_sincos_core(double x, double *s, double *c)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    /* pi/2 in 33 + 33 + 53 bits: fn*P1 and fn*P2 are exact for |fn| < 2^20 */
    const double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P3 = 2.02226624871116645580e-21;
    const double fn = rint(x * INV_PIO2);
    double r = SDRX_FMA(-fn, P1, x);
    r = SDRX_FMA(-fn, P2, r);
    r = SDRX_FMA(-fn, P3, r);
    const int k = (int)fn & 3; /* |fn| < 2^20 whenever the result is used */
    const double sr = sdrx_ksin(r);
    const double cr = sdrx_kcos(r);
    const double ss = (k & 1) ? cr : sr; /* k = 0: s =  sr, c =  cr */
    const double cc = (k & 1) ? sr : cr; /* k = 1: s =  cr, c = -sr */
    *s = (k & 2) ? -ss : ss;             /* k = 2: s = -sr, c = -cr */
    *c = ((k + 1) & 2) ? -cc : cc;       /* k = 3: s = -cr, c =  sr */
    return !(__builtin_fabs(x) <= 1.5e6);
}

/* |x| > 1.5e6, infinite or NaN */
SDRX_HD void sdrx_sincos_rare(double x, double *s, double *c)
{
    /* 2^20 * 2*pi split in three pieces with 26 trailing zero bits each (coarse stage for huge |x|) */
    const double M1 = 6588397.0, M2 = 0.31666114181280136 /* 0x1.4442d18p-2 */, M3 = 2.56827048412547e-10;
    if (!(__builtin_fabs(x) < 1.0e300)) { /* inf / nan / absurd */
        *s = x - x;
        *c = x - x;
        return;
    }
    const double q = rint(x * 1.5178198155583891e-07); /* 1/(2^20*2pi) */
    (void)sdrx_sincos_core(((x - q * M1) - q * M2) - q * M3, s, c);
}

/* s = sin(x), c = cos(x).  Same bits on every IEEE-754 target. */
SDRX_HD void sdrx_sincos(double x, double *s, double *c)
{
    if (sdrx_sincos_core(x, s, c)) sdrx_sincos_rare(x, s, c);
}

/* ---- sincos, table form (the PSK carrier loop) ------------------------------ */

/*
 * sdrx_sincos_tab: the same job as sdrx_sincos in a third of the operations and a much shorter dependent chain, for the one
 * caller where sincos sits inside a serial per-sample recurrence (BPSK.cpp:403-404, QPSK.cpp:365-366: the carrier loop).
 *   x = k * (pi/128) + r,  |r| <= pi/256:   sin x = (S_k + S_k (cos r - 1)) + C_k sin r,   cos x = (C_k + C_k (cos r - 1)) - S_k sin r
 *   (each bracket and each sum one fma: two roundings of values below 1, so within 1 ulp of the true value)
 * with (S_k, C_k) = (sin, cos)(k pi/128) from a 256-entry table (include/sdrx_sincos_table.h, written by
 * sdr_amd/tools/gen_sincos_table.py; hexadecimal constants, so every compiler reads the same doubles) and Taylor polynomials
 * of degree 5 / 6 on the small argument (next terms r^7/7! < 9e-18, r^8/8! < 2e-20).  Reduction: t = fma(x, 128/pi, 1.5 2^52)
 * holds the nearest integer fn to x 128/pi in its low significand bits (k = those bits mod 256, no conversion), fn = t - 1.5 2^52,
 * r = fma(-fn, Q2, fma(-fn, Q1, x)) with pi/128 = Q1 + Q2 to 2^-85: Q1 has 26 significant bits, so fn Q1 and the first
 * difference are exact for |fn| < 2^27, and the second fma rounds once.  The result is within 1 ulp of the true value for |x| <= 1.5e6
 * (test_oracle_golden.py measures it against libm).  Like everything in this file: IEEE operations in source order, the same bits on the host and on gfx950.
 * `tab`: 512 doubles {S_0, C_0, S_1, C_1, ...}; on the host SDRX_SINCOS_TAB, on the GPU the kernel's copy in LDS.
 */
#include "sdrx_sincos_table.h"

#if !defined(__HIP_DEVICE_COMPILE__)
static const double SDRX_SINCOS_TAB[2 * SDRX_SINCOS_TAB_N] = {SDRX_SINCOS_TAB_VALUES};
#endif

/* The common case, straight-line (|x| <= 1.5e6); returns nonzero for every other x, as sdrx_sincos_core does. */
SDRX_HD int sdrx_sincos_tab_core(double x, const double *tab, double *s, double *c)
{
    const double MAGIC = 0x1.8p+52;
    const double S1 = -1.0 / 6.0, S2 = 1.0 / 120.0;
    const double C1 = 1.0 / 24.0, C2 = -1.0 / 720.0;
    const double t = SDRX_FMA(x, SDRX_SINCOS_INV_STEP, MAGIC);
    const double fn = t - MAGIC;
    unsigned long long tb;
    __builtin_memcpy(&tb, &t, sizeof(tb));
    const unsigned k = (unsigned)tb & (SDRX_SINCOS_TAB_N - 1); /* fn mod 256 (two's complement: right for negative fn too) */
    const double Sk = tab[2 * k], Ck = tab[2 * k + 1];
    const double r = SDRX_FMA(-fn, SDRX_SINCOS_Q2, SDRX_FMA(-fn, SDRX_SINCOS_Q1, x));
    const double z = r * r;
    const double ps = SDRX_FMA(z, S2, S1);
    const double sr = SDRX_FMA(z * r, ps, r); /* sin r */
    double pc = SDRX_FMA(z, C2, C1);
    pc = SDRX_FMA(z, pc, -0.5);
    const double cm = z * pc; /* cos r - 1 */
    *s = SDRX_FMA(Ck, sr, SDRX_FMA(Sk, cm, Sk)); /* S_k cos r + C_k sin r: two roundings */
    *c = SDRX_FMA(-Sk, sr, SDRX_FMA(Ck, cm, Ck));
    return !(__builtin_fabs(x) <= 1.5e6);
}

/* s = sin(x), c = cos(x) through the table form; huge / non-finite x as in sdrx_sincos_rare. */
SDRX_HD void sdrx_sincos_tab(double x, const double *tab, double *s, double *c)
{
    if (sdrx_sincos_tab_core(x, tab, s, c)) {
        const double M1 = 6588397.0, M2 = 0.31666114181280136 /* 0x1.4442d18p-2 */, M3 = 2.56827048412547e-10;
        if (!(__builtin_fabs(x) < 1.0e300)) { /* inf / nan / absurd */
            *s = x - x;
            *c = x - x;
            return;
        }
        const double q = rint(x * 1.5178198155583891e-07); /* 1/(2^20*2pi) */
        (void)sdrx_sincos_tab_core(((x - q * M1) - q * M2) - q * M3, tab, s, c);
    }
}

/* ---- atan / atan2 ------------------------------------------------------- */

/* atan(ay/ax) for finite ay >= 0, ax > 0.  The classic 4-breakpoint reduction
 * atan(x) = atan(c) + atan((x - c)/(1 + c x)), c in {0, 1/2, 1, 3/2, inf}, written on the
 * PAIR (ay, ax) so that it costs one division instead of two, and with selects instead of
 * branches so that a 64-lane wave does not execute every interval in turn. */
SDRX_HD double sdrx_atan_ratio(double ay, double ax)
{
    const double HI0 = 4.63647609000806093515e-01, HI1 = 7.85398163397448278999e-01,
                 HI2 = 9.82793723247329054082e-01, HI3 = 1.57079632679489655800e+00;
    const double LO0 = 2.26987774529616870924e-17, LO1 = 3.06161699786838301793e-17,
                 LO2 = 1.39033110312309984516e-17, LO3 = 6.12323399573676603587e-17;
    const double A0 = 3.33333333333329318027e-01, A1 = -1.99999999998764832476e-01,
                 A2 = 1.42857142725034663711e-01, A3 = -1.11111104054623557880e-01,
                 A4 = 9.09088713343650656196e-02, A5 = -7.69187620504482999495e-02,
                 A6 = 6.66107313738753120669e-02, A7 = -5.83357013379057348645e-02,
                 A8 = 4.97687799461593236017e-02, A9 = -3.65315727442169155270e-02,
                 A10 = 1.62858201153657823623e-02;
    /* interval of ay/ax: [0,7/16) [7/16,11/16) [11/16,19/16) [19/16,39/16) [39/16,inf) */
    const int i1 = ay >= 0.4375 * ax, i2 = ay >= 0.6875 * ax, i3 = ay >= 1.1875 * ax, i4 = ay >= 2.4375 * ax;
    double num = ay, den = ax, hi = 0.0, lo = 0.0;
    num = i1 ? (2.0 * ay - ax) : num;  den = i1 ? (2.0 * ax + ay) : den;  hi = i1 ? HI0 : hi;  lo = i1 ? LO0 : lo;
    num = i2 ? (ay - ax) : num;        den = i2 ? (ay + ax) : den;        hi = i2 ? HI1 : hi;  lo = i2 ? LO1 : lo;
    num = i3 ? (ay - 1.5 * ax) : num;  den = i3 ? (ax + 1.5 * ay) : den;  hi = i3 ? HI2 : hi;  lo = i3 ? LO2 : lo;
    num = i4 ? -ax : num;              den = i4 ? ay : den;               hi = i4 ? HI3 : hi;  lo = i4 ? LO3 : lo;
    const double t = num / den;
    const double z = t * t;
    const double w = z * z;
    const double s1 = z * (A0 + w * (A2 + w * (A4 + w * (A6 + w * (A8 + w * A10)))));
    const double s2 = w * (A1 + w * (A3 + w * (A5 + w * (A7 + w * A9))));
    /* hi == lo == 0 in the first interval: hi - ((t*s - 0) - t) == t - t*s */
    return hi - ((t * (s1 + s2) - lo) - t);
}

/* atan2(y, x), C99 sign/zero conventions; nan in -> nan out.  Straight-line: the general path always runs
 * and the special cases (zeros, infinities, nan) are selected over it afterwards, so a 64-lane wave executes
 * no divergent branch (the arithmetic of every case is what the branching form did: same bits). */
SDRX_HD double sdrx_atan2(double y, double x)
{
    const double PI = 3.1415926535897931160e+00, PI_LO = 1.2246467991473531772e-16;
    const double PIO2 = 1.5707963267948965580e+00, PIO4 = 7.85398163397448278999e-01;
    const int yneg = __builtin_signbit(y) != 0, xneg = __builtin_signbit(x) != 0;
    const double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    const int xinf = ax > 1.79769313486231570815e+308;
    const int yinf = ay > 1.79769313486231570815e+308;
    double z = sdrx_atan_ratio(ay, ax); /* (0,0), infinities: overridden below */
    z = (xinf || yinf) ? ((xinf && yinf) ? PIO4 : (yinf ? PIO2 : 0.0)) : z;
    double r = xneg ? PI - (z - PI_LO) : z; /* second / third quadrant: pi - z, carrying pi's tail */
    r = yneg ? -r : r;
    r = (x == 0.0) ? (yneg ? -PIO2 : PIO2) : r;
    r = (y == 0.0) ? (xneg ? (yneg ? -PI : PI) : y) : r; /* +-0 for x >= +0, +-pi for x <= -0 */
    r = (x != x || y != y) ? x + y : r;
    return r;
}

#endif /* SDRX_MATH_H */
