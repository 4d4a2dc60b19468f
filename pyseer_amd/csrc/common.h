// common.h -- shared device helpers for libseerhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/seerhip.h"

#define SH_WAVE 64

// ---------------------------------------------------------------------------------------------
// Tail functions (fp64).  Same mathematics as scipy's chdtrc / fdtrc / stdtr, which the reference calls at
// pyseer/model.py:53,68,339,369 and pyseer/lmm.py:251.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double sh_betacf(double a, double b, double x)
{
    const double TINY = 1e-300, EPS = 1e-16;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < TINY) d = TINY;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 20000; m++) {
        double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c; if (fabs(c) < TINY) c = TINY;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < TINY) d = TINY;
        c = 1.0 + aa / c; if (fabs(c) < TINY) c = TINY;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < EPS) break;
    }
    return h;
}

// I_x(a, 1/2) with ln x and ln(1-x) supplied by the caller (they are known to full relative accuracy there).
__device__ __forceinline__ double sh_betainc_half(double a, double x, double lnx, double ln1mx)
{
    if (isnan(x) || isnan(a)) return NAN;
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double b = 0.5;
    double lnB = lgamma(a) + 0.5723649429247001 /* lgamma(1/2) */ - lgamma(a + b);
    double front = exp(a * lnx + b * ln1mx - lnB);
    if (x < (a + 1.0) / (a + b + 2.0))
        return front * sh_betacf(a, b, x) / a;
    return 1.0 - front * sh_betacf(b, a, 1.0 - x) / b;
}

// scipy.stats.chi2.sf(x, 1)
__device__ __forceinline__ double sh_chi2_sf1(double x)
{
    if (isnan(x)) return NAN;
    if (x <= 0.0) return 1.0;
    return erfc(sqrt(0.5 * x));
}

// scipy.stats.f.sf(x, 1, d)
__device__ __forceinline__ double sh_f_sf_1(double x, double d)
{
    if (isnan(x)) return NAN;
    if (x <= 0.0) return 1.0;
    if (isinf(x)) return 0.0;
    double w = d / (d + x);
    return sh_betainc_half(0.5 * d, w, -log1p(x / d), log(x / (d + x)));
}

// 2 * scipy.stats.t.sf(|t|, df)
__device__ __forceinline__ double sh_t_sf2(double t, double df)
{
    if (isnan(t) || isnan(df)) return NAN;
    double t2 = t * t;
    if (t2 == 0.0) return 1.0;
    if (isinf(t2)) return 0.0;
    double w = df / (df + t2);
    return sh_betainc_half(0.5 * df, w, -log1p(t2 / df), log(t2 / (df + t2)));
}

// ---------------------------------------------------------------------------------------------
// a1 pre_filtering (pyseer/model.py:31-70) from integer table counts.
//   binary:     t11=#(y=1,k=1) t10=#(y=1,k=0) t01=#(y=0,k=1) t00=#(y=0,k=0)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double sh_prefilter_binary(int t11, int t10, int t01, int t00, bool *bad)
{
    int le1 = (t11 <= 1) + (t10 <= 1) + (t01 <= 1) + (t00 <= 1);
    int le5 = (t11 <= 5) + (t10 <= 5) + (t01 <= 5) + (t00 <= 5);
    *bad = (le1 > 0) || (le5 > 1);                                   // model.py:65
    double o[4] = {(double)t11, (double)t10, (double)t01, (double)t00};
    double r0 = o[0] + o[1], r1 = o[2] + o[3], c0 = o[0] + o[2], c1 = o[1] + o[3], tot = r0 + r1;
    double e[4] = {r0 * c0 / tot, r0 * c1 / tot, r1 * c0 / tot, r1 * c1 / tot};
    if (e[0] == 0 || e[1] == 0 || e[2] == 0 || e[3] == 0) return NAN;   // scipy raises ValueError here
    double chi2 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { double d = o[i] - e[i]; chi2 += d * d / e[i]; }
    return sh_chi2_sf1(chi2);                                        // chi2_contingency(correction=False), dof 1
}

// continuous: Welch t-test from group sums of the CENTRED phenotype (s = sum yc, q = sum yc^2 per group)
__device__ __forceinline__ double sh_prefilter_welch(double n1, double s1, double q1, double n0, double s0, double q0)
{
    double m1 = s1 / n1, m0 = s0 / n0;
    // np.var(ddof=1) of a single observation is 0/0 = NaN (two-pass: deviations are exactly 0)
    double v1 = n1 > 1.0 ? (q1 - s1 * m1) / (n1 - 1.0) : NAN, v0 = n0 > 1.0 ? (q0 - s0 * m0) / (n0 - 1.0) : NAN;
    double vn1 = v1 / n1, vn0 = v0 / n0;
    double df = (vn1 + vn0) * (vn1 + vn0) / (vn1 * vn1 / (n1 - 1.0) + vn0 * vn0 / (n0 - 1.0));
    if (isnan(df)) df = 1.0;
    double t = (m1 - m0) / sqrt(vn1 + vn0);
    return sh_t_sf2(t, df);
}
