/*
 * orc_math.h — scalar fp32 arithmetic of the ORACLE (test infrastructure only).
 *
 * TEST INFRASTRUCTURE: nothing under oracle/ is linked, imported or executed by
 * the product (yetanotherconsolegameengine_amd/).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * PARITY UNPINNED: the reference (C#/.NET 8) has no tests, golden vectors or
 * fixtures for this path and cannot be built in this image (no dotnet/mono),
 * so this restatement is pinned only by derived known-answer tests
 * (tests/test_oracle_kats.py) and by line-by-line review against the cited
 * reference lines.
 *
 * C# semantics carried here (RyuJIT x64): every fp32 operation is rounded to
 * binary32 (SSE scalar), no FMA contraction, left-to-right evaluation;
 * `/` and MathF.Sqrt are correctly rounded; MathF.Max/Min propagate NaN;
 * (int)float is cvttss2si (out of range / NaN -> INT_MIN).
 * Build with: g++ -O2 -ffp-contract=off -fno-fast-math.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

static const float kFloatMax = std::numeric_limits<float>::max();
static const float kInf = std::numeric_limits<float>::infinity();

/* .NET MathF.Max (System.Private.CoreLib, .NET 8 — IEEE 754:2019 `maximum`):
 * NaN in either operand comes back; +0 > -0. */
static inline float cs_max(float a, float b)
{
    if (a != b) {
        if (!(a != a)) return b < a ? a : b;
        return a;
    }
    return std::signbit(b) ? a : b;
}
/* .NET MathF.Min (`minimum`). */
static inline float cs_min(float a, float b)
{
    if (a != b) {
        if (!(a != a)) return a < b ? a : b;
        return a;
    }
    return std::signbit(a) ? a : b;
}
static inline double cs_min_d(double a, double b)
{
    if (a != b) {
        if (!(a != a)) return a < b ? a : b;
        return a;
    }
    return std::signbit(a) ? a : b;
}
/* C# (int)f on x64 = cvttss2si: truncation; NaN / out of range -> 0x80000000. */
static inline int32_t cs_f2i(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}
static inline float cs_abs(float f) { return std::fabs(f); }
static inline float cs_copysign(float mag, float sgn) { return std::copysign(mag, sgn); }
static inline float cs_floor(float f) { return std::floor(f); }
static inline float cs_sqrt(float f) { return std::sqrt(f); }   /* sqrtss: correctly rounded */
static inline bool cs_isfinite(float f) { return std::isfinite(f); }
/* Math.Clamp(float) : value < min -> min, value > max -> max, NaN passes */
static inline float cs_clamp(float v, float lo, float hi)
{
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* ------------------------------------------------------------------------
 * Transcendentals.  The reference forwards MathF.SinCos / Pow / Exp / Log to
 * the platform C runtime (ucrt on its Windows target), which is not
 * bit-reproducible elsewhere.  The oracle and the HIP kernels therefore both
 * evaluate the SAME published algorithm below — binary64 polynomial kernels
 * built from + - * only (no FMA, no libm), result rounded once to binary32 —
 * which is within 1 ulp of any faithful fp32 libm.  The radiance tolerance
 * of north_star (1e-4 RMS) covers the difference to ucrt.
 * ---------------------------------------------------------------------- */

/* sin and cos of a binary32 angle |x| <= ~8 (phi = 2*pi*u2, RaytraceSampler.cs:88-89).
 * Cody-Waite reduction by pi/2 in binary64 then Taylor kernels on [-pi/4, pi/4]. */
static inline void m_sincos(float xf, float *s_out, float *c_out)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11;  /* pi/2 - pio2_hi */
    double x = (double)xf;
    double kq = x * two_over_pi;
    /* round to nearest integer without libm: valid for |kq| < 2^31 */
    int q = (int)(kq < 0.0 ? kq - 0.5 : kq + 0.5);
    double dq = (double)q;
    double r = (x - dq * pio2_hi) - dq * pio2_lo;
    double r2 = r * r;
    /* sin r = r + r^3 * P(r^2) */
    double ps = -7.6471637318198164759e-13;              /* -1/15! */
    ps = ps * r2 + 1.6059043836821614599e-10;            /*  1/13! */
    ps = ps * r2 + -2.5052108385441718775e-08;           /* -1/11! */
    ps = ps * r2 + 2.7557319223985890653e-06;            /*  1/9!  */
    ps = ps * r2 + -1.9841269841269841270e-04;           /* -1/7!  */
    ps = ps * r2 + 8.3333333333333333333e-03;            /*  1/5!  */
    ps = ps * r2 + -1.6666666666666666667e-01;           /* -1/3!  */
    double sr = r + (r * r2) * ps;
    /* cos r = 1 + r^2 * Q(r^2) */
    double pc = 4.7794773323873852974e-14;               /*  1/16! */
    pc = pc * r2 + -1.1470745597729724714e-11;           /* -1/14! */
    pc = pc * r2 + 2.0876756987868098979e-09;            /*  1/12! */
    pc = pc * r2 + -2.7557319223985890653e-07;           /* -1/10! */
    pc = pc * r2 + 2.4801587301587301587e-05;            /*  1/8!  */
    pc = pc * r2 + -1.3888888888888888889e-03;           /* -1/6!  */
    pc = pc * r2 + 4.1666666666666666667e-02;            /*  1/4!  */
    pc = pc * r2 + -5.0000000000000000000e-01;           /* -1/2!  */
    double cr = 1.0 + r2 * pc;
    double s, c;
    switch (q & 3) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

/* MathF.Pow(x, 5.0f) for the Schlick term (RaytraceRenderer.cs:754): exact
 * product in binary64 (x has 24 bits; x^2, x^4 exact or 1-ulp in binary64),
 * rounded once. */
static inline float m_pow5(float xf)
{
    double x = (double)xf;
    double x2 = x * x;
    double x4 = x2 * x2;
    return (float)(x4 * x);
}

/* e^x for binary32 x (À-trous weights RaytraceRenderer.cs:694-697, exposure
 * ToneMapper.cs:82,87).  x = k ln2 + r, |r| <= ln2/2, Taylor degree 13,
 * scale by 2^k through the exponent field. */
static inline double m_exp_d(double x)
{
    if (x != x) return x;
    if (x > 709.0) return std::numeric_limits<double>::infinity();
    if (x < -745.0) return 0.0;
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    double kf = x * inv_ln2;
    int k = (int)(kf < 0.0 ? kf - 0.5 : kf + 0.5);
    double dk = (double)k;
    double r = (x - dk * ln2_hi) - dk * ln2_lo;
    /* Taylor degree 13 in Estrin form (round 4; Horner before): the same coefficients, a dependency chain of 8 operations instead of 26 - this
     * polynomial sits on the in-place A-trous iteration's serial chain once per pixel level.  Every product and sum below is one binary64
     * operation (no contraction), in this order, in the oracle and in the kernels alike.  exp(-0) is still exactly 1: r = -0 makes r2 = +0,
     * every pair (c + c' * r) is its first coefficient and every higher term a zero. */
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = 1.0 + 1.0 * r;
    const double a1 = 5.0000000000000000000e-01 + 1.6666666666666666667e-01 * r;
    const double a2 = 4.1666666666666666667e-02 + 8.3333333333333333333e-03 * r;
    const double a3 = 1.3888888888888888889e-03 + 1.9841269841269841270e-04 * r;
    const double a4 = 2.4801587301587301587e-05 + 2.7557319223985890653e-06 * r;
    const double a5 = 2.7557319223985890653e-07 + 2.5052108385441718775e-08 * r;
    const double a6 = 2.0876756987868098979e-09 + 1.6059043836821614599e-10 * r;
    const double q0 = a0 + a1 * r2, q1 = a2 + a3 * r2, q2 = a4 + a5 * r2;
    const double h0 = q0 + q1 * r4, h1 = q2 + a6 * r4;
    double p = h0 + h1 * r8;
    /* multiply by 2^k in two steps so that subnormal results round once more
     * at most; k in [-1075, 1023] */
    int k1 = k / 2, k2 = k - k1;
    uint64_t b1 = (uint64_t)(int64_t)(k1 + 1023) << 52;
    uint64_t b2 = (uint64_t)(int64_t)(k2 + 1023) << 52;
    double s1, s2;
    std::memcpy(&s1, &b1, 8);
    std::memcpy(&s2, &b2, 8);
    return (p * s1) * s2;
}
static inline float m_exp(float x) { return (float)m_exp_d((double)x); }

/* ln x for positive finite binary64 x: x = 2^e * m, m in [sqrt(1/2), sqrt 2),
 * ln m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^23. */
static inline double m_log_d(double x)
{
    if (x != x || x < 0.0) return std::numeric_limits<double>::quiet_NaN();
    if (x == 0.0) return -std::numeric_limits<double>::infinity();
    if (x == std::numeric_limits<double>::infinity()) return x;
    uint64_t b;
    std::memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) { /* subnormal: scale up by 2^54 */
        x = x * 18014398509481984.0;
        std::memcpy(&b, &x, 8);
        e = (int)((b >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    std::memcpy(&m, &b, 8);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    double de = (double)e;
    return (de * ln2_hi + (2.0 * s) * p) + de * ln2_lo;
}
static inline float m_log(float x) { return (float)m_log_d((double)x); }

/* MathF.Pow(x, y) for the gamma encode (ToneMapper.cs:215-217), x in [0,1], y > 0. */
static inline float m_pow(float x, float y)
{
    if (x == 0.0f) return 0.0f;
    if (x == 1.0f) return 1.0f;
    return (float)m_exp_d((double)y * m_log_d((double)x));
}

} // namespace orc
#endif
