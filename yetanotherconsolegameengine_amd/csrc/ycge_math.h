// ycge_math.h — fp32 arithmetic contract of the ray-trace core (host + gfx950).
//
// The reference computes in C# on x64: every fp32 operation rounds to binary32,
// nothing is fused, `/` and sqrt are correctly rounded, MathF.Max/Min propagate
// NaN, (int)float truncates with INT_MIN on overflow/NaN.  Results must match
// it bit for bit wherever only + - * / sqrt floor are involved, so:
//   * this code is ALWAYS compiled with -ffp-contract=off (no v_fma / v_fmac
//     contraction; HIP's default is `fast`);
//   * divisions and square roots use the IEEE-correct expansions hipcc emits by
//     default (-fhip-fp32-correctly-rounded-divide-sqrt); never __fdividef,
//     rsqrt, rcp or -ffast-math;
//   * min/max are written as the compare/select chains the C# source has, or
//     through cs_max/cs_min below when the source calls MathF.Max/Min.
// Transcendentals (sincos for the cosine-hemisphere sample, x^5 for Schlick,
// exp/log/pow for the post stage) are evaluated in binary64 with explicit
// polynomial kernels and rounded once to binary32, so host and device agree
// bit for bit and stay within 1 ulp of a faithful libm (the reference forwards
// to the platform CRT, which is not reproducible across platforms anyway).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define YCGE_HD __host__ __device__ __forceinline__
#else
#define YCGE_HD inline
#endif

namespace ycge {

YCGE_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
YCGE_HD float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

#define YCGE_FLT_MAX 3.402823466e+38f
#define YCGE_INF (ycge::u2f(0x7f800000u))

YCGE_HD bool is_nan(float a) { return a != a; }
YCGE_HD bool sign_bit(float a) { return (f2u(a) >> 31) != 0; }

// MathF.Max / MathF.Min (.NET 8: IEEE 754-2019 maximum / minimum)
YCGE_HD float cs_max(float a, float b)
{
    if (a != b) {
        if (!is_nan(a)) return b < a ? a : b;
        return a;
    }
    return sign_bit(b) ? a : b;
}
YCGE_HD float cs_min(float a, float b)
{
    if (a != b) {
        if (!is_nan(a)) return a < b ? a : b;
        return a;
    }
    return sign_bit(a) ? a : b;
}
YCGE_HD double cs_min_d(double a, double b)
{
    if (a != b) {
        if (!(a != a)) return a < b ? a : b;
        return a;
    }
    union { double d; uint64_t u; } c; c.d = a;
    return (c.u >> 63) ? a : b;
}
// (int)f on x64 (cvttss2si)
YCGE_HD int32_t cs_f2i(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return (int32_t)0x80000000;
    return (int32_t)f;
}
YCGE_HD float cs_abs(float f) { return u2f(f2u(f) & 0x7fffffffu); }
YCGE_HD float cs_copysign(float mag, float sgn) { return u2f((f2u(mag) & 0x7fffffffu) | (f2u(sgn) & 0x80000000u)); }
YCGE_HD float cs_floor(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_floorf(f);
#else
    return __builtin_floorf(f);
#endif
}
YCGE_HD float cs_sqrt(float f) { return __builtin_sqrtf(f); }
YCGE_HD bool cs_isfinite(float f) { return (f2u(f) & 0x7f800000u) != 0x7f800000u; }
YCGE_HD float cs_clamp(float v, float lo, float hi) { if (v < lo) return lo; if (v > hi) return hi; return v; }
YCGE_HD float clamp01(float v) { if (v < 0.0f) return 0.0f; if (v > 1.0f) return 1.0f; return v; }

// ---- binary64 kernels (see header comment) --------------------------------
YCGE_HD void m_sincos(float xf, float *s_out, float *c_out)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    double x = (double)xf;
    double kq = x * two_over_pi;
    int q = (int)(kq < 0.0 ? kq - 0.5 : kq + 0.5);
    double dq = (double)q;
    double r = (x - dq * pio2_hi) - dq * pio2_lo;
    double r2 = r * r;
    double ps = -7.6471637318198164759e-13;
    ps = ps * r2 + 1.6059043836821614599e-10;
    ps = ps * r2 + -2.5052108385441718775e-08;
    ps = ps * r2 + 2.7557319223985890653e-06;
    ps = ps * r2 + -1.9841269841269841270e-04;
    ps = ps * r2 + 8.3333333333333333333e-03;
    ps = ps * r2 + -1.6666666666666666667e-01;
    double sr = r + (r * r2) * ps;
    double pc = 4.7794773323873852974e-14;
    pc = pc * r2 + -1.1470745597729724714e-11;
    pc = pc * r2 + 2.0876756987868098979e-09;
    pc = pc * r2 + -2.7557319223985890653e-07;
    pc = pc * r2 + 2.4801587301587301587e-05;
    pc = pc * r2 + -1.3888888888888888889e-03;
    pc = pc * r2 + 4.1666666666666666667e-02;
    pc = pc * r2 + -5.0000000000000000000e-01;
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

YCGE_HD float m_pow5(float xf)
{
    double x = (double)xf;
    double x2 = x * x;
    double x4 = x2 * x2;
    return (float)(x4 * x);
}

YCGE_HD double bits_to_double(uint64_t b) { union { double d; uint64_t u; } c; c.u = b; return c.d; }
YCGE_HD uint64_t double_to_bits(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }

YCGE_HD double m_exp_d(double x)
{
    if (x != x) return x;
    if (x > 709.0) return bits_to_double(0x7ff0000000000000ULL);
    if (x < -745.0) return 0.0;
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    double kf = x * inv_ln2;
    int k = (int)(kf < 0.0 ? kf - 0.5 : kf + 0.5);
    double dk = (double)k;
    double r = (x - dk * ln2_hi) - dk * ln2_lo;
    // Taylor degree 13 in Estrin form (round 4; Horner before): the same coefficients, a dependency chain of 8 operations instead of 26 - this
    // polynomial sits on the in-place A-trous iteration's serial chain once per pixel level.  Every product and sum below is one binary64
    // operation (no contraction), in this order, in the oracle and in the kernels alike.  exp(-0) is still exactly 1: r = -0 makes r2 = +0,
    // every pair (c + c' * r) is its first coefficient and every higher term a zero.
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
    // p * 2^k.  In two steps (below) where 2^k alone would leave the normal range; everywhere else one exact scaling does the same
    if (k > -1000) return __builtin_ldexp(p, k);
    int k1 = k / 2, k2 = k - k1;
    double s1 = bits_to_double((uint64_t)(int64_t)(k1 + 1023) << 52);
    double s2 = bits_to_double((uint64_t)(int64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}
YCGE_HD float m_exp(float x) { return (float)m_exp_d((double)x); }

YCGE_HD double m_log_d(double x)
{
    if (x != x || x < 0.0) return bits_to_double(0x7ff8000000000000ULL);
    if (x == 0.0) return bits_to_double(0xfff0000000000000ULL);
    uint64_t b = double_to_bits(x);
    if (b == 0x7ff0000000000000ULL) return x;
    int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) {
        x = x * 18014398509481984.0;
        b = double_to_bits(x);
        e = (int)((b >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = bits_to_double(b);
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
YCGE_HD float m_log(float x) { return (float)m_log_d((double)x); }

YCGE_HD float m_pow(float x, float y)
{
    if (x == 0.0f) return 0.0f;
    if (x == 1.0f) return 1.0f;
    return (float)m_exp_d((double)y * m_log_d((double)x));
}

} // namespace ycge
