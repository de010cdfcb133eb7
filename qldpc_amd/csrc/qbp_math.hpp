// FP64 elementary functions of the BP check-node update, written for the gfx950 vector ALU.
//
// The reference computes, per edge and iteration (decoding/beliefPropagation.py:114-126):
//     t = np.tanh(Q * 0.5)   ...   R = 2.0 * np.arctanh(clip(prod / t, +-0.9999999))
// gfx950 has no FP64 exp/log instruction, and the ROCm device-library tanh()/atanh() spend
// most of their instructions on argument ranges this path never sees.  These two functions
// are the same mathematical functions, accurate to 1-2 ulp (tests/test_math_cpu.py checks
// them against mpmath on the host build; tests/test_gpu_math.py on the device), built from:
//   * one v_rcp_f64 + Newton division each (operands are always normal: no v_div_scale /
//     v_div_fixup range handling needed),
//   * tanh(q/2) = (1 - E) / (1 + E), E = exp(-|q|) = 2^k (1 + tau)/(1 - tau), tau = tanh(h) from
//     a 7-term odd polynomial on |h| <= ln2/4: numerator and denominator are one fma each around
//     the exact constants 1 -+ 2^k (no cancellation for small |q|: k = 0 gives -2 tau / 2),
//   * 2 atanh(y) = log((1+y)/(1-y)) = e ln2 + 2 atanh(s), s = (N - 2^e D) / (N + 2^e D),
//     N = 1 + |y|, D = 1 - |y|: ONE division instead of the two of log1p(2y/(1-y)); numerator
//     and denominator are single fmas of |y| with exact coefficients 1 -+ 2^e, and s == |y|
//     exactly when e == 0 (small messages keep full relative accuracy).
// Coefficients: tools/fit_math_coeffs.py (mpmath near-minimax fits).
//
// The header also compiles on the host (plain C++), where the reciprocal seed is 1.0 / b:
// that build exists only so the CPU test suite can measure the ulp error without a GPU.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIP_DEVICE_COMPILE__)
#define QBP_HD __device__ __forceinline__
#define QBP_RCP(x) __builtin_amdgcn_rcp(x)
#define QBP_LDEXP(x, e) __builtin_amdgcn_ldexp(x, e)
#define QBP_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define QBP_RINT(x) __builtin_rint(x)
#define QBP_DEVICE_BITS 1
#elif defined(__HIPCC__)
#define QBP_HD __host__ __device__ inline
#define QBP_RCP(x) (1.0 / (x))
#define QBP_LDEXP(x, e) std::ldexp(x, e)
#define QBP_FREXP_EXP(x) (std::ilogb(x) + 1)
#define QBP_RINT(x) std::nearbyint(x)
#else
#define QBP_HD inline
#ifdef QBP_TEST_SEED_ERR   /* host tests: emulate the 2^-24.4 accuracy of v_rcp_f64 */
#define QBP_RCP(x) ((1.0 / (x)) * (1.0 + (QBP_TEST_SEED_ERR)))
#else
#define QBP_RCP(x) (1.0 / (x))
#endif
#define QBP_LDEXP(x, e) std::ldexp(x, e)
#define QBP_FREXP_EXP(x) (std::ilogb(x) + 1)
#define QBP_RINT(x) std::nearbyint(x)
#endif

namespace qbp {

// 2^k as a double, k in [-1022, 1023]
QBP_HD double pow2i(int k)
{
#ifdef QBP_DEVICE_BITS
    return __hiloint2double((k + 1023) << 20, 0);      // one 32-bit VALU op instead of v_ldexp_f64
#else
    return std::ldexp(1.0, k);
#endif
}

// x * 2^e for normal x whose result stays normal (exponent-field add on the high dword).
// x == 0 gives a value of magnitude 2^(e-1023): only used where that is negligible.
QBP_HD double scale2(double x, int e)
{
#ifdef QBP_DEVICE_BITS
    return __hiloint2double(__double2hiint(x) + (e << 20), __double2loint(x));
#else
    return x == 0.0 ? 0.0 : std::ldexp(x, e);
#endif
}

// biased exponent field minus 1022 == frexp exponent, for positive normal x
QBP_HD int frexp_exp_pos(double x)
{
#ifdef QBP_DEVICE_BITS
    return (__double2hiint(x) >> 20) - 1022;
#else
    return std::ilogb(x) + 1;
#endif
}

// a / b for normal b with a / b neither overflowing nor subnormal.  v_rcp_f64 is accurate to
// 2^-24.4 (measured on gfx950, tools/measure_misc.py); one Newton step brings the reciprocal to
// 2^-48; the quotient a*r then has that relative error and the residual correction
// q + (a - b q) r squares it away (2^-97 before the final rounding), i.e. the result is the
// correctly rounded quotient except when the exact quotient lies within 2^-97 of a rounding
// boundary.  (The IEEE sequence spends a second Newton step and operand scaling on top of this.)
QBP_HD double div_nr(double a, double b)
{
    double r = QBP_RCP(b);
    const double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    return __builtin_fma(rem, r, q);
}

// tanh(q * 0.5), any finite or infinite q.
//   g = |q|/2;  e^(-2g) = 2^k e^(2h), h = -g - k ln2/2, |h| <= ln2/4;  tau = tanh(h) by an odd
//   polynomial (7 terms);  e^(2h) = (1 + tau)/(1 - tau), hence with s = 2^k
//   tanh(g) = ((1 - s) - tau (1 + s)) / ((1 + s) - tau (1 - s)):
// numerator and denominator are single fmas around the exact constants 1 -+ s (k = 0 gives
// -2 tau / 2: no cancellation for small |q|; s -> 0 gives identical numerator and denominator:
// exactly 1 from |q| >= 38.2 on, like a correctly rounded tanh).  Max error 2.3 ulp, rms 0.7
// (an expm1-based form with 4 more operations per call reaches 1.85 / 0.6; both were measured
// against the reference on the 13 500-syndrome golden set with identical decoding results).
QBP_HD double tanh_half(double q)
{
    constexpr double TWO_INV_LN2 = 0x1.71547652b82fep+1;
    constexpr double LN2H_HI = 0x1.62e42f8000000p-2;      // ln2/2, 26 significant bits
    constexpr double LN2H_LO = 0x1.be8e7bcd5e4f2p-28;
    const double g = __builtin_fmin(__builtin_fabs(q) * 0.5, 20.0);   // tanh(20) rounds to 1; maps inf
    const double x = -g;
    const double kd = QBP_RINT(x * TWO_INV_LN2);           // k in [-58, 0]
    double h = __builtin_fma(-kd, LN2H_HI, x);             // exact
    h = __builtin_fma(-kd, LN2H_LO, h);
    const double z = h * h;
    double T = 0x1.c283105b585d0p-9;                       // tanh(h) = h + h z T(z)
    T = __builtin_fma(T, z, -0x1.2236d81ebea50p-7);
    T = __builtin_fma(T, z, 0x1.664eb23113e6dp-6);
    T = __builtin_fma(T, z, -0x1.ba1ba0e834e59p-5);
    T = __builtin_fma(T, z, 0x1.11111110cb360p-3);
    T = __builtin_fma(T, z, -0x1.555555555543ep-2);
    const double tau = __builtin_fma(h * z, T, h);
    const double s = pow2i((int)kd);
    const double sp = 1.0 + s, sm = 1.0 - s;               // exact for k >= -52; 1.0 below
    const double num = __builtin_fma(-tau, sp, sm);
    const double den = __builtin_fma(-tau, sm, sp);
    const double t = div_nr(num, den);
    return __builtin_copysign(t, q);
}

// 2 * atanh(y) for |y| < 1 (the caller has clipped to |y| <= 0.9999999).
//   w = (1 + a) / (1 - a), a = |y|;  pick e with w 2^-e in [1/sqrt2, sqrt2);
//   log w = e ln2 + 2 atanh(s),  s = ((1 + a) - 2^e (1 - a)) / ((1 + a) + 2^e (1 - a)).
// Numerator and denominator are linear in a with EXACT coefficients 1 -+ 2^e, so each is one
// fma of the true value (no rounding of 1 + a or 1 - a enters, and for e = 0 they are exactly 2a
// and 2: s == a, small messages keep full relative accuracy).  e depends on D = 1 - a only:
// w = 2 / D - 1, and the two range tests below are N >= sqrt2 Dm and N sqrt2 < Dm written with
// N = 2 - D, where Dm = D 2^e0 in [1, 2).
QBP_HD double atanh2(double y)
{
    constexpr double SQRT2 = 0x1.6a09e667f3bcdp+0;
    constexpr double TWO_SQRT2 = 0x1.6a09e667f3bcdp+1;
    constexpr double LN2_HI = 0x1.62e42f8000000p-1;
    constexpr double LN2_LO = 0x1.be8e7bcd5e4f2p-27;
    const double a = __builtin_fabs(y);
    const double D = 1.0 - a;                      // in [1e-7, 1]: normal
    const int e0 = 1 - frexp_exp_pos(D);
    const double Dm = scale2(D, e0);               // D 2^e0 in [1, 2)
    int e = e0;
    e += (__builtin_fma(Dm, SQRT2, D) <= 2.0) ? 1 : 0;
    e -= (__builtin_fma(D, SQRT2, Dm) > TWO_SQRT2) ? 1 : 0;
    const double pw = pow2i(e);                    // e in [0, 25]
    const double cp = 1.0 + pw, cm = 1.0 - pw;     // exact
    const double num = __builtin_fma(a, cp, cm);   // (1 + a) - 2^e (1 - a)
    const double den = __builtin_fma(a, cm, cp);   // (1 + a) + 2^e (1 - a)
    const double s = div_nr(num, den);             // |s| <= 0.1716
    const double z = s * s;
    double L = 0x1.2c7878482df85p-3;
    L = __builtin_fma(L, z, 0x1.39e6974a2f9f4p-3);
    L = __builtin_fma(L, z, 0x1.74636f948ce07p-3);
    L = __builtin_fma(L, z, 0x1.c71c6047521e3p-3);
    L = __builtin_fma(L, z, 0x1.24924930099b8p-2);
    L = __builtin_fma(L, z, 0x1.9999999993f92p-2);
    L = __builtin_fma(L, z, 0x1.5555555555559p-1);
    const double ed = (double)e;
    double tail = __builtin_fma(s * z, L, ed * LN2_LO);   // s z L(z) + e ln2_lo
    tail = __builtin_fma(2.0, s, tail);                   // + 2 s
    const double res = __builtin_fma(ed, LN2_HI, tail);   // + e ln2_hi (exact product)
    return __builtin_copysign(res, y);
}

// Variant-aware forms used by the kernels.  Plain sum-product (VARIANT 0) never produces a NaN
// message from NaN-free priors (|R| is clipped, inf - finite = inf), so it keeps the two-instruction
// min/max clip, which drops NaNs.  The damped variant (VARIANT 1, rework/decoding.py:131-191) can:
// damping = 1 with an infinite prior gives Q = 1 * inf + 0 * inf = NaN (:179), and numpy then carries
// the NaN through tanh, the row product, np.clip and arctanh into every message of that row.
template <int VARIANT>
QBP_HD double tanh_half_msg(double q)
{
    const double t = tanh_half(q);
    if (VARIANT == 1) return q != q ? q : t;
    return t;
}

template <int VARIANT>
QBP_HD double clip_unit(double x)
{
    constexpr double C = 0.9999999;                      // beliefPropagation.py:110
    if (VARIANT == 1) {
        const double y = x < -C ? -C : x;                // np.clip: NaN stays NaN
        return y > C ? C : y;
    }
    return __builtin_fmin(__builtin_fmax(x, -C), C);
}

}  // namespace qbp
