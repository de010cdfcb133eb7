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

#include "qbp_np_tables.hpp"

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

// ---------------------------------------------------------------------------------------------
// numpy-exact forms (round 3).  The reference's np.tanh / np.arctanh are, on the AVX512 hosts its
// golden vectors come from, two straight-line FMA kernels over small tables (numpy 2.2.6's vendored SVML:
// qbp_np_tables.hpp, tools/extract_np_svml.py).  Restated here operation by operation, they return numpy's
// BITS, so the posterior LLRs of the device equal the reference's bit for bit instead of within a drift
// that any other 1-ulp implementation accumulates over the iterations.
//   tanh:    16 intervals by exponent and top mantissa bit of |x| (0.1875, 0.25, 0.375 ... 16, 24),
//            degree-16 polynomial in r = |x| - midpoint, Horner.  No division.
//   arctanh: 0.5 (log(1 + a) - log(1 - a)); each log by R = round4(rcp14(.)), r = R x - 1 (fma),
//            -log R from a 16-entry table, degree-8 polynomial in r; double-double style recombination.
//            rcp14 is the AVX512 instruction VRCP14PD: rounded to 4 mantissa bits it is a step function of
//            the operand's top 16 mantissa bits (16 thresholds, tabulated on the build host).
// Tables live in LDS (NP_LDS_BYTES per workgroup), one row per tanh interval (18 doubles: 144 B stride,
// conflict-free for any mix of intervals across a wavefront).
struct NpImage {
    uint64_t tanh_row[16][18];      // {midpoint, c16, c15 .. c0}: Horner order, nine 16-byte pairs
    uint64_t log_hl[16][2];         // {log(1 + j/16) high, low}
    uint32_t rcp_lut[64];           // by the top 6 mantissa bits: see np_make_image, np_rcp14_hi
    uint32_t rcp_lut_p[64];         // the same + 0x03ff0000: for operands in [1, 2) taken as hi >> 4 (np_rcp14_hi_12)
};
constexpr int NP_LDS_BYTES = (int)sizeof(NpImage);      // 3072
constexpr int NP_LDS_DOUBLES = NP_LDS_BYTES / 8;

constexpr NpImage np_make_image()
{
    NpImage im{};
    for (int i = 0; i < 16; ++i) {
        im.tanh_row[i][0] = NP_TANH_SHIFTER[i];
        for (int k = 0; k < 17; ++k) im.tanh_row[i][17 - k] = NP_TANH_COEF[k][i];
        im.log_hl[i][0] = NP_ATANH_LOG_HI[i];
        im.log_hl[i][1] = NP_ATANH_LOG_LO[i];
    }
    // R_hi = 0x3ff00000 - (k << 16), k = number of thresholds <= m16 (the operand's top 16 mantissa bits).
    // Bin b holds m16 in [b << 10, (b + 1) << 10): at most one threshold inside.  With
    //   ent = 0x3ff0ffff - (k_left << 16) - (0x10000 - thr)      (thr = 0x10000 when the bin has none)
    // (ent - m16) & 0xffff0000 is R_hi: m16 >= thr borrows exactly one unit of bit 16.  The stored entries
    // carry, on top, the constant their reader would add next (bits >= 20: no effect on the borrow).
    for (int b = 0; b < 64; ++b) {
        uint32_t k_left = 0, thr = 0x10000u;
        for (int j = 0; j < 16; ++j) {
            if (NP_RCP14_THR16[j] <= (uint32_t)(b << 10)) ++k_left;
            else if (NP_RCP14_THR16[j] < (uint32_t)((b + 1) << 10)) thr = NP_RCP14_THR16[j];
        }
        const uint32_t ent = 0x3ff0ffffu - (k_left << 16) - (0x10000u - thr);
        im.rcp_lut[b] = ent + 0x3ff00000u;          // (+ the exponent bias of the result: np_rcp14_hi)
        im.rcp_lut_p[b] = ent + 0x03ff0000u;        // (+ the operand's own exponent field >> 4: np_rcp14_hi_12)
    }
    return im;
}

#if defined(__HIPCC__)
static __device__ const NpImage g_np_image = np_make_image();
// copy the tables into LDS (every thread of the workgroup calls this; the caller's barrier publishes)
__device__ __forceinline__ void np_tables_to_lds(double* lds, int tid, int nthreads)
{
    const uint64_t* src = reinterpret_cast<const uint64_t*>(&g_np_image);
    uint64_t* dst = reinterpret_cast<uint64_t*>(lds);
    for (int i = tid; i < NP_LDS_DOUBLES; i += nthreads) dst[i] = src[i];
}
#endif

QBP_HD unsigned np_hi(double x)
{
#ifdef QBP_DEVICE_BITS
    return (unsigned)__double2hiint(x);
#else
    uint64_t u; __builtin_memcpy(&u, &x, 8); return (unsigned)(u >> 32);
#endif
}
QBP_HD double np_from_hi_lo(unsigned hi, unsigned lo)
{
#ifdef QBP_DEVICE_BITS
    return __hiloint2double((int)hi, (int)lo);
#else
    uint64_t u = ((uint64_t)hi << 32) | lo; double d; __builtin_memcpy(&d, &u, 8); return d;
#endif
}
QBP_HD unsigned np_lo(double x)
{
#ifdef QBP_DEVICE_BITS
    return (unsigned)__double2loint(x);
#else
    uint64_t u; __builtin_memcpy(&u, &x, 8); return (unsigned)u;
#endif
}

// A 16-byte pair of table entries: ONE ds_read_b128 (conflict-free for any mix of rows: the row stride of
// 36 dwords visits every fourth of the 64 banks once in 16 rows).  Two 8-byte reads (ds_read2_b64) go through
// 32 banks at half the rate, where rows i and i + 8 collide -- measured: every variant of the kernel then ran
// at the same LDS-bound 5.0e6 syndromes/s.  T must be 16-byte aligned.
struct alignas(16) NpPair { double a, b; };

// Where the tables are.  On the device: the LDS BYTE ADDRESS of the NpImage as an integer -- every kernel keeps it
// at a compile-time constant address (0: the start of the dynamic LDS of a kernel without static LDS, or the
// address of a static array), so that a table access is `row offset + immediate`; through a pointer to the
// dynamic-LDS symbol the compiler emits an add of the symbol's address -- zero -- per access.  Host build (tests):
// a plain pointer.
#if defined(QBP_DEVICE_BITS)
typedef unsigned NpT;
QBP_HD NpPair np_ld_pair(NpT t, unsigned byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) NpPair*>(t + byte_off);
}
QBP_HD unsigned np_ld_u32(NpT t, unsigned byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) unsigned*>(t + byte_off);
}
#elif defined(__HIPCC__)      /* host pass of a device translation unit: same types, never executed */
typedef unsigned NpT;
QBP_HD NpPair np_ld_pair(NpT, unsigned) { return NpPair{0.0, 0.0}; }
QBP_HD unsigned np_ld_u32(NpT, unsigned) { return 0u; }
#else
typedef const double* NpT;
QBP_HD NpPair np_ld_pair(NpT t, unsigned byte_off)
{
    return *reinterpret_cast<const NpPair*>(reinterpret_cast<const char*>(t) + byte_off);
}
QBP_HD unsigned np_ld_u32(NpT t, unsigned byte_off)
{
    return *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(t) + byte_off);
}
#endif
#if defined(__HIPCC__)
// LDS byte address of a pointer into shared memory
__device__ __forceinline__ unsigned lds_address(const void* p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
#endif
constexpr unsigned NP_OFF_LOG = 16 * 18 * 8, NP_OFF_LUT = NP_OFF_LOG + 256, NP_OFF_LUT_P = NP_OFF_LUT + 256;

// np.tanh(q * 0.5).  T: the NpImage in LDS (as doubles).  inf -> +-1; NaN -> +-1 here (the kernels' NaN-
// preserving variant wraps it, tanh_half_msg).
// Byte offset of the table row of |x| = q/2 from the high dword of x: the interval (0 .. 15) is the exponent
// and top mantissa bit of |x| clamped to [0x7f8, 0x807] (v_bfe, v_med3); times 144 and minus the bias in ONE
// 24-bit multiply-add.  Opaque on the device, so that the compiler keeps the offset in a register and
// addresses the row's nine pairs as offset + immediate (it would otherwise re-associate the bias into every
// access -- or, given an opaque interval, multiply with the quarter-rate v_mul_lo_u32).
QBP_HD unsigned np_tanh_row(unsigned x_hi)
{
    int e = (int)((x_hi >> 19) & 0xfffu);
    e = e < 0x7f8 ? 0x7f8 : (e > 0x7f8 + 15 ? 0x7f8 + 15 : e);
#ifdef QBP_DEVICE_BITS
    unsigned off = (unsigned)(__mul24(e, 144) - 0x7f8 * 144);
    asm volatile("" : "+v"(off));
    return off;
#else
    return (unsigned)(e - 0x7f8) * 144u;
#endif
}

QBP_HD double np_tanh_half(double q, NpT T)
{
    const double x = q * 0.5;                                   // beliefPropagation.py:114
    const unsigned row = np_tanh_row(np_hi(x));
    NpPair c = np_ld_pair(T, row);                              // {midpoint, c16}
    // (|x| >= 24 selects the constant row {0, 0 ... 0, 1}: the clamp only keeps 0 * inf out of it)
    const double r = __builtin_fmin(__builtin_fabs(x), 32.0) - c.a;
    double p = c.b;
#pragma unroll
    for (int s = 1; s <= 8; ++s) {
        c = np_ld_pair(T, row + 16 * s);
        p = __builtin_fma(p, r, c.a);
        p = __builtin_fma(p, r, c.b);
    }
    return __builtin_copysign(p, x);                            // (p >= 0: the routine ORs the sign bit in)
}

// hi dword of round4(rcp14(v)) for a positive normal v (its low dword is 0)
QBP_HD unsigned np_rcp14_hi(unsigned v_hi, NpT T)
{
    const unsigned m16 = (v_hi >> 4) & 0xffffu;
    const unsigned ent = np_ld_u32(T, NP_OFF_LUT + ((v_hi >> 12) & 0xfcu));
    // biased mantissa part (as for v in [1, 2): exponent field 0x3ff), then the operand's exponent: 2^-e
    return ((ent - m16) & 0xffff0000u) - (v_hi & 0x7ff00000u);
}

// the same for P in [1, 2) (P = 1 + a, a <= 0.9999999): the exponent term is a constant, folded into the
// second table
QBP_HD unsigned np_rcp14_hi_12(unsigned p_hi, NpT T)
{
    return (np_ld_u32(T, NP_OFF_LUT_P + ((p_hi >> 12) & 0xfcu)) - (p_hi >> 4)) & 0xffff0000u;
}

// 2.0 * np.arctanh(a) for 0 <= a <= 0.9999999 -- the magnitude; np.arctanh is odd in every bit (the routine
// works on |y| and multiplies by +-0.5 at the end), so the kernels clip |y|, call this and set the sign bit.
// NaN in -> NaN out.
template <bool SCALE = true>
QBP_HD double np_arctanh_x2_abs(double a, NpT T)
{
    const double P = a + 1.0, M = 1.0 - a;
    const unsigned rp_hi = np_rcp14_hi_12(np_hi(P), T), rm_hi = np_rcp14_hi(np_hi(M), T);
    const NpPair lp = np_ld_pair(T, NP_OFF_LOG + ((rp_hi >> 12) & 0xf0u));
    const NpPair lm = np_ld_pair(T, NP_OFF_LOG + ((rm_hi >> 12) & 0xf0u));
    const double Ph = P - 1.0, Mh = M - 1.0;
    const double Pl = a - Ph;                       // 1 + a = P + Pl
    const double Ml = a + Mh;                       // 1 - a = M - Ml
    const double Rp = np_from_hi_lo(rp_hi, 0u), Rm = np_from_hi_lo(rm_hi, 0u);
    double rp = __builtin_fma(Rp, P, -1.0);
    rp = __builtin_fma(Pl, Rp, rp);
    double rm = __builtin_fma(M, Rm, -1.0);
    rm = __builtin_fma(-Ml, Rm, rm);
    constexpr double C0 = __builtin_bit_cast(double, NP_ATANH_POLY[0]), C1 = __builtin_bit_cast(double, NP_ATANH_POLY[1]),
                     C2 = __builtin_bit_cast(double, NP_ATANH_POLY[2]), C3 = __builtin_bit_cast(double, NP_ATANH_POLY[3]),
                     C4 = __builtin_bit_cast(double, NP_ATANH_POLY[4]), C5 = __builtin_bit_cast(double, NP_ATANH_POLY[5]),
                     C6 = __builtin_bit_cast(double, NP_ATANH_POLY[6]), C7 = __builtin_bit_cast(double, NP_ATANH_POLY[7]),
                     C8 = __builtin_bit_cast(double, NP_ATANH_POLY[8]);
    constexpr double LN2_HI = __builtin_bit_cast(double, NP_ATANH_LN2_HI), LN2_LO = __builtin_bit_cast(double, NP_ATANH_LN2_LO);
    double pp = __builtin_fma(C0, rp, C1), pm = __builtin_fma(C0, rm, C1);
    pp = __builtin_fma(rp, pp, C2); pm = __builtin_fma(rm, pm, C2);
    pp = __builtin_fma(rp, pp, C3); pm = __builtin_fma(rm, pm, C3);
    pp = __builtin_fma(rp, pp, C4); pm = __builtin_fma(rm, pm, C4);
    pp = __builtin_fma(rp, pp, C5); pm = __builtin_fma(rm, pm, C5);
    pp = __builtin_fma(rp, pp, C6); pm = __builtin_fma(rm, pm, C6);
    pp = __builtin_fma(rp, pp, C7); pm = __builtin_fma(rm, pm, C7);
    pp = __builtin_fma(rp, pp, C8); pm = __builtin_fma(rm, pm, C8);
    const double dE = (double)((int)(rm_hi >> 20) - (int)(rp_hi >> 20));     // VGETEXPPD difference
    const double Kh = __builtin_fma(LN2_HI, dE, lm.a - lp.a);
    const double Kl = __builtin_fma(LN2_LO, dE, lm.b - lp.b);
    const double rp2 = rp * rp, rm2 = rm * rm;
    const double S1 = rp + Kh;
    const double t4 = Kh - S1;
    const double S2 = S1 - rm;
    const double e1 = rp + t4;
    const double t5 = S2 - S1;
    const double A = __builtin_fma(rp2, pp, Kl);
    const double B = __builtin_fma(-rm2, pm, e1);
    const double e2 = rm + t5;
    double s = A + B;
    s = s - e2;
    s = S2 + s;
    // np.arctanh's * 0.5, then the caller's 2.0 * (:126): the identity unless s * 0.5 is subnormal
    // (|s| < 2^-1021, i.e. |a| of that size); callers that can rule that out skip the two multiplications
    return SCALE ? 2.0 * (s * 0.5) : s;
}

// 2.0 * np.arctanh(y), |y| <= 0.9999999 (NaN in -> NaN out)
QBP_HD double np_arctanh_x2(double y, NpT T)
{
    const double t = np_arctanh_x2_abs(__builtin_fabs(y), T);
    return np_from_hi_lo(np_hi(t) ^ (np_hi(y) & 0x80000000u), np_lo(t));       // * (+-0.5) * 2.0
}

// QBP_MATH_FAST (build-time, A/B only): the round-1/2 functions above (2.3 / 1.2 ulp, not numpy's bits)
#ifndef QBP_MATH_FAST
#define QBP_MATH_FAST 0
#endif
// T: the NpImage in LDS
template <int VARIANT>
QBP_HD double tanh_half_msg(double q, NpT T)
{
#if QBP_MATH_FAST
    const double t = tanh_half(q);
#else
    const double t = np_tanh_half(q, T);
#endif
    if (VARIANT == 1) return q != q ? q : t;
    return t;
}

// 2 * arctanh(y) of a clipped message (beliefPropagation.py:126)
QBP_HD double atanh2_msg(double y, NpT T)
{
#if QBP_MATH_FAST
    return atanh2(y);
#else
    return np_arctanh_x2(y, T);
#endif
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

// Scheduling fence between the messages of a wide row: the compiler otherwise overlaps all of a row's table-driven
// evaluations, whose live values then exceed the register budget of the wide builds (general-H kernel on
// 2592 x 7776, messages partly in L2: 1.33e5 -> 1.47e5 syndromes/s, 208 -> 184 spilled bytes; no change elsewhere:
// profiles/r03_ab_math.txt, block 6).  -DQBP_NO_EDGE_FENCES for A/B.
#if defined(QBP_DEVICE_BITS) && !defined(QBP_NO_EDGE_FENCES)
#define QBP_EDGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define QBP_EDGE_FENCE() ((void)0)
#endif

// The tail of the check update for one edge (beliefPropagation.py:125-126):
//     R = 2.0 * np.arctanh(np.clip(x * syndrome_sign, -0.9999999, 0.9999999)),   x = prod / t_safe.
// Multiplying by +-1, clipping to a symmetric interval and np.arctanh (which works on |y| and multiplies by
// +-0.5 at the end) are all odd functions bit for bit, so the magnitude goes through the clip and the
// arctanh and the sign bit -- sign(x) XOR the syndrome bit -- is set at the end: one min and one bit-field
// insert instead of a sign flip, a max, a min and the routine's own sign handling.
// NORMAL: the caller guarantees |x| >= 2^-1000 or so (see np_arctanh_x2_abs)
template <int VARIANT, bool NORMAL = false>
QBP_HD double check_message(double x, unsigned sbit, NpT T)
{
#if QBP_MATH_FAST
    const double xs = np_from_hi_lo(np_hi(x) ^ (sbit << 31), np_lo(x));
    return atanh2(clip_unit<VARIANT>(xs));
#else
    constexpr double C = 0.9999999;                          // beliefPropagation.py:110
    const double ax = __builtin_fabs(x);
    double a;
    if (VARIANT == 1) a = ax > C ? C : ax;                   // np.clip: a NaN stays a NaN
    else a = __builtin_fmin(ax, C);
    const double t = np_arctanh_x2_abs<!NORMAL>(a, T);
    return np_from_hi_lo((np_hi(t) & 0x7fffffffu) | ((np_hi(x) ^ (sbit << 31)) & 0x80000000u), np_lo(t));
#endif
}

}  // namespace qbp
