// Special functions of the EM-Dirichlet hot path, written once for HIP device code (gfx950)
// and for a plain host build (oracle/ and the CPU unit tests compile this header with g++).
//
// The reference does all of its arithmetic with PyTorch CPU fp32 ops, so "parity" means
// reproducing what THOSE implementations return, not the mathematically exact value:
//   torch.polygamma(0, x)  -> ATen calc_digamma(float)  (torch/include/ATen/native/Math.h:434-483,
//                             Cephes-derived: recurrence to x>=10, PSI_10, 7-term asymptotic series,
//                             libm logf = glibc 2.35 logf, itself an fp64 table algorithm)
//   torch.lgamma(x)        -> Sleef lgammaf_u10; measured in the build container to equal the
//                             correctly rounded value for all but 4e-3 (x in 2.5..10) / <3e-4 (x>10)
//                             of inputs, and ~20 % of inputs in 1..2.5 (1 ulp apart there)
//   torch.log / exp        -> Sleef logf_u10 / expf_u10
// Call sites in the reference: src/methods/zero_shot/em_dirichlet.py:35-38,143,151,154-155,163.
//
// Everything here is written so that no floating-point contraction is needed or wanted:
// compile with -ffp-contract=off; fused operations are spelled __builtin_fma(f) where the
// mimicked implementation fuses them.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TCLIP_HD __host__ __device__ __forceinline__
#define TCLIP_CONST_TABLE __device__ __constant__
#else
#define TCLIP_HD static inline
#endif

namespace tclip {

TCLIP_HD uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
TCLIP_HD float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
TCLIP_HD uint64_t f64_bits(double f) { return __builtin_bit_cast(uint64_t, f); }
TCLIP_HD double bits_f64(uint64_t u) { return __builtin_bit_cast(double, u); }

// ---------------------------------------------------------------------------------------------
// log tables.  {1/c, log(c)} for 16 sub-intervals of [0.7, 1.4): the published table of glibc's
// logf (sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c, glibc 2.35; origin ARM optimized-routines),
// values as found in this image's libm.so.6.
struct LogTabEntry { double invc, logc; };
#define TCLIP_LOG_TABLE_INIT { \
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, \
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, \
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},   \
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, \
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},                             \
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},   \
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},   \
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}}

#if defined(__HIP_DEVICE_COMPILE__)
static TCLIP_CONST_TABLE LogTabEntry kLogTab[16] = TCLIP_LOG_TABLE_INIT;
#else
static const LogTabEntry kLogTab[16] = TCLIP_LOG_TABLE_INIT;
#endif

constexpr double kLn2 = 0x1.62e42fefa39efp-1;

// Range reduction shared by every log below: x = 2^k * z, z in [0.7,1.4), r = z/c - 1 with
// |r| < 0.035, log(x) = log1p(r) + log(c) + k ln2.  x must be a positive normal float.
struct LogReduced { double r, y0; };
TCLIP_HD LogReduced log_reduce_f32(float x) {
    uint32_t ix = f32_bits(x);
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> 19) & 15;
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double z = (double)bits_f32(iz);
    LogReduced o;
    o.r = __builtin_fma(z, kLogTab[i].invc, -1.0);
    o.y0 = kLogTab[i].logc + (double)k * kLn2;
    return o;
}

// glibc 2.35 logf, FMA build (the ifunc variant every FMA-capable x86-64 selects); this is what
// ATen's calc_digamma(float) gets from `logf(x)`.  Positive normal inputs only.
TCLIP_HD double logf_glibc_as_double(float x) {
    if (f32_bits(x) == 0x3f800000u) return 0.0;
    LogReduced q = log_reduce_f32(x);
    double r = q.r, r2 = r * r;
    double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
    y = __builtin_fma(y, r2, q.y0 + r);
    return y;
}
TCLIP_HD float logf_glibc(float x) { return (float)logf_glibc_as_double(x); }

// log1p(r) for |r| < 0.04 to ~3e-13 absolute: enough for a correctly-rounded-in-practice fp32
// log and for the fp64 Stirling evaluation of lgamma below.
TCLIP_HD double log1p_small(double r) {
    double p = -1.0 / 8.0;
    p = __builtin_fma(p, r, 1.0 / 7.0);
    p = __builtin_fma(p, r, -1.0 / 6.0);
    p = __builtin_fma(p, r, 1.0 / 5.0);
    p = __builtin_fma(p, r, -1.0 / 4.0);
    p = __builtin_fma(p, r, 1.0 / 3.0);
    p = __builtin_fma(p, r, -1.0 / 2.0);
    double r2 = r * r;
    return __builtin_fma(p, r2, r);
}

// Accurate log of a positive normal float, as a double (|err| < 1e-12).
TCLIP_HD double log_f32_as_double(float x) {
    LogReduced q = log_reduce_f32(x);
    return q.y0 + log1p_small(q.r);
}

// Accurate log of a positive normal double (same table, indexed by the high word).
TCLIP_HD double log_f64(double v) {
    uint64_t iv = f64_bits(v);
    uint32_t hi = (uint32_t)(iv >> 32);
    uint32_t tmp = hi - 0x3fe66000u;
    int i = (tmp >> 16) & 15;
    int k = (int32_t)tmp >> 20;
    uint64_t iz = iv - ((uint64_t)(tmp & 0xfff00000u) << 32);
    double z = bits_f64(iz);
    double r = __builtin_fma(z, kLogTab[i].invc, -1.0);
    double y0 = kLogTab[i].logc + (double)k * kLn2;
    return y0 + log1p_small(r);
}

// fp32 log standing in for Sleef logf_u10 (torch.log on CPU): the correctly rounded value,
// which Sleef returns for >99.9 % of inputs.  Handles 0 (-> -inf) and subnormals.
TCLIP_HD float log_f32(float x) {
    if (x == 0.0f) return -__builtin_inff();
    if (x < 0.0f || x != x) return __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    double s = 0.0;
    if (f32_bits(x) < 0x00800000u) { x *= 0x1p64f; s = -64.0 * kLn2; }
    return (float)(log_f32_as_double(x) + s);
}

// ---------------------------------------------------------------------------------------------
// digamma, bit-for-bit ATen calc_digamma(float) for x > 0 (Math.h:434-483).  The x<=0 branches
// of the original (poles, reflection) are unreachable on this path (arguments are alpha+1 >= 1
// and row sums of positive alpha) and are reduced to their IEEE special values.
TCLIP_HD float digamma_asymptotic_f32(float x, float acc) {
    // acc + logf(x) - 0.5/x - y,   y = z*polevl(z, A, 6),  z = 1/(x*x)
    float y = 0.0f;
    if (x < 1.0e17f) {
        float z = 1.0f / (x * x);
        float p = 8.33333333333333333333E-2f;
        p = __builtin_fmaf(p, z, -2.10927960927960927961E-2f);
        p = __builtin_fmaf(p, z, 7.57575757575757575758E-3f);
        p = __builtin_fmaf(p, z, -4.16666666666666666667E-3f);
        p = __builtin_fmaf(p, z, 3.96825396825396825397E-3f);
        p = __builtin_fmaf(p, z, -8.33333333333333333333E-3f);
        p = __builtin_fmaf(p, z, 8.33333333333333333333E-2f);
        y = z * p;
    }
    return acc + logf_glibc(x) - (0.5f / x) - y;
}

TCLIP_HD float digamma_f32(float x) {
    if (x == 0.0f) return __builtin_copysignf(__builtin_inff(), -x);
    if (!(x > 0.0f)) return __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    float acc = 0.0f;
    while (x < 10.0f) {
        acc -= 1.0f / x;
        x += 1.0f;
    }
    if (x == 10.0f) return acc + 2.25175258906672110764f;
    return digamma_asymptotic_f32(x, acc);
}

// ---------------------------------------------------------------------------------------------
// lgamma for x > 0, evaluated in fp64 (shift to >= 10 by the recurrence, Stirling series, one
// log of the shift product) and rounded once: the correctly rounded fp32 value except within
// ~1e-14 absolute of a rounding boundary.
TCLIP_HD double stirling_tail(double x) {
    // sum_{n>=1} B_2n / (2n(2n-1) x^(2n-1)), x >= 10: 5 terms, truncation < 2e-14
    double t = 1.0 / x, t2 = t * t;
    double s = 1.0 / 1188.0;
    s = __builtin_fma(s, t2, -1.0 / 1680.0);
    s = __builtin_fma(s, t2, 1.0 / 1260.0);
    s = __builtin_fma(s, t2, -1.0 / 360.0);
    s = __builtin_fma(s, t2, 1.0 / 12.0);
    return s * t;
}

TCLIP_HD double lgamma_pos_as_double(float xf) {
    double x = (double)xf, prod = 1.0;
    bool shifted = false;
    while (x < 10.0) {
        prod *= x;
        x += 1.0;
        shifted = true;
    }
    double lx = log_f64(x);
    double r = __builtin_fma(x - 0.5, lx, -x) + 0.91893853320467274178 + stirling_tail(x);
    if (shifted) r -= log_f64(prod);
    return r;
}

TCLIP_HD float lgamma_f32(float x) {
    if (x != x) return x;
    if (x == __builtin_inff()) return x;
    if (!(x > 0.0f)) return __builtin_inff();            // poles / negative: unreachable here
    if (x < 0x1p-100f) return (float)(-log_f64((double)x));   // lgamma(x) = -log(x) - gamma*x + ...
    return (float)lgamma_pos_as_double(x);
}

// ---------------------------------------------------------------------------------------------
// Correctly rounded fp32 reciprocal / quotient / square root from the 1-ulp hardware
// approximations plus FMA residual corrections (Markstein-style).  On gfx950 these replace the
// compiler's generic IEEE expansions (v_div_scale/v_div_fmas/v_div_fixup, ~12 instructions each)
// on the hot path; tests/test_gpu_primitives.py checks them on the device against the IEEE
// operators, exhaustively over a binade for rcp and sqrt.  Preconditions: operands positive and
// normal with exponents in [-60, 60] (true on the MM path: arguments are alpha+1, alpha^2,
// 2*curvature, b^2+4*curvature; the callers fall back to the IEEE operator outside that range).
// On the host the IEEE operators are used directly.
// v_rcp_f32 (1 ulp) + ONE residual correction: the device self-test finds it equal to the IEEE
// quotient 1/x for every float of a binade (at three exponents), so a second step buys nothing.
TCLIP_HD float rcp_rn_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
#else
    return 1.0f / x;
#endif
}

// Two correction steps (kept for the self-test's comparison).
TCLIP_HD float rcp_rn2_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
#else
    return 1.0f / x;
#endif
}

// Branch-free forms for operands known to be in range (see the MM-path domain note at
// digamma_lgamma_xp1): no range test, no IEEE fallback.
TCLIP_HD float div_rn_inrange_f32(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = rcp_rn_f32(b);
    const float q = a * r;
    const float rem = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(rem, r, q);
#else
    return a / b;
#endif
}

TCLIP_HD float sqrt_rn_inrange_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    s = __builtin_fmaf(s, e, s);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(d, h, s);
    const float d2 = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d2, h, s);
#else
    return __builtin_sqrtf(x);
#endif
}

TCLIP_HD bool fast_range_f32(float x) {   // positive normal, |exponent| <= 60
    const uint32_t b = f32_bits(x);
    return b >= 0x21800000u && b <= 0x5d800000u;
}

TCLIP_HD float div_rn_f32(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (fast_range_f32(b) && fast_range_f32(__builtin_fabsf(a))) {
        const float r = rcp_rn_f32(b);
        const float q = a * r;
        const float rem = __builtin_fmaf(-b, q, a);
        return __builtin_fmaf(rem, r, q);
    }
#endif
    return a / b;
}

TCLIP_HD float sqrt_rn_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (fast_range_f32(x)) {
        const float y = __builtin_amdgcn_rsqf(x);
        float s = x * y, h = 0.5f * y;
        const float e = __builtin_fmaf(-h, s, 0.5f);
        s = __builtin_fmaf(s, e, s);
        h = __builtin_fmaf(h, e, h);
        const float d = __builtin_fmaf(-s, s, x);
        s = __builtin_fmaf(d, h, s);
        const float d2 = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(d2, h, s);
    }
#endif
    return __builtin_sqrtf(x);
}

TCLIP_HD void log_reduce_tab(float x, const LogTabEntry* tab, double& r, double& y0) {
    const uint32_t ix = f32_bits(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) & 15;
    const int k = (int32_t)tmp >> 23;
    const double z = (double)bits_f32(ix - (tmp & 0xff800000u));
    const LogTabEntry t = tab[i];
    r = __builtin_fma(z, t.invc, -1.0);
    y0 = t.logc + (double)k * kLn2;
}

TCLIP_HD double log_f64_tab(double v, const LogTabEntry* tab) {
    const uint64_t iv = f64_bits(v);
    const uint32_t tmp = (uint32_t)(iv >> 32) - 0x3fe66000u;
    const int i = (tmp >> 16) & 15;
    const int k = (int32_t)tmp >> 20;
    const double z = bits_f64(iv - ((uint64_t)(tmp & 0xfff00000u) << 32));
    const LogTabEntry t = tab[i];
    const double r = __builtin_fma(z, t.invc, -1.0);
    return (t.logc + (double)k * kLn2) + log1p_small(r);
}

// digamma_f32 for positive finite arguments with the fast reciprocal (row sums of alpha).
TCLIP_HD float digamma_pos_f32(float x, const LogTabEntry* tab) {
    if (!fast_range_f32(x) || !fast_range_f32(x * x)) return digamma_f32(x);
    float acc = 0.0f;
    while (x < 10.0f) {
        acc -= rcp_rn_f32(x);
        x += 1.0f;
    }
    if (x == 10.0f) return acc + 2.25175258906672110764f;
    const float rx = rcp_rn_f32(x), z = rcp_rn_f32(x * x);
    float p = 8.33333333333333333333E-2f;
    p = __builtin_fmaf(p, z, -2.10927960927960927961E-2f);
    p = __builtin_fmaf(p, z, 7.57575757575757575758E-3f);
    p = __builtin_fmaf(p, z, -4.16666666666666666667E-3f);
    p = __builtin_fmaf(p, z, 3.96825396825396825397E-3f);
    p = __builtin_fmaf(p, z, -8.33333333333333333333E-3f);
    p = __builtin_fmaf(p, z, 8.33333333333333333333E-2f);
    double r, y0;
    log_reduce_tab(x, tab, r, y0);
    const double r2 = r * r;
    double yl = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    yl = __builtin_fma(-0x1.00ea348b88334p-2, r2, yl);
    yl = __builtin_fma(yl, r2, y0 + r);
    return acc + (float)yl - (0.5f * rx) - z * p;
}

// ---------------------------------------------------------------------------------------------
// The two special functions of one MM update, fused:  psi1 = digamma(a+1) exactly as
// digamma_f32 computes it, lg1 = lgamma(a+1) exactly as lgamma_f32 computes it, for a >= 0.
// Straight-line code (9 predicated recurrence steps, no data-dependent branches) so that the
// compiler can interleave the independent elements a lane holds; one table reduction serves
// logf(x) of the digamma series, log(x) of Stirling's formula and, re-indexed, log of the
// recurrence product.  `tab` is the 16-entry log table (LDS copy on the device).
// Domain of the branch-free form: 0 <= a <= 2^40 (then x = a+1, x*x, the reciprocals and every
// intermediate stay normal); callers route anything else (NaN, inf, negative, huge) to the generic
// digamma_f32 / lgamma_f32.
TCLIP_HD bool mm_fast_domain(float a) { return a >= 0.0f && a <= 0x1p40f; }

TCLIP_HD void digamma_lgamma_xp1(float a, const LogTabEntry* tab, float& psi1, float& lg1) {
    float x = a + 1.0f;
    const double xd0 = (double)x;
    double prod = 1.0;
    float acc = 0.0f, nf = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 9; j++) {                            // while (x < 10) {acc -= 1/x; x += 1}
        const bool small = x < 10.0f;
        const float rxj = rcp_rn_f32(x);
        acc -= small ? rxj : 0.0f;
        prod *= small ? xd0 + (double)j : 1.0;               // exact shifts for the lgamma recurrence
        const float inc = small ? 1.0f : 0.0f;
        x += inc;
        nf += inc;
    }
    const double xd = xd0 + (double)nf;
    // digamma: asymptotic series at x >= 10 (the x == 10 case returns the tabulated psi(10))
    const float xx = x * x;
    const float rx = rcp_rn_f32(x);
    const float z = rcp_rn_f32(xx);                          // x <= 2^40 + 9 < 1e17: always the series
    float p = 8.33333333333333333333E-2f;
    p = __builtin_fmaf(p, z, -2.10927960927960927961E-2f);
    p = __builtin_fmaf(p, z, 7.57575757575757575758E-3f);
    p = __builtin_fmaf(p, z, -4.16666666666666666667E-3f);
    p = __builtin_fmaf(p, z, 3.96825396825396825397E-3f);
    p = __builtin_fmaf(p, z, -8.33333333333333333333E-3f);
    p = __builtin_fmaf(p, z, 8.33333333333333333333E-2f);
    const float yser = z * p;
    double r, y0;
    log_reduce_tab(x, tab, r, y0);
    const double r2 = r * r;
    double yl = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);   // glibc logf polynomial
    yl = __builtin_fma(-0x1.00ea348b88334p-2, r2, yl);
    yl = __builtin_fma(yl, r2, y0 + r);
    const float logx = (float)yl;                            // x >= 10 here, never 1.0
    const float asym = acc + logx - (0.5f * rx) - yser;     // 0.5f/x == 0.5f*RN(1/x): scaling by 2 is exact
    psi1 = (x == 10.0f) ? acc + 2.25175258906672110764f : asym;
    // lgamma(a+1) = Stirling(xd) - log(prod), all in fp64, rounded once
    const double xf = (double)x;
    double t = (double)rx;                                   // ~1/xd to 1e-6: two Newton steps
    t = t * __builtin_fma(-xd, t, 2.0);
    t = t * __builtin_fma(-xd, t, 2.0);
    const double dl = (xd - xf) * t;                         // log(xd) = log(x) + log1p((xd-x)/x)
    const double lxd = (y0 + log1p_small(r)) + __builtin_fma(-0.5 * dl, dl, dl);
    const double t2 = t * t;
    double st = 1.0 / 1188.0;
    st = __builtin_fma(st, t2, -1.0 / 1680.0);
    st = __builtin_fma(st, t2, 1.0 / 1260.0);
    st = __builtin_fma(st, t2, -1.0 / 360.0);
    st = __builtin_fma(st, t2, 1.0 / 12.0);
    double lg = __builtin_fma(xd - 0.5, lxd, -xd) + 0.91893853320467274178 + st * t;
    lg -= log_f64_tab(prod, tab);
    // a < 2^-10: Stirling minus log-product cancels 12.8 - 12.8 and keeps only ~1e-15 absolute;
    // the Taylor series of lgamma(1+a) = -gamma a + sum_k (-1)^k zeta(k) a^k / k is exact to fp64 there
    const double ad = (double)a;
    double ser = 0.20738555102867398527;                          //  zeta(5)/5
    ser = __builtin_fma(ser, ad, -0.27058080842778454788);        // -zeta(4)/4
    ser = __builtin_fma(ser, ad, 0.40068563438653142847);         //  zeta(3)/3
    ser = __builtin_fma(ser, ad, -0.82246703342411321824);        // -zeta(2)/2
    ser = __builtin_fma(ser, ad, -0.57721566490153286061);        // -gamma
    lg1 = (float)(a < 0x1p-10f ? ser * ad : lg);
}

// ---------------------------------------------------------------------------------------------
// expf as Sleef expf_u10 computes it (torch softmax on CPU goes through Vectorized::exp):
// Cody-Waite reduction by ln2 in two fp32 pieces, degree-5 polynomial, all fused.
TCLIP_HD float exp_f32_sleef(float d) {
    float qf = __builtin_rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
    int q = (int)qf;
    float s = __builtin_fmaf(qf, -0.693145751953125f, d);
    s = __builtin_fmaf(qf, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
    u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
    u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
    u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
    u = __builtin_fmaf(u, s, 0.5f);
    u = 1.0f + __builtin_fmaf(s * s, u, s);
    // ldexp2kf: scale by 2^(q>>1) twice
    int q1 = q >> 1, q2 = q - q1;
    u = u * bits_f32((uint32_t)(q1 + 127) << 23) * bits_f32((uint32_t)(q2 + 127) << 23);
    if (d < -104.0f) u = 0.0f;
    if (d > 104.0f) u = __builtin_inff();
    return u;
}

}  // namespace tclip
