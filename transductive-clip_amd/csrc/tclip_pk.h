// Two-wide forms of the MM update's arithmetic for gfx950's packed fp32 pipe.
//
// One lane owns elements d, d+32, d+64, ... of a row; two of them are advanced together so that
// every add / multiply / fma of the restated library routines issues as ONE v_pk_{add,mul,fma}_f32
// (IEEE-rounded per component, exactly the scalar instruction's result) instead of two scalar
// instructions.  What cannot be packed stays per component: v_rcp_f32, the table look-ups of
// logf and of the VRSQRT14 emulation, the fp64 tail of logf, selects.
//
// Every function here is the component-wise image of the scalar function of the same name in
// tclip_math.h / tclip_device.h (same operations, same order, contraction off), valid on the
// same fast domain 0 <= a <= 2^40; k_selftest compares the two bit for bit on 2^24 arguments.
// Device only.
#pragma once
#include "tclip_device.h"

namespace tclip {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_sel(i2 m, f2 a, f2 b) { return m ? a : b; }
__device__ __forceinline__ f2 pk_sel(i2 m, float a, float b) { return m ? pk(a) : pk(b); }

// RN(1/x), see rcp_rn_f32
__device__ __forceinline__ f2 pk_rcp_rn(f2 x) {
    f2 r{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    const f2 e = pk_fma(-x, r, pk(1.0f));
    return pk_fma(e, r, r);
}

// RN(a/b), see div_rn_inrange_f32
__device__ __forceinline__ f2 pk_div_rn(f2 a, f2 b) {
    const f2 r = pk_rcp_rn(b);
    const f2 q = a * r;
    const f2 rem = pk_fma(-b, q, a);
    return pk_fma(rem, r, q);
}

// RN(a / b) from nb = -b: the reciprocal takes -nb through its source modifier and both fused steps take nb as it is, so no
// instruction is spent on a negation (with b in a register the compiler materialised -b with two v_xor per pair)
__device__ __forceinline__ f2 pk_div_rn_negden(f2 a, f2 nb) {
    const f2 r0{__builtin_amdgcn_rcpf(-nb.x), __builtin_amdgcn_rcpf(-nb.y)};
    const f2 r = pk_fma(pk_fma(nb, r0, pk(1.0f)), r0, r0);
    const f2 q = a * r;
    return pk_fma(pk_fma(nb, q, a), r, q);
}

// logf_glibc_tab of two arguments that are not 1.0f (here: >= 10, the recurrence has run), without its x == 1 branch
// and with both table entries fetched before either is used
__device__ __forceinline__ f2 pk_logf_glibc_ne1(f2 x, const LogTabEntry* tab) {
    const uint32_t ix0 = f32_bits(x.x), ix1 = f32_bits(x.y);
    const uint32_t tmp0 = ix0 - 0x3f330000u, tmp1 = ix1 - 0x3f330000u;
    const LogTabEntry e0 = tab[(tmp0 >> 19) & 15], e1 = tab[(tmp1 >> 19) & 15];
    const double z0 = (double)bits_f32(ix0 - (tmp0 & 0xff800000u)), z1 = (double)bits_f32(ix1 - (tmp1 & 0xff800000u));
    const double r0 = __builtin_fma(z0, e0.invc, -1.0), r1 = __builtin_fma(z1, e1.invc, -1.0);
    const double y00 = e0.logc + (double)((int32_t)tmp0 >> 23) * kLn2, y01 = e1.logc + (double)((int32_t)tmp1 >> 23) * kLn2;
    const double q0 = r0 * r0, q1 = r1 * r1;
    double y0 = __builtin_fma(0x1.5575b0be00b6ap-2, r0, -0x1.ffffef20a4123p-2);
    double y1 = __builtin_fma(0x1.5575b0be00b6ap-2, r1, -0x1.ffffef20a4123p-2);
    y0 = __builtin_fma(-0x1.00ea348b88334p-2, q0, y0);
    y1 = __builtin_fma(-0x1.00ea348b88334p-2, q1, y1);
    y0 = __builtin_fma(y0, q0, y00 + r0);
    y1 = __builtin_fma(y1, q1, y01 + r1);
    return f2{(float)y0, (float)y1};
}

// digamma(a+1), see digamma_xp1.  The step mask is a float: acc - m*RN(1/x) and x + m round
// once, exactly as the scalar's conditional updates (m*r is exact for m in {0,1}).
__device__ __forceinline__ f2 pk_digamma_xp1(f2 a, const LogTabEntry* tab) {
    f2 x = a + pk(1.0f), acc = pk(0.0f);
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const f2 m{below10_f32(x.x), below10_f32(x.y)};
        const f2 r = pk_rcp_rn(x);
        acc = pk_fma(-m, r, acc);
        x = x + m;
    }
    const f2 z = pk_rcp_rn(x * x);
    f2 p = pk(8.33333333333333333333E-2f);
    p = pk_fma(p, z, pk(-2.10927960927960927961E-2f));
    p = pk_fma(p, z, pk(7.57575757575757575758E-3f));
    p = pk_fma(p, z, pk(-4.16666666666666666667E-3f));
    p = pk_fma(p, z, pk(3.96825396825396825397E-3f));
    p = pk_fma(p, z, pk(-8.33333333333333333333E-3f));
    p = pk_fma(p, z, pk(8.33333333333333333333E-2f));
    const f2 y = z * p;
    const f2 lg = pk_logf_glibc_ne1(x, tab);
    const f2 series = ((acc + lg) - pk(0.5f) * pk_rcp_rn(x)) - y;
    return pk_sel(x == pk(10.0f), acc + pk(2.25175258906672110764f), series);
}

// the two pieces of digamma(a+1) the class-split kernel evaluates in phase C (the recurrence's partial sum comes from
// the dense queue): where the recurrence leaves x (digamma_rec_x) and the rest of calc_digamma (digamma_after_rec)
__device__ __forceinline__ f2 pk_digamma_rec_acc(f2 x1) {     // digamma_rec_acc
    f2 x = x1, acc = pk(0.0f);
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const f2 m{below10_f32(x.x), below10_f32(x.y)};
        const f2 r = pk_rcp_rn(x);
        acc = pk_fma(-m, r, acc);
        x = x + m;
    }
    return acc;
}
// the same for arguments in [1, 2.3) (class A of the split kernel): the first eight steps are always taken (x1 + 7 < 10),
// only the ninth needs its mask
__device__ __forceinline__ f2 pk_digamma_rec_acc_lt23(f2 x1) {
    f2 x = x1, acc = pk(0.0f);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        acc = acc - pk_rcp_rn(x);
        x = x + pk(1.0f);
    }
    const f2 m{below10_f32(x.x), below10_f32(x.y)};
    return pk_fma(-m, pk_rcp_rn(x), acc);
}
__device__ __forceinline__ f2 pk_digamma_rec_x(f2 x1) {
    const f2 f = x1 - __builtin_elementwise_floor(x1);
    const f2 r8 = ((f + pk(2.0f)) + pk(2.0f)) + pk(4.0f);
    const f2 up{__builtin_amdgcn_fmed3f(__builtin_fmaf(r8.x, 0x1p20f, 1.0f - 9.0f * 0x1p20f), 0.0f, 1.0f),
                __builtin_amdgcn_fmed3f(__builtin_fmaf(r8.y, 0x1p20f, 1.0f - 9.0f * 0x1p20f), 0.0f, 1.0f)};
    const f2 small_x = (r8 + pk(2.0f)) - up;
    return pk_sel(x1 < pk(10.0f), small_x, x1);
}
__device__ __forceinline__ f2 pk_digamma_after_rec(f2 x, f2 acc, const LogTabEntry* tab) {
    const f2 z = pk_rcp_rn(x * x);
    f2 p = pk(8.33333333333333333333E-2f);
    p = pk_fma(p, z, pk(-2.10927960927960927961E-2f));
    p = pk_fma(p, z, pk(7.57575757575757575758E-3f));
    p = pk_fma(p, z, pk(-4.16666666666666666667E-3f));
    p = pk_fma(p, z, pk(3.96825396825396825397E-3f));
    p = pk_fma(p, z, pk(-8.33333333333333333333E-3f));
    p = pk_fma(p, z, pk(8.33333333333333333333E-2f));
    const f2 y = z * p;
    const f2 lg = pk_logf_glibc_ne1(x, tab);
    const f2 series = ((acc + lg) - pk(0.5f) * pk_rcp_rn(x)) - y;
    return pk_sel(x == pk(10.0f), acc + pk(2.25175258906672110764f), series);
}

// double-float helpers, see the df_* functions
struct P2 { f2 x, y; };
__device__ __forceinline__ P2 pk_df_add2_f2_f(P2 a, f2 b) {
    const f2 s = a.x + b, v = s - a.x;
    const f2 t = (a.x - (s - v)) + (b - v);
    return P2{s, t + a.y};
}
__device__ __forceinline__ P2 pk_df_mul_f_f(f2 a, f2 b) {
    const f2 s = a * b;
    return P2{s, pk_fma(a, b, -s)};
}
__device__ __forceinline__ P2 pk_df_mul_f2_f(P2 a, f2 b) {
    const f2 s = a.x * b;
    return P2{s, pk_fma(a.y, b, pk_fma(a.x, b, -s))};
}

// lgamma on [0.5, 2.3), see lgamma_sleef_05_23 / sleef_lgamma_poly
__device__ __forceinline__ f2 pk_lgamma_sleef_05_23(f2 x) {
    const i2 o0 = x <= pk(1.2f);
    const P2 d = pk_df_add2_f2_f(P2{x, pk(0.0f)}, pk_sel(o0, -1.0f, -2.0f));
    const f2 t = d.x + d.y;
    f2 u = pk_sel(o0, +0.9435157776e+0f, +0.1102489550e-3f);
    u = pk_fma(u, t, pk_sel(o0, +0.8670063615e+0f, +0.8160019934e-4f));
    u = pk_fma(u, t, pk_sel(o0, +0.4826702476e+0f, +0.1528468856e-3f));
    u = pk_fma(u, t, pk_sel(o0, -0.8855129778e-1f, -0.2355068718e-3f));
    u = pk_fma(u, t, pk_sel(o0, +0.1013825238e+0f, +0.4962242092e-3f));
    u = pk_fma(u, t, pk_sel(o0, -0.1493408978e+0f, -0.1193488017e-2f));
    u = pk_fma(u, t, pk_sel(o0, +0.1697509140e+0f, +0.2891599433e-2f));
    u = pk_fma(u, t, pk_sel(o0, -0.2072454542e+0f, -0.7385451812e-2f));
    u = pk_fma(u, t, pk_sel(o0, +0.2705872357e+0f, +0.2058077045e-1f));
    P2 z = pk_df_add2_f2_f(pk_df_mul_f_f(u, t), pk_sel(o0, -0.400686534596170958447352690395e+0f, -0.673523028297382446749257758235e-1f));
    z = pk_df_add2_f2_f(pk_df_mul_f2_f(z, t), pk_sel(o0, +0.822466960142643054450325495997e+0f, +0.322467033928981157743538726901e+0f));
    z = pk_df_add2_f2_f(pk_df_mul_f2_f(z, t), pk_sel(o0, -0.577215665946766039837398973297e+0f, +0.422784335087484338986941629852e+0f));
    z = pk_df_mul_f2_f(z, t);
    return z.x + z.y;
}

// lgamma on [1, 2.3) with the fp64 tail, see lgamma_sleef_1_23_f64; arguments it is not sure about
// (6e-5 of them) send the wave through the double-float form.
// The branch x <= 1.2 selects one of two coefficient sets.  v_cndmask_b32 issues at half the rate of
// an fp32 FMA, so the selection is arithmetic here: with m = 1.0f (x <= 1.2) or 0.0f,
// c = fma(m, c1 - c2, c2) is c2 for m = 0 and RN(RN(c1 - c2) + c2) for m = 1, which is c1 for every
// pair used (checked at compile time); the fp64 tail does the same in double, where c1 - c2 is exact.
constexpr bool sel_by_fma_is_exact(float c1, float c2) { return (c1 - c2) + c2 == c1; }
#define TCLIP_PK_COEF(m, c1, c2) \
    ([&]() { static_assert(sel_by_fma_is_exact(c1, c2), "coefficient pair not selectable by fma"); \
             return pk_fma(m, pk((c1) - (c2)), pk(c2)); }())

// 1.0f where x <= 1.2f, else 0.0f: one v_fma_f32 with the clamp modifier (floats above 1.2f are at
// least one ulp = 2^-23 above it)
__device__ __forceinline__ float le_1p2_f32(float x) {
    constexpr float kNext = 0x1.333336p+0f;             // the float after 1.2f = 0x1.333334p+0
    return __builtin_amdgcn_fmed3f(__builtin_fmaf(-0x1p24f, x, 0x1p24f * kNext), 0.0f, 1.0f);
}

__device__ __forceinline__ float lgamma_tail_f64_m(float u, float t, double md, bool& sure) {
    const double td = (double)t;
    constexpr double a1 = (double)-0.400686534596170958447352690395e+0f, a2 = (double)-0.673523028297382446749257758235e-1f;
    constexpr double b1 = (double)+0.822466960142643054450325495997e+0f, b2 = (double)+0.322467033928981157743538726901e+0f;
    constexpr double c1 = (double)-0.577215665946766039837398973297e+0f, c2 = (double)+0.422784335087484338986941629852e+0f;
    static_assert((a1 - a2) + a2 == a1 && (b1 - b2) + b2 == b1 && (c1 - c2) + c2 == c1, "fp64 coefficient selection not exact");
    double z = __builtin_fma((double)u, td, __builtin_fma(md, a1 - a2, a2));
    z = __builtin_fma(z, td, __builtin_fma(md, b1 - b2, b2));
    z = __builtin_fma(z, td, __builtin_fma(md, c1 - c2, c2));
    z = z * td + 0.0;
    const uint32_t below = (uint32_t)f64_bits(z) & 0x1fffffffu;
    sure = (below - (0x10000000u - 0x4000u)) > 0x8000u;
    return (float)z;
}

__device__ __forceinline__ f2 pk_lgamma_sleef_1_23(f2 x) {
    const f2 m{le_1p2_f32(x.x), le_1p2_f32(x.y)};
    const f2 t = x - (pk(2.0f) - m);
    f2 u = TCLIP_PK_COEF(m, +0.9435157776e+0f, +0.1102489550e-3f);
    u = pk_fma(u, t, TCLIP_PK_COEF(m, +0.8670063615e+0f, +0.8160019934e-4f));
    u = pk_fma(u, t, TCLIP_PK_COEF(m, +0.4826702476e+0f, +0.1528468856e-3f));
    u = pk_fma(u, t, TCLIP_PK_COEF(m, -0.8855129778e-1f, -0.2355068718e-3f));
    u = pk_fma(u, t, TCLIP_PK_COEF(m, +0.1013825238e+0f, +0.4962242092e-3f));
    u = pk_fma(u, t, TCLIP_PK_COEF(m, -0.1493408978e+0f, -0.1193488017e-2f));
    u = pk_fma(u, t, TCLIP_PK_COEF(m, +0.1697509140e+0f, +0.2891599433e-2f));
    u = pk_fma(u, t, TCLIP_PK_COEF(m, -0.2072454542e+0f, -0.7385451812e-2f));
    u = pk_fma(u, t, TCLIP_PK_COEF(m, +0.2705872357e+0f, +0.2058077045e-1f));
    bool s0, s1;
    f2 r{lgamma_tail_f64_m(u.x, t.x, (double)m.x, s0), lgamma_tail_f64_m(u.y, t.y, (double)m.y, s1)};
    if (__builtin_expect(__ballot(!(s0 && s1)) != 0ull, 0)) {
        const f2 slow = pk_lgamma_sleef_05_23(x);
        r = f2{s0 ? r.x : slow.x, s1 ? r.y : slow.y};
    }
    return r;
}

// torch.sqrt, see sqrt_torch_inrange_f32
// both table look-ups of a pair issued together (one memory latency per pair instead of two in a row)
__device__ __forceinline__ f2 pk_rsqrt14(f2 x) {
    const uint32_t b0 = f32_bits(x.x), b1 = f32_bits(x.y);
    const uint32_t t0 = rsqrt14_entry(b0), t1 = rsqrt14_entry(b1);
    return f2{rsqrt14_from_entry(b0, t0), rsqrt14_from_entry(b1, t1)};
}

__device__ __forceinline__ f2 pk_sqrt_torch_inrange(f2 x) {
    const f2 y = pk_rsqrt14(x);
    const f2 s = x * y;
    return pk_fma(pk_fma(-s, s, x), pk(0.5f) * y, s);
}

// (lgamma(1) - lg1) + m with lgamma(1) = +0, the reference's `self.log_gamma_1 - torch.lgamma(alpha + 1) + digam * alpha`
// (em_dirichlet.py:155), as ONE subtraction m - lg1: (0 - lg1) + m and m - lg1 are the same sum of the same two numbers,
// rounded once, and differ only in the sign of a zero result (lg1 = +0 with m = -0, i.e. alpha = 0, where the curvature is
// the constant and t is discarded; a zero t with alpha > 1e-11 becomes |2 t / alpha^2| = +0 either way).
__device__ __forceinline__ f2 pk_t_of(f2 lg1, f2 m) { return m - lg1; }

// the quotient nume / deno of the update with the reference's IEEE behaviour at deno == 0 (total cancellation in t:
// +-inf or nan); the zero test is wave-uniform and per pair: min(|deno.x|, |deno.y|) == 0.  (Until round 5 it tested the
// PRODUCT of the two, which a NaN partner hides: the slot beyond the row of a ragged register pair holds a = 0, whose
// curvature without the small-parameter select is (|t| + |t|) / 0 = NaN.  v_min_f32 returns the other operand for a NaN.)
#ifndef TCLIP_QUOTIENT_FIXUP
#define TCLIP_QUOTIENT_FIXUP 1
#endif
__device__ __forceinline__ f2 pk_update_quotient(f2 nume, f2 deno) {
    f2 q = pk_div_rn(nume, deno);
#if TCLIP_QUOTIENT_FIXUP
    // v_div_fixup_f32 is the hardware's own table of IEEE division's special cases (x / 0 = +-inf, 0 / 0 = NaN, ...) and hands the
    // quotient through otherwise: one instruction per component and NO branch - the wave-uniform test it replaces ended a basic
    // block per register pair, so the scheduler could not move one pair's table look-ups, LDS reads or divisions under
    // another pair's dependent chain (round 5)
    return f2{__builtin_amdgcn_div_fixupf(q.x, deno.x, nume.x), __builtin_amdgcn_div_fixupf(q.y, deno.y, nume.y)};
#else
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(__builtin_fminf(__builtin_fabsf(deno.x), __builtin_fabsf(deno.y)) == 0.0f) != 0ull, 0))
        q = pk_sel(deno == pk(0.0f), nume * pk(__builtin_inff()), q);
    return q;
#endif
}

// One MM update of two parameters (see mm_update_algebra) in two stages, so that a caller can put other work (the next pair's digamma) between the
// table look-ups of torch.sqrt's VRSQRT14 emulation and their first use: stage 1 ends by issuing the two loads.
struct PkUpdateStage {
    f2 b, curv, delta;
    uint32_t t0, t1;            // table entries of delta.x, delta.y (loaded, not yet used)
};
// stage 1 from t = lgamma-and-digamma term of the curvature (pk_t_of) and d = digamma(a+1) - digamma(row sum), each given per
// component.  |2 t / a^2| is formed as (|t| + |t|) / a^2 - the same number: the doubling is exact and the quotient's rounding is
// symmetric in sign - because the scalar add takes |t| through its source modifiers, where the packed pipe has none and the
// absolute value of the quotient cost a select per component.
// kTiny = false: the caller knows that every parameter of the wavefront's rows exceeds 1e-11 (the curvature of smaller ones is
// the constant polygamma(1, 1), em_dirichlet.py:155) and the two compares and selects per pair go too.
template <bool kTiny>
__device__ __forceinline__ PkUpdateStage pk_mm_update_stage1_core(f2 a, f2 y, float t0, float t1, f2 d) {
    const f2 t2{__builtin_fabsf(t0) + __builtin_fabsf(t0), __builtin_fabsf(t1) + __builtin_fabsf(t1)};
    const f2 bigv = pk_div_rn_negden(t2, -a * a);
    PkUpdateStage st;
    if (kTiny) st.curv = pk_sel(a > pk(1e-11f), bigv, pk(1.6449340668482264f));
    else st.curv = bigv;
    st.b = (d - st.curv * a) - y;
    st.delta = pk_fma(pk(4.0f), st.curv, st.b * st.b);           // 4 curv is exact: one rounding, as b b + 4 curv has
    st.t0 = rsqrt14_entry(f32_bits(st.delta.x));
    st.t1 = rsqrt14_entry(f32_bits(st.delta.y));
    return st;
}
// stage 1 from digamma(a+1) and lgamma(a+1) of the two parameters, each in a register of its own (the split kernel reads them
// from its LDS planes: packing them first cost three moves per pair, and a fourth for the row's digamma)
template <bool kTiny>
__device__ __forceinline__ PkUpdateStage pk_mm_update_stage1_given(f2 a, f2 y, float psi_s, float psi10, float psi11, float lg10, float lg11) {
    const float m0 = psi10 * a.x, m1 = psi11 * a.y;
    return pk_mm_update_stage1_core<kTiny>(a, y, m0 - lg10, m1 - lg11, f2{psi10 - psi_s, psi11 - psi_s});
}
__device__ __forceinline__ PkUpdateStage pk_mm_update_stage1(f2 a, f2 y, f2 psi_s, f2 lg_big, const LogTabEntry* tab) {
    const f2 x1 = a + pk(1.0f);
    const i2 big = x1 >= pk(2.3f);
    const f2 lg_small = pk_lgamma_sleef_1_23(pk_sel(big, pk(2.0f), x1));
    const f2 psi1 = pk_digamma_xp1(a, tab);
    const f2 lg1 = pk_sel(big, lg_big, lg_small);
    const f2 t = pk_t_of(lg1, psi1 * a);
    return pk_mm_update_stage1_core<true>(a, y, t.x, t.y, psi1 - psi_s);
}
// stage 1 for two parameters whose arguments a + 1 BOTH lie below 2.3 (the caller's wave-uniform knowledge: nothing of the
// wavefront's rows is queued for the large-argument lgamma) - what pk_mm_update_stage1 computes for such a pair, without what
// the general form spends on the other case: the first eight steps of digamma's recurrence are always taken (a + 1 + 7 < 10:
// no step masks; the class-A pass of the split kernel relies on the same fact), the polynomial lgamma needs no stand-in
// argument and nothing is picked up from the queue.
__device__ __forceinline__ PkUpdateStage pk_mm_update_stage1_small(f2 a, f2 y, f2 psi_s, const LogTabEntry* tab) {
    const f2 x1 = a + pk(1.0f);
    const f2 lg1 = pk_lgamma_sleef_1_23(x1);
    f2 xr = x1, acc = pk(0.0f);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        acc = acc - pk_rcp_rn(xr);
        xr = xr + pk(1.0f);
    }
    const f2 m{below10_f32(xr.x), below10_f32(xr.y)};
    acc = pk_fma(-m, pk_rcp_rn(xr), acc);
    xr = xr + m;
    const f2 psi1 = pk_digamma_after_rec(xr, acc, tab);
    const f2 t = pk_t_of(lg1, psi1 * a);
    return pk_mm_update_stage1_core<true>(a, y, t.x, t.y, psi1 - psi_s);
}
__device__ __forceinline__ f2 pk_mm_update_stage2(const PkUpdateStage& st) {
    const f2 yr{rsqrt14_from_entry(f32_bits(st.delta.x), st.t0), rsqrt14_from_entry(f32_bits(st.delta.y), st.t1)};
    const f2 s = st.delta * yr;
    const f2 root = pk_fma(pk_fma(-s, s, st.delta), pk(0.5f) * yr, s);
    const f2 nume = -st.b + root, deno = pk(2.0f) * st.curv;
    return pk_update_quotient(nume, deno);
}

// both stages at once (k_selftest): lg_big = lgamma(a+1) for the components with a+1 >= 2.3, evaluated by the caller
__device__ __forceinline__ f2 pk_mm_update(f2 a, f2 y, f2 psi_s, f2 lg_big, const LogTabEntry* tab) {
    return pk_mm_update_stage2(pk_mm_update_stage1(a, y, psi_s, lg_big, tab));
}

}  // namespace tclip
