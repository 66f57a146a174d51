// TEST INFRASTRUCTURE: host build of the product's special-function header so that the CPU
// test-suite can compare it, value by value, with torch's CPU implementations, and so that the
// GPU self-test's checksums (same functions, same pseudo-random arguments, evaluated on the
// device) have a host value to be compared with.  Never loaded by the product path.
#include <math.h>
#include "../transductive-clip_amd/csrc/tclip_math.h"
#include "../transductive-clip_amd/csrc/tclip_selftest_inputs.h"
#include "../transductive-clip_amd/csrc/tclip_rsqrt14_table_dev.h"
extern "C" {
#define MC_MAP(name, expr) \
    void name(const float* x, float* y, long n) { for (long i = 0; i < n; i++) { const float v = x[i]; y[i] = (expr); } }
MC_MAP(mc_digamma, tclip::digamma_f32(v))
MC_MAP(mc_digamma_pos, tclip::digamma_pos_f32(v, tclip::kLogTab))
MC_MAP(mc_lgamma, tclip::lgamma_f32(v))
MC_MAP(mc_log, tclip::log_f32(v))
MC_MAP(mc_logf_glibc, tclip::logf_glibc(v))
MC_MAP(mc_libm_logf, logf(v))
MC_MAP(mc_exp, tclip::exp_f32_sleef(v))
MC_MAP(mc_sqrt_torch, tclip::sqrt_torch_f32(v))
MC_MAP(mc_lgamma_cr, (float)lgamma((double)v))
void mc_xp1_psi(const float* x, float* y, long n) { for (long i = 0; i < n; i++) { float p, l; tclip::digamma_lgamma_xp1(x[i], tclip::kLogTab, p, l); y[i] = p; } }
void mc_xp1_lg(const float* x, float* y, long n) { for (long i = 0; i < n; i++) { float p, l; tclip::digamma_lgamma_xp1(x[i], tclip::kLogTab, p, l); y[i] = l; } }
// every float of [1, 2.3): out[0] = arguments where the fp64 form of lgamma is sure and differs from the
// double-float form (must be 0), out[1] = arguments it is unsure about, out[2] = arguments visited
void mc_lgamma_f64_form(unsigned long long* out) {
    out[0] = out[1] = out[2] = 0;
    for (uint32_t b = tclip::f32_bits(1.0f); b < tclip::f32_bits(2.3f); b++) {
        const float x = tclip::bits_f32(b);
        bool sure;
        const float fast = tclip::lgamma_sleef_1_23_f64(x, sure);
        const float ref = tclip::lgamma_sleef_05_23(x);
        out[2]++;
        if (!sure) out[1]++;
        else if (tclip::f32_bits(fast) != tclip::f32_bits(ref)) out[0]++;
    }
}
// floats with bit patterns lo_bits..hi_bits-1, all in [2.3, 2^41]: the fp64 form of the large-argument lgamma against
// the double-float form.  out[0] = sure and different (must be 0), out[1] = unsure, out[2] = visited,
// out[3] = the largest distance from the rounding midpoint (in 2^-29 ulp) at which the two forms differ,
// whether called sure or not (the window that would have been needed).
void mc_lgamma_ge23_f64_form(unsigned int lo_bits, unsigned int hi_bits, unsigned int stride, unsigned long long* out) {
    unsigned long long bad = 0, unsure = 0, seen = 0, need = 0;
#pragma omp parallel for reduction(+ : bad, unsure, seen) reduction(max : need) schedule(static)
    for (long long bb = lo_bits; bb < (long long)hi_bits; bb += stride) {
        const float x = tclip::bits_f32((uint32_t)bb);
        bool sure;
        const float fast = tclip::lgamma_sleef_ge23_f64<false>(x, sure);
        const float ref = tclip::lgamma_sleef_ge23<false>(x);
        seen++;
        if (!sure) unsure++;
        if (tclip::f32_bits(fast) != tclip::f32_bits(ref)) {
            if (sure) bad++;
            double pd, v;
            (void)tclip::lgamma_sleef_ge23_f64_core<false>(x, pd, v);
            unsigned long long d = tclip::f64_distance_from_f32_midpoint(v);
            if (x <= 7.0f) { const unsigned long long dp = tclip::f64_distance_from_f32_midpoint(pd); d = dp < d ? dp : d; }
            need = d > need ? d : need;
        }
    }
    out[0] = bad; out[1] = unsure; out[2] = seen; out[3] = need;
}
// every float of [1, 10) (and a stretch beyond): the closed form of where digamma's recurrence leaves x against the loop,
// digamma_xp1 against its two-piece form, and the a > 7 specialisation of the fp64 lgamma against the general one.
// out[0] = mismatches of x, out[1] = mismatches of digamma, out[2] = mismatches of the lgamma specialisation, out[3] = visited
void mc_rec_closed_form(unsigned long long* out) {
    unsigned long long bad_x = 0, bad_psi = 0, bad_lg = 0, seen = 0;
#pragma omp parallel for reduction(+ : bad_x, bad_psi, bad_lg, seen) schedule(static)
    for (long long bb = tclip::f32_bits(1.0f); bb < (long long)tclip::f32_bits(24.0f); bb++) {
        const float x1 = tclip::bits_f32((uint32_t)bb);
        seen++;
        const float xc = tclip::digamma_rec_x(x1), xl = tclip::digamma_rec_x_loop(x1);
        if (tclip::f32_bits(xc) != tclip::f32_bits(xl)) bad_x++;
        if ((bb & 7) == 0) {            // the functions themselves on every 8th float
            if (x1 >= 2.3f && tclip::f32_bits(tclip::digamma_rec_acc_ge23(x1)) != tclip::f32_bits(tclip::digamma_rec_acc(x1))) bad_psi++;
            const float whole = tclip::digamma_pos_f32(x1, tclip::kLogTab);
            const float pieces = tclip::digamma_after_rec(xc, tclip::digamma_rec_acc(x1), tclip::kLogTab);
            if (tclip::f32_bits(whole) != tclip::f32_bits(pieces)) bad_psi++;
            // the dense passes' lean tails (no `x < 1e17` / `x == 1` branches; no partial sum from 10 on)
            if (tclip::f32_bits(whole) != tclip::f32_bits(tclip::digamma_after_rec_ge10<false>(xc, tclip::digamma_rec_acc(x1), tclip::kLogTab))) bad_psi++;
            if (x1 >= 10.0f && tclip::f32_bits(whole) != tclip::f32_bits(tclip::digamma_after_rec_ge10<true>(x1, 0.0f, tclip::kLogTab))) bad_psi++;
            if (x1 > 7.0f) {                          // the no-shift form where it is sure (mc_lgamma_gt7_f64_form sweeps the whole domain)
                bool s1;
                const float g1 = tclip::lgamma_sleef_gt7_f64<false>(x1, s1);
                if (s1 && tclip::f32_bits(g1) != tclip::f32_bits(tclip::lgamma_sleef_ge23<false>(x1))) bad_lg++;
            }
        }
    }
    out[0] = bad_x; out[1] = bad_psi; out[2] = bad_lg; out[3] = seen;
}
// floats with bit patterns lo_bits..hi_bits-1, all in (7, 2^41]: the no-shift fp64 form (lgamma_sleef_gt7_f64, the class-C
// pass of k_mm_split) against Sleef's double-float form.  out as mc_lgamma_ge23_f64_form.
void mc_lgamma_gt7_f64_form(unsigned int lo_bits, unsigned int hi_bits, unsigned int stride, unsigned long long* out) {
    unsigned long long bad = 0, unsure = 0, seen = 0, need = 0;
#pragma omp parallel for reduction(+ : bad, unsure, seen) reduction(max : need) schedule(static)
    for (long long bb = lo_bits; bb < (long long)hi_bits; bb += stride) {
        const float x = tclip::bits_f32((uint32_t)bb);
        bool sure;
        const float fast = tclip::lgamma_sleef_gt7_f64<false>(x, sure);
        const float ref = tclip::lgamma_sleef_ge23<false>(x);
        seen++;
        if (!sure) unsure++;
        if (tclip::f32_bits(fast) != tclip::f32_bits(ref)) {
            if (sure) bad++;
            double pd, v;
            (void)tclip::lgamma_sleef_ge23_f64_core<false, true>(x, pd, v);
            const unsigned long long d = tclip::f64_distance_from_f32_midpoint(v);
            need = d > need ? d : need;
        }
    }
    out[0] = bad; out[1] = unsure; out[2] = seen; out[3] = need;
}
// torch.sqrt as the MM kernels evaluate it (tclip_device.h: rsqrt14_entry / rsqrt14_from_entry / sqrt_torch_inrange_dev, restated
// here on the same derived table, whose estimate is the table value also at the exact powers of 4) against sqrt_torch_inrange_f32 on
// VRSQRT14PS's own values: every float of [1, 4) - every entry, both parities - at exponents 0, -60, +60, -100 and +98, and
// every power of two of the range.  out: {differences, arguments visited, powers of 4 among them}
static const uint32_t kMcRsqrt14DevTab[65536] = {TCLIP_RSQRT14_DEV_TABLE_VALUES};
static float mc_sqrt_dev_form(float x) {
    const uint32_t b = tclip::f32_bits(x);
    const uint32_t t = kMcRsqrt14DevTab[(b >> 8) & 0xffffu];
    const float y = tclip::bits_f32(t - (((b + 0x00800000u) >> 1) & 0x7f800000u));
    const float s = x * y;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * y, s);
}
void mc_sqrt_without_pow4(unsigned long long* out) {
    unsigned long long bad = 0, seen = 0, pow4 = 0;
    const float scales[5] = {1.0f, 0x1p-60f, 0x1p60f, 0x1p-100f, 0x1p98f};
    for (uint32_t m = 0; m < (1u << 24); m++) {
        const float x1 = tclip::bits_f32(0x3f800000u + m);
        for (int k = 0; k < 5; k++) {
            const float x = x1 * scales[k];
            bad += tclip::f32_bits(mc_sqrt_dev_form(x)) != tclip::f32_bits(tclip::sqrt_torch_inrange_f32(x));
            seen++;
        }
        pow4 += m == 0;
    }
    for (int e = -100; e <= 100; e++) {
        const float x = tclip::bits_f32((uint32_t)(127 + e) << 23);
        bad += tclip::f32_bits(mc_sqrt_dev_form(x)) != tclip::f32_bits(tclip::sqrt_torch_inrange_f32(x));
        seen++;
        pow4 += (e & 1) == 0;
    }
    out[0] = bad; out[1] = seen; out[2] = pow4;
}
// checksums of the routines over the self-test's argument streams (see tclip_selftest_inputs.h)
void mc_checksums(unsigned long long* out) {
    for (int f = 0; f < tclip::kSelfTestFunctions; f++) out[f] = 0;
    for (uint32_t i = 0; i < tclip::kSelfTestCount; i++)
        for (int f = 0; f < tclip::kSelfTestFunctions; f++) out[f] += tclip::selftest_term(f, i, tclip::kLogTab);
}
}

// torch's x.norm(p=2, dim=-1) order on the AVX-512 fixture host (see row_norm_torch in tclip_kernels.hip): eight fused
// accumulators by element index mod 8, added in order, then the tail (first four as product + add, rest fused)
extern "C" float mc_norm8(const float* x, long n) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long nv = n & ~7L;
    for (long d = 0; d < nv; d++) acc[d & 7] = __builtin_fmaf(x[d], x[d], acc[d & 7]);
    float b = acc[0];
    for (int l = 1; l < 8; l++) b += acc[l];
    long d = nv;
    if (n - d >= 4) for (int k = 0; k < 4; k++, d++) b = b + x[d] * x[d];
    for (; d < n; d++) b = __builtin_fmaf(x[d], x[d], b);
    return sqrtf(b);
}
