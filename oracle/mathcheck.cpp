// TEST INFRASTRUCTURE: host build of the product's special-function header so that the CPU
// test-suite can compare it, value by value, with torch's CPU implementations.  Never loaded
// by the product path.
#include <math.h>
#include "../transductive-clip_amd/csrc/tclip_math.h"
extern "C" {
void mc_digamma(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = tclip::digamma_f32(x[i]); }
void mc_lgamma(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = tclip::lgamma_f32(x[i]); }
void mc_log(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = tclip::log_f32(x[i]); }
void mc_logf_glibc(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = tclip::logf_glibc(x[i]); }
void mc_libm_logf(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = logf(x[i]); }
void mc_exp(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = tclip::exp_f32_sleef(x[i]); }
void mc_xp1_psi(const float* x, float* y, long n) { for (long i = 0; i < n; i++) { float p, l; tclip::digamma_lgamma_xp1(x[i], tclip::kLogTab, p, l); y[i] = p; } }
void mc_xp1_lg(const float* x, float* y, long n) { for (long i = 0; i < n; i++) { float p, l; tclip::digamma_lgamma_xp1(x[i], tclip::kLogTab, p, l); y[i] = l; } }
void mc_digamma_pos(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = tclip::digamma_pos_f32(x[i], tclip::kLogTab); }
void mc_lgamma_cr(const float* x, float* y, long n) { for (long i = 0; i < n; i++) y[i] = (float)lgamma((double)x[i]); }
}
