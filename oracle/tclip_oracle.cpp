// TEST INFRASTRUCTURE - CPU oracle #2: plain C++ restatement of the EM-Dirichlet /
// Hard EM-Dirichlet loop, decomposed the way the HIP kernels are (one (task, class) row =
// one majorize-minimize problem; batch-global stop test every 50 iterations).
//
// Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg load the library built from this file.  It shares nothing with the product except the
// special-function header (transductive-clip_amd/csrc/tclip_math.h), which is itself checked
// against torch's CPU implementations by tests/test_math_host.py.
//
// Reference lines restated (paths relative to the reference repository):
//   src/methods/zero_shot/em_dirichlet.py:28-40   logits  -> estep_logits()
//   src/methods/zero_shot/em_dirichlet.py:132-143 u       -> estep_softmax()
//   src/methods/zero_shot/em_dirichlet.py:145-151 v       -> inside run()
//   src/methods/zero_shot/em_dirichlet.py:153-177 MM loop -> mm_step_row(), mm_solve()
//   src/methods/zero_shot/em_dirichlet.py:214-244 outer   -> run()
//   src/methods/zero_shot/hard_em_dirichlet.py:255-258    -> hard one-hot in run()
//   src/methods/few_shot/em_dirichlet.py:186-200          -> support statistics in run()
//
// Parity pinning: tests/test_oracle_c_golden.py compares this oracle with the golden vectors
// the reference produced (tests/golden/*.npz): identical MM iteration counts and argmax, alpha
// within the tolerance written in that test.  It is NOT bit-exact to the reference: torch's
// lgamma/log are replaced by correctly rounded values (see tclip_math.h), everything else
// (digamma, exp, operation order, reduction order of torch's AVX-512 CPU kernels) is followed
// to the bit.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../transductive-clip_amd/csrc/tclip_math.h"

namespace {

constexpr float kEps = 1e-15f;

inline int ceil_log2(long n) {
    int l = 0;
    while ((1L << l) < n) l++;
    return l;
}

// torch sum over a strided (outer) dimension: aten/src/ATen/native/cpu/SumKernel.cpp
// multi_row_sum - 4-level cascade, 2^max(4, ceil_log2(n)/4) elements per level-0 block.
template <typename F>
inline float sum_cascade(long n, F get) {
    const int level_power = ceil_log2(n) / 4 > 4 ? ceil_log2(n) / 4 : 4;
    const long step = 1L << level_power, mask = step - 1;
    float acc[4] = {0, 0, 0, 0};
    long i = 0;
    for (; i + step <= n;) {
        for (long j = 0; j < step; ++j, ++i) acc[0] += get(i);
        for (int j = 1; j < 4; ++j) {
            acc[j] += acc[j - 1];
            acc[j - 1] = 0;
            if ((i & (mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < n; ++i) acc[0] += get(i);
    for (int j = 1; j < 4; ++j) acc[0] += acc[j];
    return acc[0];
}

// ILP-4 row sum (SumKernel.cpp row_sum): element i goes to partial i%4, each partial is a
// cascade over i/4, the <4 leftovers join partial 0, then p0 += p1, p2, p3.
template <typename F>
inline float sum_ilp4(long n, F get) {
    const long size_ilp = n / 4;
    float p[4];
    for (int r = 0; r < 4; r++) p[r] = sum_cascade(size_ilp, [&](long m) { return get(4 * m + r); });
    for (long i = size_ilp * 4; i < n; i++) p[0] += get(i);
    for (int r = 1; r < 4; r++) p[0] += p[r];
    return p[0];
}

// torch sum over a strided dimension, for output column `col` of `ncols` contiguous columns
// (vectorized_outer_sum): the leading multiple of 4 vectors (32 floats; sum_stub is registered
// without the AVX-512 variant, so vectors are 8 floats wide even on AVX-512 hosts) goes
// through the plain cascade, the remaining columns through the ILP-4 row sum.  With fewer than 8
// columns ATen takes scalar_outer_sum instead: groups of 4 columns through the cascade, the
// 1-3 left over through the ILP-4 row sum.
inline bool outer_column_is_cascade(long col, long ncols) {
    return ncols >= 8 ? col < (ncols / 32) * 32 : col < (ncols / 4) * 4;
}
template <typename F>
inline float sum_outer(long n, long col, long ncols, F get) {
    return outer_column_is_cascade(col, ncols) ? sum_cascade(n, get) : sum_ilp4(n, get);
}

// torch sum over the contiguous last dimension (vectorized_inner_sum with 8-float vectors and
// 4 interleaved vector accumulators; scalar ILP-4 row sum when n < 8).
template <typename F>
inline float sum_inner(long n, F get) {
    constexpr int W = 8;
    if (n < W) return sum_ilp4(n, get);
    const long vec_size = n / W, size_ilp = vec_size / 4;
    float p[4][W];
    for (int r = 0; r < 4; r++)
        for (int j = 0; j < W; j++)
            p[r][j] = sum_cascade(size_ilp, [&](long m) { return get(W * (4 * m + r) + j); });
    for (long v = size_ilp * 4; v < vec_size; v++)
        for (int j = 0; j < W; j++) p[0][j] += get(W * v + j);
    for (int r = 1; r < 4; r++)
        for (int j = 0; j < W; j++) p[0][j] += p[r][j];
    float fin = 0;
    for (long k = vec_size * W; k < n; k++) fin += get(k);
    for (int j = 0; j < W; j++) fin += p[0][j];
    return fin;
}

// vec::reduce_all(+) as torch's softmax uses it (ATen/cpu/vec/functional_base.h): 16 lanes
// accumulate strided, partial last vector merged lane-wise, then an 8/4/2/1 butterfly.
inline float sum_reduce_all(const float* x, long n) {
    if (n < 16) {
        float a = x[0];
        for (long i = 1; i < n; i++) a += x[i];
        return a;
    }
    float acc[16];
    for (int j = 0; j < 16; j++) acc[j] = x[j];
    long d = 16;
    for (; d < n - (n % 16); d += 16)
        for (int j = 0; j < 16; j++) acc[j] += x[d + j];
    for (long j = 0; d + j < n; j++) acc[j] += x[d + j];
    for (int s = 8; s >= 1; s >>= 1) {
        float t[16];
        for (int j = 0; j < 16; j++) t[j] = acc[j] + acc[j ^ s];
        memcpy(acc, t, sizeof t);
    }
    return acc[0];
}

struct MMRowResult { double num, den; };

// One MM iteration on one (task, class) row: beta -> next, both length K.
inline void mm_step_row(const float* beta, const float* y, float* next, int K) {
    const float s = sum_inner(K, [&](long d) { return beta[d]; });
    const float psi_s = tclip::digamma_f32(s);
    for (int d = 0; d < K; d++) {
        const float a = beta[d];
        float psi1, lg1;
        tclip::digamma_lgamma_xp1(a, tclip::kLogTab, psi1, lg1);
        float curv;
        if (a > 1e-11f) {
            float t = (0.0f - lg1) + psi1 * a;
            curv = fabsf((2.0f * t) / (a * a));
        } else {
            curv = 1.6449340668482264f;   // polygamma(1, 1) = pi^2/6 in fp32
        }
        float b = (psi1 - psi_s) - curv * a;
        b = b - y[d];
        const float delta = b * b + 4.0f * curv;
        next[d] = (-b + tclip::sqrt_torch_f32(delta)) / (2.0f * curv);   // torch.sqrt = MKL vsSqrt (HA)
    }
}

}  // namespace

static double g_min_stop_margin = 1e300;
// Instrumentation (design studies only, scripts/live_row_cycles.py): when set, every live row's MM trajectory is watched for
// the first iteration at which its state equals one of the previous `g_cycle_max_period` states (64-bit hashes of the
// row); g_cycle_out[(it * N*K + row) * 2] = that iteration (or -1), [+1] = the period.
static int32_t* g_cycle_out = nullptr;
static int g_cycle_max_period = 0;
static inline uint64_t row_hash(const float* x, int K) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (int d = 0; d < K; d++) {
        uint32_t b;
        memcpy(&b, &x[d], 4);
        h = (h ^ b) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    return h;
}
// Second design-study hook (scripts/stationary_elements.py): g_same_out[(it * iter_mm + l) * 2] = number of elements of live
// rows that MM iteration l of outer iteration `it` left bitwise unchanged, [+1] = number of elements of live rows.
static int64_t* g_same_out = nullptr;
extern "C" {
void tclip_oracle_set_cycle_probe(int32_t* out, int max_period) { g_cycle_out = out; g_cycle_max_period = max_period; }
void tclip_oracle_set_stationary_probe(int64_t* out) { g_same_out = out; }

// Runs the whole loop for ONE reference batch of n_task tasks.
//   z        [N,Q,K] probability features (query)
//   xs, ys   [N,S,K], [N,S] support features / labels, or NULL for zero-shot
//   lambd    the reference's integer lambd (int(K/5)*Q zero-shot, int(K/k_eff)*Q few-shot)
//   outputs  u [N,Q,K], v [N,K], alpha [N,K,K], criterions [iters], mm_iters [iters],
//            argmax_trace [iters,N,Q] (may be NULL)
int tclip_oracle_run(const float* z, const float* xs, const int64_t* ys, int N, int Q, int K, int S,
                     int iters, int iter_mm, int lambd, int hard, float* u, float* v, float* alpha,
                     float* criterions, int32_t* mm_iters, int16_t* argmax_trace) {
    const bool few = xs != nullptr;
    const size_t NKK = (size_t)N * K * K, NQK = (size_t)N * Q * K;
    std::vector<float> logz(NQK), y(NKK), alpha_old(NKK), next(NKK), beta(NKK), cs((size_t)N * K);
    std::vector<float> sup_sum, sup_cnt;
    std::vector<uint8_t> live((size_t)N * K, 1);
    for (size_t i = 0; i < NQK; i++) logz[i] = tclip::log_f32(z[i] + kEps);
    for (size_t i = 0; i < NQK; i++) u[i] = z[i];
    for (size_t i = 0; i < NKK; i++) alpha[i] = alpha_old[i] = 1.0f;
    for (size_t i = 0; i < (size_t)N * K; i++) v[i] = 0.0f;
    if (few) {
        // sum_s 1[y_s = k] * log(x_s + eps), in torch's outer-sum order over all S rows
        // (zeros from the one-hot product included: they only matter for the cascade blocks)
        sup_sum.assign(NKK, 0.0f);
        sup_cnt.assign((size_t)N * K, 0.0f);
        std::vector<float> logs((size_t)S * K);
        for (int n = 0; n < N; n++) {
            for (size_t i = 0; i < (size_t)S * K; i++) logs[i] = tclip::log_f32(xs[(size_t)n * S * K + i] + kEps);
            for (int k = 0; k < K; k++) {
                sup_cnt[(size_t)n * K + k] =
                    sum_outer(S, k, K, [&](long s) { return ys[(size_t)n * S + s] == k ? 1.0f : 0.0f; });
                for (int d = 0; d < K; d++)
                    sup_sum[((size_t)n * K + k) * K + d] = sum_outer(S, (long)k * K + d, (long)K * K, [&](long s) {
                        return (ys[(size_t)n * S + s] == k ? 1.0f : 0.0f) * logs[(size_t)s * K + d];
                    });
            }
        }
    }

    for (int it = 0; it < iters; it++) {
        // ---- M-step statistics
        for (int n = 0; n < N; n++)
            for (int k = 0; k < K; k++) {
                const float* un = u + (size_t)n * Q * K;
                const float c = sum_outer(Q, k, K, [&](long q) { return un[q * K + k]; });
                cs[(size_t)n * K + k] = c;
                float* yr = &y[((size_t)n * K + k) * K];
                const float* lz = &logz[(size_t)n * Q * K];
                if (few) {
                    const float w = 1.0f / (sup_cnt[(size_t)n * K + k] + c);
                    for (int d = 0; d < K; d++) {
                        float t = sum_outer(Q, (long)k * K + d, (long)K * K, [&](long q) { return un[q * K + k] * lz[q * K + d]; });
                        yr[d] = w * (sup_sum[((size_t)n * K + k) * K + d] + t);
                    }
                } else {
                    const bool alive = c > kEps;
                    live[(size_t)n * K + k] = alive;
                    const float den = c < kEps ? kEps : c;
                    for (int d = 0; d < K; d++) {
                        float t = sum_outer(Q, (long)k * K + d, (long)K * K, [&](long q) { return un[q * K + k] * lz[q * K + d]; });
                        t = t / den;
                        // y*live + (1-live)*1*(-10), as the reference spells it
                        yr[d] = alive ? (t * 1.0f + (0.0f * -10.0f)) : (t * 0.0f + -10.0f);
                    }
                }
            }
        // ---- MM fixed point, batch-global stop test
        memcpy(beta.data(), alpha, NKK * sizeof(float));
        int executed = 0;
        bool result_in_next = true;
        std::vector<uint64_t> ring;
        const int P = g_cycle_out ? g_cycle_max_period : 0;
        if (P) {
            ring.assign((size_t)N * K * P, 0);
            for (long row = 0; row < (long)N * K; row++) { g_cycle_out[((size_t)it * N * K + row) * 2] = -1; g_cycle_out[((size_t)it * N * K + row) * 2 + 1] = 0; }
        }
        for (int l = 0; l < iter_mm; l++) {
#pragma omp parallel for schedule(static)
            for (long row = 0; row < (long)N * K; row++) {
                mm_step_row(&beta[row * K], &y[row * K], &next[row * K], K);
                if (P && (few || live[row]) && g_cycle_out[((size_t)it * N * K + row) * 2] < 0) {
                    if (l == 0) ring[(size_t)row * P] = row_hash(&beta[row * K], K);          // state 0
                    const uint64_t h = row_hash(&next[row * K], K);                          // state l + 1
                    for (int p = 1; p <= P && p <= l + 1; p++)
                        if (ring[(size_t)row * P + (l + 1 - p) % P] == h) {
                            g_cycle_out[((size_t)it * N * K + row) * 2] = l + 1;
                            g_cycle_out[((size_t)it * N * K + row) * 2 + 1] = p;
                            break;
                        }
                    ring[(size_t)row * P + (l + 1) % P] = h;
                }
            }
            if (g_same_out) {
                int64_t same = 0, total = 0;
                for (long row = 0; row < (long)N * K; row++) {
                    if (!(few || live[row])) continue;
                    total += K;
                    for (int d = 0; d < K; d++) same += memcmp(&next[row * K + d], &beta[row * K + d], 4) == 0;
                }
                g_same_out[((size_t)it * iter_mm + l) * 2] = same;
                g_same_out[((size_t)it * iter_mm + l) * 2 + 1] = total;
            }
            executed++;
            result_in_next = true;
            if (l > 0 && l % 50 == 0) {
                double num = 0, den = 0;
                for (size_t i = 0; i < NKK; i++) {
                    const double dlt = (double)next[i] - (double)beta[i];
                    num += dlt * dlt;
                    den += (double)beta[i] * (double)beta[i];
                }
                const float nn = (float)sqrt(num), dn = (float)sqrt(den);
                {   // how close this run's stop tests came to the threshold (tests/golden/find_borderline.py)
                    const double m = fabs((double)((nn * nn) / (dn * dn)) / 1e-11 - 1.0);
                    if (m < g_min_stop_margin) g_min_stop_margin = m;
                }
                if ((nn * nn) / (dn * dn) < 1e-11f) break;
            }
            beta.swap(next);
            result_in_next = false;
        }
        if (!result_in_next) beta.swap(next);       // `next` holds the result from here on
        mm_iters[it] = executed;
        for (size_t row = 0; row < (size_t)N * K; row++) {
            const bool keep = few || live[row];
            for (int d = 0; d < K; d++) alpha[row * K + d] = keep ? next[row * K + d] : alpha_old[row * K + d];
        }
        // ---- v from the responsibilities BEFORE this E-step
        for (size_t i = 0; i < (size_t)N * K; i++) v[i] = tclip::log_f32(cs[i] / (float)Q + kEps) + 1.0f;
        // ---- E-step
        std::vector<float> rowc(K), logit(K), ex(K);
        for (int n = 0; n < N; n++) {
            for (int k = 0; k < K; k++) {
                const float* ar = alpha + ((size_t)n * K + k) * K;
                const float l1 = tclip::lgamma_f32(sum_inner(K, [&](long d) { return ar[d]; }));
                const float l2 = -sum_inner(K, [&](long d) { return tclip::lgamma_f32(ar[d]); });
                rowc[k] = l1 + l2;
            }
            for (int q = 0; q < Q; q++) {
                const float* lz = &logz[((size_t)n * Q + q) * K];
                float mx = -INFINITY;
                for (int k = 0; k < K; k++) {
                    const float* ar = alpha + ((size_t)n * K + k) * K;
                    const float l3 = sum_inner(K, [&](long d) { return (ar[d] - 1.0f) * lz[d]; });
                    const float pen = ((float)lambd * v[(size_t)n * K + k]) / (float)Q;
                    logit[k] = (rowc[k] + l3) + pen;
                    mx = logit[k] > mx ? logit[k] : mx;
                }
                for (int k = 0; k < K; k++) ex[k] = tclip::exp_f32_sleef(logit[k] - mx);
                const float inv = 1.0f / sum_reduce_all(ex.data(), K);
                float* ur = u + ((size_t)n * Q + q) * K;
                int best = 0;
                for (int k = 0; k < K; k++) {
                    ur[k] = ex[k] * inv;
                    if (ur[k] > ur[best]) best = k;
                }
                if (hard) {
                    for (int k = 0; k < K; k++) ur[k] = 0.0f;
                    ur[best] = 1.0f;
                }
                if (argmax_trace) argmax_trace[((size_t)it * N + n) * Q + q] = (int16_t)best;
            }
        }
        // ---- convergence record: mean_n ||alpha_old - alpha||_F / ||alpha_old||_F
        float csum = 0;   // torch mean over N floats (N small): sequential is what sum_inner gives below
        std::vector<float> ratios(N);
        // torch's x.norm(dim=(1,2)): one serial pass, eight fused accumulators by index mod 8, added in order, then the
        // n mod 8 tail (first four as product + add, the rest fused), one correctly rounded square root
        auto norm8 = [](const float* x, size_t n) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const size_t nv = n & ~(size_t)7;
            for (size_t d = 0; d < nv; d++) acc[d & 7] = __builtin_fmaf(x[d], x[d], acc[d & 7]);
            float b = acc[0];
            for (int l = 1; l < 8; l++) b += acc[l];
            size_t d = nv;
            if (n - d >= 4) for (int k = 0; k < 4; k++, d++) b = b + x[d] * x[d];
            for (; d < n; d++) b = __builtin_fmaf(x[d], x[d], b);
            return sqrtf(b);
        };
        std::vector<float> diff((size_t)K * K);
        for (int n = 0; n < N; n++) {
            const size_t off = (size_t)n * K * K;
            for (size_t i = 0; i < (size_t)K * K; i++) diff[i] = alpha_old[off + i] - alpha[off + i];
            ratios[n] = norm8(diff.data(), (size_t)K * K) / norm8(&alpha_old[off], (size_t)K * K);
        }
        csum = sum_inner(N, [&](long n) { return ratios[n]; });
        criterions[it] = (few && hard) ? 0.0f : csum / (float)N;
        memcpy(alpha_old.data(), alpha, NKK * sizeof(float));
    }
    return 0;
}

// SOFT_KMEANS (src/methods/zero_shot/soft_kmeans.py:105-220), one batch; w [N,K,K] centroids out.
int tclip_oracle_soft_kmeans(const float* z, int N, int Q, int K, int iters, int temperature, float* u, float* w,
                             int16_t* argmax_trace) {
    const size_t NQK = (size_t)N * Q * K;
    for (size_t i = 0; i < NQK; i++) u[i] = z[i];
    std::vector<float> logit(K), ex(K);
    for (int it = -1; it < iters; it++) {            // it = -1: w_init (every row written)
        for (int n = 0; n < N; n++)
            for (int k = 0; k < K; k++) {
                const float* un = u + (size_t)n * Q * K;
                const float* zn = z + (size_t)n * Q * K;
                const float c = sum_outer(Q, k, K, [&](long q) { return un[q * K + k]; });
                const bool alive = c > kEps;
                const float den = c < kEps ? kEps : c;
                float* wr = w + ((size_t)n * K + k) * K;
                for (int d = 0; d < K; d++) {
                    const float t = sum_outer(Q, (long)k * K + d, (long)K * K,
                                              [&](long q) { return zn[q * K + d] * un[q * K + k]; }) / den;
                    if (it < 0) wr[d] = t;
                    else wr[d] = alive ? (t * 1.0f + wr[d] * 0.0f) : (t * 0.0f + wr[d] * 1.0f);
                }
            }
        if (it < 0) continue;
        for (int n = 0; n < N; n++)
            for (int q = 0; q < Q; q++) {
                const float* zq = z + ((size_t)n * Q + q) * K;
                float mx = -INFINITY;
                for (int k = 0; k < K; k++) {
                    const float* wr = w + ((size_t)n * K + k) * K;
                    const float s = sum_inner(K, [&](long d) { const float df = wr[d] - zq[d]; return df * df; });
                    logit[k] = (float)temperature * (-0.5f * s);
                    mx = logit[k] > mx ? logit[k] : mx;
                }
                for (int k = 0; k < K; k++) ex[k] = tclip::exp_f32_sleef(logit[k] - mx);
                const float inv = 1.0f / sum_reduce_all(ex.data(), K);
                float* ur = u + ((size_t)n * Q + q) * K;
                int best = 0;
                for (int k = 0; k < K; k++) {
                    ur[k] = ex[k] * inv;
                    if (ur[k] > ur[best]) best = k;
                }
                if (argmax_trace) argmax_trace[((size_t)it * N + n) * Q + q] = (int16_t)best;
            }
    }
    return 0;
}

// exposed for unit tests of the reduction-order emulation
float tclip_oracle_sum_inner(const float* x, long n) { return sum_inner(n, [&](long i) { return x[i]; }); }
float tclip_oracle_sum_outer(const float* x, long n, long col, long ncols) { return sum_outer(n, col, ncols, [&](long i) { return x[i]; }); }
float tclip_oracle_sum_reduce_all(const float* x, long n) { return sum_reduce_all(x, n); }

}  // extern "C"

// softmax over one row as torch's CPU kernel does it (vec_softmax_lastdim: max, Sleef expf of
// the shifted row, reduce_all sum, one reciprocal, multiply); exposed for the unit tests.
extern "C" void tclip_oracle_softmax_row(const float* x, float* out, long n) {
    float mx = x[0];
    for (long i = 1; i < n; i++) mx = x[i] > mx ? x[i] : mx;
    for (long i = 0; i < n; i++) out[i] = tclip::exp_f32_sleef(x[i] - mx);
    const float inv = 1.0f / sum_reduce_all(out, n);
    for (long i = 0; i < n; i++) out[i] *= inv;
}

extern "C" double tclip_oracle_min_stop_margin(int reset) {
    const double m = g_min_stop_margin;
    if (reset) g_min_stop_margin = 1e300;
    return m;
}

// Third design-study hook (scripts/dead_row_cycles.py): the trajectory of ONE dead row (y = -10 everywhere, em_dirichlet.py:222-223)
// started at `row`: mu = first iteration whose state is met again later (length of the transient), period = length of the
// limit cycle of the fp32 map; -1 / -1 when no state repeats within max_iter iterations.
extern "C" void tclip_oracle_dead_row_cycle(const float* row, int K, int max_iter, int32_t* mu, int32_t* period) {
    std::vector<float> cur(row, row + K), next(K), y(K, -10.0f);
    std::vector<uint64_t> seen;
    std::vector<std::vector<float>> states;
    *mu = -1;
    *period = -1;
    for (int l = 0; l <= max_iter; l++) {
        const uint64_t h = row_hash(cur.data(), K);
        for (int j = (int)seen.size() - 1; j >= 0; j--)
            if (seen[j] == h && memcmp(states[j].data(), cur.data(), sizeof(float) * K) == 0) {
                *mu = j;
                *period = l - j;
                return;
            }
        seen.push_back(h);
        states.push_back(cur);
        mm_step_row(cur.data(), y.data(), next.data(), K);
        cur.swap(next);
    }
}
// ... and the state after `iters` iterations of that trajectory (out: K floats), for a look at what the cycle's states consist of
extern "C" void tclip_oracle_dead_row_state(const float* row, int K, int iters, float* out) {
    std::vector<float> cur(row, row + K), next(K), y(K, -10.0f);
    for (int l = 0; l < iters; l++) {
        mm_step_row(cur.data(), y.data(), next.data(), K);
        cur.swap(next);
    }
    memcpy(out, cur.data(), sizeof(float) * K);
}
