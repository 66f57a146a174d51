// HIP kernels (gfx950 / MI355X) and C ABI of the EM-Dirichlet engine.  See include/tclip.h for the
// contract and DESIGN.md for the data layout and the per-kernel rooflines.
//
// Stream of one outer iteration (reference: src/methods/zero_shot/em_dirichlet.py:214-244):
//   k_cluster_sizes   cs = sum_q u, live mask, v                                   (:217-218, :151)
//   k_mstats          y = u^T log z / cs  (+ support statistics in few-shot)       (:219-222)
//                     (k_mstats_cols75 for the reference's 75 queries: 64 feature columns staged once per block)
//   k_build_rows      compact the rows that must iterate: live rows; dead rows whose stop-test
//                     contributions are not cached yet
//   k_mm_live x20     <=50 majorize-minimize iterations per launch for the live rows, alpha rows
//                     register-resident, large-argument lgamma work queued block-wide in LDS
//   k_mm_split        the same for the live rows from the second outer iteration on: every element executes only what its
//                     value class a+1 < 2.3 / [2.3, 10) / >= 10 needs, through three dense per-wavefront LDS queues
//                     (both also compiled with the row length as a constant for K = 1000 / 397 / 100, launch_mm)
//   k_mm_live<dead>   the same for the listed dead rows (y = -10, iterate kept in a scratch copy),
//   k_mm_probe        after chunk 0: limit-cycle detection that spares them the remaining chunks
//   k_mm_decide x20   batch-global stop test on device, no host round trip          (:157-177)
//   k_row_consts      lgamma(sum alpha) - sum lgamma(alpha) for rows that changed  (:35-36)
//   k_logits          (alpha-1) . log z contraction for rows that changed          (:37-38)
//   k_softmax         u = softmax_k(logit + lambd*v/Q), argmax, one-hot if hard    (:142-143)
//   k_criterion(+_mean) per-task relative change of alpha; alpha_old <- alpha      (:236-239)
//
// The other methods behind the same boundary reuse k_cluster_sizes, k_mstats*,
// k_softmax and add:
//   k_kmeans_logits_tile / k_kmeans_logits_rows   squared distances to the centroids (SOFT/HARD_KMEANS, EM_GAUSSIAN, PADDLE,
//                     BDCSPN): one lane per class on a 64-centroid LDS tile (rows of 32 .. 511 elements) / 32 lanes per class
//   k_mstats*<true>, k_cov_logits_rows   inverse diagonal covariances, Mahalanobis + log-det logits (EM_GAUSSIAN_COV)
//   k_kl_centroids, k_kl_divergences, k_argmin_rows, k_hard_assign   KL_KMEANS / HARD_KMEANS assignment
//   k_support_stats, k_div_rows   support class means (few-shot EM-Dirichlet, PADDLE, BDCSPN)
//   k_gather_log_features   u = z and log z read from a feature table through index tensors (tclip_em_dirichlet_run_tasks)
//   k_col_mean, k_bdcspn_normalize, k_bdcspn_eta   BD-CSPN normalisation (torch's norm order) and query shift
//   k_argmax_rows   inductive CLIP baseline
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/tclip.h"
#include "tclip_device.h"
#include "tclip_pk.h"
#include "tclip_selftest_inputs.h"

#ifndef TCLIP_G64_MIN_K
#define TCLIP_G64_MIN_K 897           // rows from this length on: one wavefront per row, 16 registers per lane instead of 32 lanes x 32
                                      // registers (half the code, 4 instead of 3 wavefronts per SIMD): K = 1000 bench shape 12.55 -> 11.90 s
#endif

namespace tclip {

// ------------------------------------------------------------------------------------------
// element-wise log of the features: log(x + 1e-15)                      (em_dirichlet.py:38)
// `nonfinite` (optional) is raised when some log is NaN or infinite (features off the simplex).
__global__ void k_log_features(const float* __restrict__ x, float* __restrict__ out, size_t n, int32_t* __restrict__ nonfinite) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = log_f32(x[i] + kEpsF);
        out[i] = v;
        bad = bad || (f32_bits(v) & 0x7f800000u) == 0x7f800000u;
    }
    if (nonfinite && bad) atomicOr(nonfinite, 1);
}

// The E-step terms of the initial alpha = 1: (alpha - 1) . log z is a sum of zeros, +0 in torch's order whenever every
// log z is finite, so logit0[t,q,k] = rowc[t,k] + 0 = rowc[t,k] - a fill instead of T*K*Q dot products of length K
// (K = 1000, 417 tasks: 75 ms per call).  With a non-finite log z (0 * inf = NaN) the general kernel runs instead.
__global__ void k_init_logits(const float* __restrict__ rowc, const int32_t* __restrict__ nonfinite, int Q, int K, size_t n,
                              float* __restrict__ logit0) {
    if (*nonfinite) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t t = i / ((size_t)Q * K);
        logit0[i] = rowc[t * K + i % K] + 0.0f;
    }
}

__global__ void k_fill(float* __restrict__ p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_copy(const float* __restrict__ s, float* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

__global__ void k_one_row_list(int32_t* __restrict__ rows, int32_t* __restrict__ n) { rows[0] = 0; n[0] = 1; }

// Where the feature rows of a call come from: a dense (T, R, K) tensor (idx == nullptr), or rows of a feature TABLE named by
// an index tensor, optionally with a per-task column permutation - the task-batch loop's `all_features[indices, :]`
// (eval_few_shot.py:233-241) and Tasks_Generator_few_shot.get_task's `data[:, unique_labels]`
// (task_generator_few_shot.py:41-52) read in place instead of materialised.
struct RowSrc {
    const float* base;       // dense: row r at base + r K;  indexed: the table
    const int64_t* idx;      // [T * R] table rows, or nullptr
    const int32_t* cols;     // [T, K]: column d of task t is table column cols[t K + d], or nullptr (identity)
};
static RowSrc dense_rows(const float* x) { return RowSrc{x, nullptr, nullptr}; }

// u = z and log z = log(z + 1e-15) of an indexed source, one pass over the table rows (em_dirichlet.py:38, :206)
__global__ __launch_bounds__(256) void k_gather_log_features(RowSrc src, int rows_per_task, int K, size_t n_rows, float* __restrict__ u,
                                                             float* __restrict__ logz, int32_t* __restrict__ nonfinite) {
    bool bad = false;
    for (size_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const float* row = src.base + (size_t)src.idx[r] * K;
        const int32_t* c = src.cols ? src.cols + (r / rows_per_task) * K : nullptr;
        for (int d = threadIdx.x; d < K; d += blockDim.x) {
            const float x = row[c ? c[d] : d];
            const float v = log_f32(x + kEpsF);
            u[r * K + d] = x;
            logz[r * K + d] = v;
            bad = bad || (f32_bits(v) & 0x7f800000u) == 0x7f800000u;
        }
    }
    if (nonfinite && bad) atomicOr(nonfinite, 1);
}

// p[i] = p[0] for 0 < i < n
__global__ void k_broadcast_first(float* p, size_t n) {
    const float v = p[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (i) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// Few-shot support statistics, once per run (few_shot/em_dirichlet.py:196-200 recomputes them
// every iteration from an (N,S,K,K) temporary although they are constant):
//   cnt[t,k]   = sum_s 1[y_s = k]
//   sup[t,k,d] = sum_s 1[y_s = k] * log(x_s[d] + eps)   in torch's outer-sum order over ALL s
// The zero products only matter through the positions of the cascade dumps, so only the members
// of class k are visited and the dumps that fall between two members are replayed.
struct SparseCascade {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int lp, nfull, done = 0;   // done = largest block boundary already applied
    __device__ SparseCascade(int n) {
        const int cl = dev_ceil_log2(n) / 4;
        lp = cl > 4 ? cl : 4;
        nfull = (n >> lp) << lp;
    }
    // apply every dump with boundary position in (done, upto]
    __device__ void dumps_upto(int upto) {
        if (upto > nfull) upto = nfull;
        const int step = 1 << lp;
        int e1 = ((done >> lp) + 1) << lp;                 // next level-1 boundary
        if (e1 > upto) return;
        a1 += a0; a0 = 0.f;
        const int s2 = step << lp;
        int e2 = ((e1 + s2 - 1) / s2) * s2;                // first level-2 boundary >= e1
        if (e2 <= upto) {
            a2 += a1; a1 = 0.f;
            const int s3 = s2 << lp;
            int e3 = ((e2 + s3 - 1) / s3) * s3;
            if (e3 <= upto) { a3 += a2; a2 = 0.f; }
        }
        done = (upto >> lp) << lp;
    }
    __device__ void add(int pos, float v) { dumps_upto(pos); a0 += v; }
    __device__ float finish() {
        dumps_upto(nfull);
        float r = a0 + a1;
        r += a2;
        r += a3;
        return r;
    }
};

// NOTE on SparseCascade::dumps_upto: between two members several level-1 boundaries may pass; only
// the first moves data (a0 is zero afterwards), likewise for the higher levels, but a level-2 dump
// can also happen at a LATER boundary than the first level-1 dump (first multiple of step^2 after
// it).  Because a1 then holds exactly what the first dump put there, replaying "first level-1
// boundary, then first level-2 boundary >= it, then first level-3 boundary >= that" is equivalent.

__global__ void k_support_stats(RowSrc src, const int64_t* __restrict__ ys, int S, int K, int take_log,
                                float* __restrict__ sup, float* __restrict__ cnt) {
    extern __shared__ int members[];               // indices s with ys == k, ascending
    __shared__ int n_members;
    const int t = blockIdx.y, k = blockIdx.x;
    const int64_t* yt = ys + (size_t)t * S;
    __shared__ int wave_count[16];
    if (threadIdx.x == 0) n_members = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63, n_waves = (blockDim.x + 63) >> 6;
    for (int s0 = 0; s0 < S; s0 += blockDim.x) {   // order-preserving compaction of {s : y_s == k}
        const int s = s0 + threadIdx.x;
        const bool m = s < S && yt[s] == k;
        const unsigned long long bal = __ballot(m);
        if (wl == 0) wave_count[wave] = __popcll(bal);
        __syncthreads();
        int off = n_members;
        for (int w = 0; w < wave; w++) off += wave_count[w];
        if (m) members[off + __popcll(bal & ((1ull << wl) - 1ull))] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < n_waves; w++) tot += wave_count[w];
            n_members += tot;
        }
        __syncthreads();
    }
    const int nm = n_members;
    const float* xt = src.base + (src.idx ? 0 : (size_t)t * S * K);
    const int64_t* it = src.idx ? src.idx + (size_t)t * S : nullptr;
    const int32_t* ct = src.cols ? src.cols + (size_t)t * K : nullptr;
    auto fetch = [&](int s_, int d_) { return xt[(size_t)(it ? it[s_] : (int64_t)s_) * K + (ct ? ct[d_] : d_)]; };
    const long ncols = (long)K * K;
    for (int d = threadIdx.x; d < K; d += blockDim.x) {
        const long col = (long)k * K + d;
        float r;
        if (outer_column_is_cascade(col, ncols)) {
            SparseCascade c(S);
            for (int i = 0; i < nm; i++) {
                const float x = fetch(members[i], d);
                c.add(members[i], take_log ? log_f32(x + kEpsF) : x);
            }
            r = c.finish();
        } else {   // 4 interleaved cascades over s/4, leftovers (s >= 4*(S/4)) into partial 0
            const int size_ilp = S >> 2;
            SparseCascade c0(size_ilp), c1(size_ilp), c2(size_ilp), c3(size_ilp);
            float extra = 0.f;
            bool has_extra = false;
            float p0 = 0.f;
            // leftovers are added after partial 0's cascade is complete, in order
            for (int i = 0; i < nm; i++) {
                const int s = members[i];
                const float x = fetch(s, d);
                const float v = take_log ? log_f32(x + kEpsF) : x;
                if (s >= size_ilp * 4) {
                    if (!has_extra) { p0 = c0.finish(); has_extra = true; }
                    p0 += v;
                    (void)extra;
                } else {
                    const int m = s >> 2;
                    switch (s & 3) {
                        case 0: c0.add(m, v); break;
                        case 1: c1.add(m, v); break;
                        case 2: c2.add(m, v); break;
                        default: c3.add(m, v); break;
                    }
                }
            }
            if (!has_extra) p0 = c0.finish();
            p0 += c1.finish();
            p0 += c2.finish();
            p0 += c3.finish();
            r = p0;
        }
        sup[((size_t)t * K + k) * K + d] = r;
    }
    if (threadIdx.x == 0) cnt[(size_t)t * K + k] = (float)nm;
}

// ------------------------------------------------------------------------------------------
// cs[t,k] = sum_q u[t,q,k] (torch outer-sum order), live mask, v = log(cs/Q + eps) + 1.
// Also invalidates the dead-row cache of rows that are alive.
__global__ void k_cluster_sizes(const float* __restrict__ u, int T, int Q, int K, int zero_shot,
                                float* __restrict__ cs, uint8_t* __restrict__ live, float* __restrict__ v,
                                int32_t* __restrict__ cache_len) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * K) return;
    const int t = i / K, k = i % K;
    const float* ut = u + (size_t)t * Q * K;
    const float c = dsum_outer(Q, k, K, [&](int q) { return ut[(size_t)q * K + k]; });
    cs[i] = c;
    const bool alive = zero_shot ? (c > kEpsF) : true;
    live[i] = alive;
    if (alive && cache_len) cache_len[i] = 0;
    if (v) v[i] = log_f32(c / (float)Q + kEpsF) + 1.0f;
}

// y[t,k,d] = sum_q u[t,q,k] * f[t,q,d] / max(cs, eps)                         (zero-shot, live rows)
//          = (sup[t,k,d] + sum_q u f) * (1 / (cnt[t,k] + cs[t,k]))           (few-shot)
//          = (sum_q u f + sup[t,k,d]) / (cs[t,k] + cnt[t,k])                 (PADDLE centroids)
// The same kernel, with f = raw features and u = one-hot predictions, gives the cluster
// prototypes of the accuracy tail (em_dirichlet.py:66-67).
// kCov: the inverse diagonal covariances of EM_GAUSSIAN_COV (em_gaussian_cov.py:172-193),
//          y[t,k,d] = cs[t,k] / max(sum_q (wc[t,k,d] - f[t,q,d])^2 * u[t,q,k], eps)
template <bool kCov>
__global__ void k_mstats(const float* __restrict__ u, const float* __restrict__ f, const float* __restrict__ cs,
                         const uint8_t* __restrict__ live, const float* __restrict__ sup,
                         const float* __restrict__ cnt, int Q, int K, float* __restrict__ y, int paddle, int k_first,
                         const float* __restrict__ wc) {
    const int t = blockIdx.z, k = blockIdx.y + k_first;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = (size_t)t * K + k;
    if (d >= K || !live[row]) return;
    const float* ut = u + (size_t)t * Q * K + k;
    const float* ft = f + (size_t)t * Q * K + d;
    const float wcv = kCov ? wc[row * K + d] : 0.0f;
    float s = dsum_outer(Q, (long)k * K + d, (long)K * K, [&](int q) {
        if (kCov) {
            const float df = wcv - ft[(size_t)q * K];
            return (df * df) * ut[(size_t)q * K];
        }
        return ut[(size_t)q * K] * ft[(size_t)q * K];
    });
    const float c = cs[row];
    if (kCov) {
        y[row * K + d] = c / (s < kEpsF ? kEpsF : s);
    } else if (paddle == 2) {      // BD-CSPN rectified prototypes (few_shot/bdcspn.py:139-141): plain quotient
        y[row * K + d] = s / c;
    } else if (sup && paddle) {           // PADDLE centroid (few_shot/paddle.py:154-158): (sum_q u z + support sum) / (sum_q u + count)
        y[row * K + d] = (s + sup[row * K + d]) / (c + cnt[row]);
    } else if (sup) {
        const float w = 1.0f / (cnt[row] + c);
        y[row * K + d] = w * (sup[row * K + d] + s);
    } else {
        y[row * K + d] = s / (c < kEpsF ? kEpsF : c);
    }
}

// The same statistics for kMstatsRows consecutive classes per thread, for the rows whose columns all
// take the cascade order (every row but the last few): the feature block of a task is then read
// from L2 once per 8 classes instead of once per class, the u values are wave-uniform loads, and
// each output keeps its own cascade state, so the sums are the ones dsum_cascade builds.
constexpr int kMstatsRows = 8;
template <bool kCov>
__global__ __launch_bounds__(64) void k_mstats_rows(const float* __restrict__ u, const float* __restrict__ f,
                                                    const float* __restrict__ cs, const uint8_t* __restrict__ live,
                                                    const float* __restrict__ sup, const float* __restrict__ cnt, int Q,
                                                    int K, float* __restrict__ y, int paddle, const float* __restrict__ wc) {
    const int t = blockIdx.z, k0 = blockIdx.y * kMstatsRows;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= K) return;
    {   // nothing is stored for rows that are not live: skip the group when none of its rows is (block-uniform).
        // After the first E-step most classes of a task are empty (K = 1000: ~40 of 1000 live; the accuracy
        // tail's one-hot statistics: <= 10 clusters), so most groups end here.
        bool any = false;
#pragma unroll
        for (int j = 0; j < kMstatsRows; j++) any = any || live[(size_t)t * K + k0 + j];
        if (!any) return;
    }
    const float* ut = u + (size_t)t * Q * K + k0;
    const float* ft = f + (size_t)t * Q * K + d;
    float wcv[kMstatsRows];
#pragma unroll
    for (int j = 0; j < kMstatsRows; j++) wcv[j] = kCov ? wc[((size_t)t * K + k0 + j) * K + d] : 0.0f;
    auto term = [&](int j, float uv, float fv) {
        if (kCov) {
            const float df = wcv[j] - fv;
            return (df * df) * uv;
        }
        return uv * fv;
    };
    const int cl = dev_ceil_log2(Q) / 4;
    const int level_power = cl > 4 ? cl : 4;
    const int step = 1 << level_power, mask = step - 1;
    float a0[kMstatsRows], a1[kMstatsRows], a2[kMstatsRows], a3[kMstatsRows];
#pragma unroll
    for (int j = 0; j < kMstatsRows; j++) a0[j] = a1[j] = a2[j] = a3[j] = 0.0f;
    int i = 0;
    for (; i + step <= Q;) {
        for (int jj = 0; jj < step; ++jj, ++i) {
            const float fv = ft[(size_t)i * K];
#pragma unroll
            for (int j = 0; j < kMstatsRows; j++) a0[j] += term(j, ut[(size_t)i * K + j], fv);
        }
        const bool l2 = (i & (mask << level_power)) == 0, l3 = l2 && (i & (mask << (2 * level_power))) == 0;
#pragma unroll
        for (int j = 0; j < kMstatsRows; j++) {
            a1[j] += a0[j]; a0[j] = 0.0f;
            if (l2) { a2[j] += a1[j]; a1[j] = 0.0f; }
            if (l3) { a3[j] += a2[j]; a2[j] = 0.0f; }
        }
    }
    for (; i < Q; ++i) {
        const float fv = ft[(size_t)i * K];
#pragma unroll
        for (int j = 0; j < kMstatsRows; j++) a0[j] += term(j, ut[(size_t)i * K + j], fv);
    }
#pragma unroll
    for (int j = 0; j < kMstatsRows; j++) {
        const size_t row = (size_t)t * K + k0 + j;
        if (!live[row]) continue;
        float s = a0[j];
        s += a1[j];
        s += a2[j];
        s += a3[j];
        const float c = cs[row];
        if (kCov) {
            y[row * K + d] = c / (s < kEpsF ? kEpsF : s);
        } else if (paddle == 2) {
            y[row * K + d] = s / c;
        } else if (sup && paddle) {
            y[row * K + d] = (s + sup[row * K + d]) / (c + cnt[row]);
        } else if (sup) {
            const float w = 1.0f / (cnt[row] + c);
            y[row * K + d] = w * (sup[row * K + d] + s);
        } else {
            y[row * K + d] = s / (c < kEpsF ? kEpsF : c);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Row lists for one outer iteration.  live_rows: rows whose alpha will change (k_mm_live iterates
// them, and their E-step terms must be recomputed); dead_rows: dead rows whose cached stop-test
// terms are incomplete (k_mm_live<.., true> iterates them).  Order inside the lists is irrelevant to the results.
__device__ __forceinline__ int lanes_below_mask(unsigned long long m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

__global__ void k_build_rows(const uint8_t* __restrict__ live, const int32_t* __restrict__ cache_len, int n_rows,
                             int n_checks, int32_t* __restrict__ dead_rows, int32_t* __restrict__ live_rows,
                             int32_t* __restrict__ n_dead, int32_t* __restrict__ n_live) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n_rows;
    const bool alive = in && live[i];
    const bool need = in && !alive && n_checks > 0 && cache_len[i] < n_checks;
    // one atomic per wavefront and list: the 64 consecutive rows of a wavefront (same task, neighbouring classes) stay
    // together in the lists, so the kernels that sweep them touch the task's log z and y while they are in L2
    const unsigned long long ml = __ballot(alive), md = __ballot(need);
    const int lane64 = threadIdx.x & 63;
    int base_l = 0, base_d = 0;
    if (lane64 == 0) {
        if (ml) base_l = atomicAdd(n_live, __popcll(ml));
        if (md) base_d = atomicAdd(n_dead, __popcll(md));
    }
    base_l = __shfl(base_l, 0, 64);
    base_d = __shfl(base_d, 0, 64);
    if (alive) live_rows[base_l + lanes_below_mask(ml, lane64)] = i;
    if (need) dead_rows[base_d + lanes_below_mask(md, lane64)] = i;
}

// ------------------------------------------------------------------------------------------
// The hot kernel: MM iterations l0..l1 of em_dirichlet.py:157-177 for every listed row.
// One row = one 32-lane group; its K parameters stay in registers for the whole chunk
// (element d in register d/32 of lane d%32), the row sum is rebuilt in torch's association
// order every iteration, and at a checkpoint iteration the row's share of
// ||b'-b||^2 and ||b||^2 is emitted in fp64.
//   live rows iterate in place in `alpha`;
//   dead rows (zero-shot clusters without members) iterate with y = -10 in `beta_dead`
//   (their alpha row must keep its old value, em_dirichlet.py:224-226) and write their stop-test
//   terms to `cache`, which later outer iterations reuse instead of recomputing the same
//   trajectory (same start, same y => same numbers).
struct MMArgs {
    float* alpha;
    float* beta_dead;
    const float* y;
    const uint8_t* live;
    int32_t* cache_len;
    double* cache;          // [row][n_checks][2]
    double* rowpart;        // [row][2]
    const int32_t* rows;
    const int32_t* n_rows;
    const int32_t* stop;    // [B]
    unsigned long long* work_counter;   // optional: element-updates executed (instrumentation)
    // dead rows only: the list of the NEXT chunk.  The last kernel that handles a listed row in this chunk
    // (k_mm_live<.., true>, or k_mm_probe when it follows) appends the row if it still lacks checkpoints, so
    // that every dead-row launch sweeps a dense list: after the probe only ~10 % of the rows that died keep
    // iterating, and a launch over the original list ran its 8-row blocks with one or two rows active.
    int32_t* next_rows;
    int32_t* next_count;
    int K, rows_per_batch, chunk, l0, l1, has_check, n_checks;
    int keep_placement;     // k_mm_split: 1 = keep the class queues' placement across the iterations of a launch (the default),
                            // 0 = sort in every iteration as rounds 3-4 did (tclip_debug_set_split_keep_placement: same bits)
};

// One MM iteration of a row held in registers, in place.  If `measure`, also accumulates this
// lane's share of ||b'-b||^2 and ||b||^2 (fp64).
//
// lgamma(a+1) has two regimes in torch's Sleef routine: a cheap polynomial for a+1 < 2.3 and a
// ~250-instruction double-float Stirling evaluation above.  Most Dirichlet parameters of a row are
// small and only a few are large, so the wave first queues its large arguments in LDS (ballot +
// prefix count), evaluates the expensive branch on the DENSE queue (ceil(n/64) passes instead of
// E passes over mostly idle lanes), and then every lane picks its results back up.
// `queue` = 64*E floats of LDS private to this wave.
// y of a row: in registers for short rows, re-read from global memory (L1/L2 hits, read-only)
// every iteration for long rows, where 2 x E registers per lane would cost occupancy and spills.
#ifndef TCLIP_Y_REGS_MAX_E
#define TCLIP_Y_REGS_MAX_E 16
#endif
// Element held by register e of lane `lane` of a row's lane group.  G <= 32: element e G + lane.  G = 64 (rows of
// 897..1024 elements, one wavefront per row): lanes 0..31 hold the first 16 steps of 32 elements, lanes 32..63 the
// rest, so that each half accumulates its part of torch's 16-step cascade locally (see group_sum_torch_64).
template <int E, int G>
__device__ __forceinline__ int elem_of(int e, int lane) {
    if constexpr (G == 64) return (lane >> 5) * (E * 32) + e * 32 + (lane & 31);
    else return e * G + lane;
}
// registers below this index hold only elements inside the row (wave-uniform)
template <int E, int G>
__host__ __device__ constexpr int full_registers(int K) {
    if constexpr (G == 64) return K > E * 32 ? (K - E * 32) / 32 : 0;
    else return K / G;
}

template <int E, int G = kGroup, int kRegsMaxE = TCLIP_Y_REGS_MAX_E>
struct RowY {
    static constexpr bool kInRegs = E <= kRegsMaxE;
    float r[kInRegs ? E : 1];
    const float* g;      // row base in global memory, or nullptr for a dead row (y = -10)
    int lane, K, n_full; // n_full = K / 32: registers below it lie entirely inside the row (wave-uniform)
    __device__ __forceinline__ void load(const float* row_y, int lane_, int K_) {
        g = row_y; lane = lane_; K = K_; n_full = full_registers<E, G>(K_);
        if (kInRegs) {
#pragma unroll
            for (int e = 0; e < (kInRegs ? E : 1); e++) {
                const int d = elem_of<E, G>(e, lane);
                r[e] = d < K ? (g ? g[d] : -10.0f) : 0.0f;
            }
        }
    }
    // every y of the lane at once (the split kernel's phase C, rows whose y is not kept in registers): the loads are issued
    // together and waited for one by one, with nothing else of the memory pipe between them
    __device__ __forceinline__ void fetch_all(float (&out)[E]) const {
        if (!g) {                                   // a dead row (wave-uniform in the 64-lane layout, per lane group otherwise)
#pragma unroll
            for (int e = 0; e < E; e++) out[e] = -10.0f;
            return;
        }
        if constexpr (G == 64) {
            // one row per wavefront: its base is a scalar; a buffer descriptor over the row's K floats, one lane offset for all
            // E loads (the register index is the instruction's immediate offset) and the hardware's own range check for the
            // slots beyond the row (a read past num_records returns 0).  The asm keeps the loads in the iteration that uses
            // them: hoisted out of the MM loop they are 16 registers that live across the dense passes again.
            const float* base = g;                  // (the MM kernels make the row index of this layout a scalar: readfirstlane)
            asm volatile("" : "+s"(base));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, K * 4, 0x00020000);
            int voff = elem_of<E, G>(0, lane) * 4;
            asm volatile("" : "+v"(voff));  // (or the E sums voff + 128 e are loop invariants: sixteen registers instead of sixteen immediates)
#pragma unroll
            for (int e = 0; e < E; e++) out[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + e * 128, 0, 0));
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) out[e] = get(e);
        }
    }
    __device__ __forceinline__ float get(int e) const {
        if (kInRegs) return r[kInRegs ? e : 0];
        if (!g) return -10.0f;                       // dead rows: y is not read by lanes beyond the row either
        const int d = elem_of<E, G>(e, lane);
        if (e < n_full) return g[d];                 // uniform branch: no lane mask on the load
        return d < K ? g[d] : 0.0f;
    }
};

// Number of set bits of a wave mask below the calling lane (v_mbcnt_lo/hi: two instructions).
__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Sleef's large-argument lgamma on a dense pass of queued arguments: the fp64 form (tclip_math.h), and the
// double-float original for the whole pass when some lane's argument is one of the 3e-5 the fp64 form is not
// sure about.
__device__ __forceinline__ float lgamma_big_dense(float v) {
    bool sure;
    float r = lgamma_sleef_ge23_f64<true>(v, sure);
    if (__builtin_expect(__ballot(!sure) != 0ull, 0)) r = sure ? r : lgamma_sleef_ge23<true>(v);
    return r;
}

// Phase C of one MM iteration: every element's digamma, the cheap lgamma branch, the pick-up of
// the large-argument results queued at `queue[base...]` in ballot order, and the update algebra.
// Elements are advanced two at a time on the packed fp32 pipe (tclip_pk.h); an odd last one
// takes the scalar form.
#ifndef TCLIP_MM_PACKED
#define TCLIP_MM_PACKED 1
#endif
template <int E, int G = kGroup>
__device__ __forceinline__ void mm_apply_updates(float (&beta)[E], const RowY<E, G>& yv, int K, int lane, float psi_s,
                                                 const LogTabEntry* tab, const float* queue, int base, bool measure,
                                                 double& num, double& den) {
    const int n_full = full_registers<E, G>(K);    // registers below it hold only elements of the row (wave-uniform)
    // y of rows that do not keep it in registers: fetched here, all at once (see split_apply_updates)
    constexpr bool kYLocal = !RowY<E, G>::kInRegs;
    float yl[kYLocal ? E : 1];
    if constexpr (kYLocal) yv.fetch_all(yl);
    auto y_of = [&](int e) { if constexpr (kYLocal) return yl[e]; else return yv.get(e); };
    // finishes pair p (the square root's table entries were requested one pair ago), stores and measures it
    auto finish = [&](int p, const PkUpdateStage& st) {
        const int e = 2 * p;
        const f2 a{beta[e], beta[e + 1]};
        const f2 nb = pk_mm_update_stage2(st);
        const bool full = e + 1 < n_full;
        const bool ok0 = full || elem_of<E, G>(e, lane) < K, ok1 = full || elem_of<E, G>(e + 1, lane) < K;
        if (measure) {
            const double d0 = (double)nb.x - (double)a.x, d1 = (double)nb.y - (double)a.y;
            if (ok0) { num += d0 * d0; den += (double)a.x * (double)a.x; }
            if (ok1) { num += d1 * d1; den += (double)a.y * (double)a.y; }
        }
        if (full) {                                // uniform branch: nothing to mask
            beta[e] = nb.x;
            beta[e + 1] = nb.y;
        } else {
            beta[e] = ok0 ? nb.x : 0.0f;
            beta[e + 1] = ok1 ? nb.y : 0.0f;
        }
    };
    PkUpdateStage pending;
#pragma unroll
    for (int p = 0; p < (TCLIP_MM_PACKED ? E / 2 : 0); p++) {
        const int e = 2 * p;
        const f2 a{beta[e], beta[e + 1]};
        const bool big0 = a.x + 1.0f >= 2.3f, big1 = a.y + 1.0f >= 2.3f;
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(big0);
        const float lg0 = big0 ? queue[base + lanes_below(m0)] : 0.0f;
        base += __popcll(m0);
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64(big1);
        const float lg1 = big1 ? queue[base + lanes_below(m1)] : 0.0f;
        base += __popcll(m1);
        const PkUpdateStage st = pk_mm_update_stage1(a, f2{y_of(e), y_of(e + 1)}, pk(psi_s), f2{lg0, lg1}, tab);
        if (p > 0) finish(p - 1, pending);         // while this pair's table look-ups are in flight
        pending = st;
    }
    if (TCLIP_MM_PACKED && E / 2 > 0) finish(E / 2 - 1, pending);
#pragma unroll
    for (int e = (TCLIP_MM_PACKED ? E / 2 * 2 : 0); e < E; e++) {
        const float a = beta[e];
        const float x1 = a + 1.0f;
        const bool big = x1 >= 2.3f;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(big);
        const float lg_big = big ? queue[base + lanes_below(m)] : 0.0f;
        base += __popcll(m);
        bool sure;
        float lg_small = lgamma_sleef_1_23_f64(big ? 2.0f : x1, sure);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!sure) != 0ull, 0)) lg_small = sure ? lg_small : lgamma_sleef_05_23(big ? 2.0f : x1);
        const float psi1 = digamma_xp1(a, tab);
        const float nb = mm_update_algebra(a, y_of(e), psi_s, psi1, big ? lg_big : lg_small);
        const bool ok = elem_of<E, G>(e, lane) < K;
        if (measure && ok) {
            const double df = (double)nb - (double)a;
            num += df * df;
            den += (double)a * (double)a;
        }
        beta[e] = ok ? nb : 0.0f;
    }
}

// Phase C for a wavefront that queued NOTHING for the large-argument lgamma in this iteration (every a + 1 of its rows is
// below 2.3 - wave-uniform, known from phase A's count): the same values as mm_apply_updates, without the per-pair ballots,
// ranks and LDS pick-up of the queue, without the stand-in argument of the polynomial lgamma and with the first eight steps
// of digamma's recurrence unmasked (pk_mm_update_stage1_small).  In the first outer iteration every parameter starts at 1:
// all wavefronts take this path for the first iterations, 44 % of the K = 1000 rows still do after 50 (HISTORY.md section 8);
// dead rows, whose parameters collapse to ~0.14 within a dozen iterations, always do.
#ifndef TCLIP_MM_SMALL_PATH
#define TCLIP_MM_SMALL_PATH 1
#endif
template <int E, int G = kGroup>
__device__ __forceinline__ void mm_apply_updates_small(float (&beta)[E], const RowY<E, G>& yv, int K, int lane, float psi_s,
                                                       const LogTabEntry* tab, bool measure, double& num, double& den) {
    const int n_full = full_registers<E, G>(K);
    constexpr bool kYLocal = !RowY<E, G>::kInRegs;
    float yl[kYLocal ? E : 1];
    if constexpr (kYLocal) yv.fetch_all(yl);
    auto y_of = [&](int e) { if constexpr (kYLocal) return yl[e]; else return yv.get(e); };
    auto finish = [&](int p, const PkUpdateStage& st) {
        const int e = 2 * p;
        const f2 a{beta[e], beta[e + 1]};
        const f2 nb = pk_mm_update_stage2(st);
        const bool full = e + 1 < n_full;
        const bool ok0 = full || elem_of<E, G>(e, lane) < K, ok1 = full || elem_of<E, G>(e + 1, lane) < K;
        if (measure) {
            const double d0 = (double)nb.x - (double)a.x, d1 = (double)nb.y - (double)a.y;
            if (ok0) { num += d0 * d0; den += (double)a.x * (double)a.x; }
            if (ok1) { num += d1 * d1; den += (double)a.y * (double)a.y; }
        }
        if (full) {
            beta[e] = nb.x;
            beta[e + 1] = nb.y;
        } else {
            beta[e] = ok0 ? nb.x : 0.0f;
            beta[e + 1] = ok1 ? nb.y : 0.0f;
        }
    };
    PkUpdateStage pending;
#pragma unroll
    for (int p = 0; p < E / 2; p++) {
        const int e = 2 * p;
        const PkUpdateStage st = pk_mm_update_stage1_small(f2{beta[e], beta[e + 1]}, f2{y_of(e), y_of(e + 1)}, pk(psi_s), tab);
        if (p > 0) finish(p - 1, pending);
        pending = st;
    }
    if (E / 2 > 0) finish(E / 2 - 1, pending);
    if (E & 1) {
        constexpr int e = E - 1;
        const float a = beta[e];
        const float x1 = a + 1.0f;
        bool sure;
        float lg = lgamma_sleef_1_23_f64(x1, sure);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!sure) != 0ull, 0)) lg = sure ? lg : lgamma_sleef_05_23(x1);
        const float nb = mm_update_algebra(a, y_of(e), psi_s, digamma_xp1(a, tab), lg);
        const bool ok = elem_of<E, G>(e, lane) < K;
        if (measure && ok) {
            const double df = (double)nb - (double)a;
            num += df * df;
            den += (double)a * (double)a;
        }
        beta[e] = ok ? nb : 0.0f;
    }
}

// Registers of an MM kernel instantiation that lie inside the row - and inside the 4-way interleaved part of torch's row
// sum - for EVERY row length the instantiation is launched for (launch_mm: the smallest E that covers K, so at most
// three registers of slack, the fourth for the upper half of the 64-lane layout, which starts at K = 897).
// KC > 0: the kernel is compiled for rows of exactly KC elements (launch_mm: the reference's datasets, K = 1000 / 397 / 100),
// and the count is exact: with the row length a constant the compiler resolves every step, tail and mask of the row sum,
// of the placement sweeps and of the update by itself - this constant and first_ragged_register are the two places where the
// kernels' own template arithmetic has to be told.
template <int E, int G, int KC = 0>
constexpr int sure_registers() {
    if (KC > 0) {
        const int size_ilp = (KC >> 3) >> 2;                               // steps of the 4-way interleaved part
        if (G == 64) return size_ilp - 16 < 0 ? 0 : (size_ilp - 16 > E ? E : size_ilp - 16);     // the upper half's step of register e is 16 + e
        if (G == 32) return size_ilp > E ? E : size_ilp;
        const int P = 4 / (G / 8);                                          // registers per step (group_sum_torch_g)
        return size_ilp * P > E ? E : size_ilp * P;
    }
    if (G == 64) return (TCLIP_G64_MIN_K >= 897 && E == 16) ? 12 : 0;      // 16 + e < K / 32 for K >= 897
    // G = 32: one register per step of 32 elements; G = 16: two.  The instantiation E is launched for K > (next smaller
    // size) x G, whose interleaved part has at least E - 4 registers' worth of steps for every E launch_mm_G hands out.
    // G = 8 has FOUR registers per step and buckets (10 -> 13 -> 16) whose smallest row ends inside register E - 4
    // (E = 13: K = 81 has size_ilp = 2, i.e. 8 sure registers, not 9): nothing is taken for sure there.
    if (G == 8) return 0;
    return E > 4 ? E - 4 : 0;
}
// registers below it hold only slots inside the row in every lane (the split kernel queues them whole)
template <int E, int G, int KC = 0>
constexpr int first_ragged_register() {
    if (KC > 0) return full_registers<E, G>(KC) > E ? E : full_registers<E, G>(KC);
    return E > 4 ? E - 4 : 0;
}

// the row sum in torch's order, valid in every lane of the row's lane group
template <int E, int G, int KC = 0>
__device__ __forceinline__ float row_sum_torch_all(const float (&x)[E], int K, int lane) {
    if constexpr (G == kGroup) return group_sum_torch<E, false, sure_registers<E, G, KC>()>(x, K, lane);
    else if constexpr (G == 64) return group_sum_torch_64<E, false, sure_registers<E, G, KC>()>(x, K, lane);
    else return group_sum_torch_g<E, G, false, sure_registers<E, G, KC>()>(x, K, lane);
}

template <int E, int G>
__device__ __forceinline__ void mm_iterate(float (&beta)[E], const RowY<E, G>& yv, int K, int lane,
                                           const LogTabEntry* tab, float* queue, bool measure, double& num,
                                           double& den) {
    const float s = row_sum_torch_all<E, G>(beta, K, lane);
    bool in_domain = fast_range_f32(s) && s <= 0x1p40f;
    uint32_t largest = 0u;
#pragma unroll
    for (int e = 0; e < E; e++) largest = max(largest, f32_bits(beta[e]));
    in_domain = in_domain & mm_fast_domain_of_max_bits(largest);
    if (__builtin_expect(!__all(in_domain), 0)) {   // NaN / inf / out of range somewhere in the wave
        const float psi_s = digamma_f32(s);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const float nb = mm_update_generic(beta[e], yv.get(e), psi_s);
            const bool ok = elem_of<E, G>(e, lane) < K;
            if (measure && ok) {
                const double df = (double)nb - (double)beta[e];
                num += df * df;
                den += (double)beta[e] * (double)beta[e];
            }
            beta[e] = ok ? nb : 0.0f;
        }
        return;
    }
    const float psi_s = digamma_pos_f32(s, tab);   // row sums are mostly >= 10: the recurrence loop is rarely entered
    // phase A: queue the arguments of the expensive lgamma branch
    int n_big = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const float x1 = beta[e] + 1.0f;
        const bool big = x1 >= 2.3f;
        const unsigned long long m = __ballot(big);
        if (big) queue[n_big + lanes_below(m)] = x1;
        n_big += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    // phase B: dense evaluation by the ACTIVE lanes (one half of the wave may be idle), results
    // overwrite the queue
    const unsigned long long active = __ballot(true);
    const int rank = lanes_below(active), n_active = __popcll(active);
    for (int start = 0; start < n_big; start += n_active) {
        const int idx = start + rank;
        const float v = idx < n_big ? queue[idx] : 8.0f;
        const float r = lgamma_big_dense(v);
        if (idx < n_big) queue[idx] = r;
    }
    __builtin_amdgcn_wave_barrier();
    // phase C: per element digamma, cheap lgamma branch, pick-up, algebra
    if (TCLIP_MM_SMALL_PATH && n_big == 0) mm_apply_updates_small<E, G>(beta, yv, K, lane, psi_s, tab, measure, num, den);
    else mm_apply_updates<E, G>(beta, yv, K, lane, psi_s, tab, queue, 0, measure, num, den);
    __builtin_amdgcn_wave_barrier();
}

#ifndef TCLIP_PROBE_CHUNKS
#define TCLIP_PROBE_CHUNKS 2        // the limit-cycle probe runs after each of the first this-many chunks
#endif
#ifndef TCLIP_MAX_CYCLE
#define TCLIP_MAX_CYCLE 64
#endif
constexpr int kMaxCycle = TCLIP_MAX_CYCLE;  // longest limit cycle looked for on dead rows (periods up to 20 seen at K=1000)

#ifndef TCLIP_MM_WAVES_LARGE
#define TCLIP_MM_WAVES_LARGE 3     // waves per SIMD requested for long rows (E > 8)
#endif
#ifndef TCLIP_MM_WAVES_SMALL
#define TCLIP_MM_WAVES_SMALL 4     // waves per SIMD requested for short rows (E <= 8)
#endif
// Limit-cycle probe for the dead rows that have just run chunk 0 (k_mm_live<.., true> left b_51 in
// `beta_dead` and the first stop-test pair in the cache).
// Dead rows (y = -10 everywhere) contract within ~10-30 iterations onto a short limit cycle of the
// fp32 map (periods 1..20 observed).  Once b_{l+p} == b_l bit for bit, the trajectory is periodic
// for ever, so every later checkpoint's (||b'-b||^2, ||b||^2) is one of the p pairs measured
// here: the remaining ~900 iterations of this row need not be executed.  No cycle within
// kMaxCycle steps: nothing is assumed, the row keeps iterating chunk by chunk.
template <int E, int G>
__global__ __launch_bounds__(256, (E > 16 ? TCLIP_MM_WAVES_LARGE : TCLIP_MM_WAVES_SMALL)) void k_mm_probe(MMArgs a) {
    __shared__ LogTabEntry tab[16];
    __shared__ double cyc[256 / G][kMaxCycle][2];
    __shared__ float lg_queue[4][64 * E];             // per wave: arguments / results of the large-x lgamma
    load_log_table(tab);
    float* queue = lg_queue[threadIdx.x >> 6];
    const int lane = threadIdx.x & (G - 1);
    const int group = threadIdx.x / G;
    const int groups_per_block = blockDim.x / G;
    const int n = *a.n_rows;
    const int K = a.K;
    for (int i = blockIdx.x * groups_per_block + group; i < n; i += gridDim.x * groups_per_block) {
        const int row = a.rows[i];
        if (a.stop[row / a.rows_per_batch]) continue;
        const int have = a.cache_len[row];
        if (have != a.chunk + 1) {                                  // not iterated in this chunk: its turn comes later
            if (a.next_rows && have < a.n_checks && lane == 0) a.next_rows[atomicAdd(a.next_count, 1)] = row;
            continue;
        }
        const float* ref = a.beta_dead + (size_t)row * K;          // b_{l1+1}
        float beta[E];
        RowY<E, G> yv;
        yv.load(nullptr, lane, K);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = elem_of<E, G>(e, lane);
            beta[e] = d < K ? ref[d] : 0.0f;
        }
        int period = 0;
        for (int j = 0; j < kMaxCycle && period == 0; j++) {
            double pn = 0.0, pd = 0.0;
            mm_iterate<E, G>(beta, yv, K, lane, tab, queue, true, pn, pd);
            pn = group_sum_f64_g<G>(pn);
            pd = group_sum_f64_g<G>(pd);
            if (lane == 0) { cyc[group][j][0] = pn; cyc[group][j][1] = pd; }
            bool same = true;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int d = elem_of<E, G>(e, lane);
                if (d < K) same = same && (beta[e] == ref[d]);
            }
            const unsigned long long bal = __ballot(same);
            constexpr unsigned long long kAll = G == 64 ? ~0ull : (1ull << (G & 63)) - 1ull;
            const unsigned long long mine = (bal >> ((threadIdx.x & 63) & ~(G - 1))) & kAll;      // this group's lanes
            if (mine == kAll) period = j + 1;
        }
        if (period && lane == 0) {
            for (int m = a.chunk + 1; m < a.n_checks; m++) {         // the checkpoints still ahead
                const int j = (50 * (m + 1) - (a.l1 + 1)) % period;
                double* c = a.cache + ((size_t)row * a.n_checks + m) * 2;
                c[0] = cyc[group][j][0];
                c[1] = cyc[group][j][1];
            }
            a.cache_len[row] = a.n_checks;
        }
        if (!period && a.next_rows && lane == 0) a.next_rows[atomicAdd(a.next_count, 1)] = row;
    }
}

// Early limit-cycle probe (round 6) for rows that have JUST died: k_mm_live<.., true> has run only the first `a.l0`
// iterations (TCLIP_DEAD_HEAD = 12, not the whole first chunk) and left b_{l0} in `beta_dead`.
// Measured on the reference's trajectories (scripts/dead_row_cycles.py, CPU oracle, the alpha rows of a task after its first
// outer iteration): a dead row is ON its cycle after 8..15 iterations at K = 100 (median 10; periods 1, 2, 4) and after
// 8..28 at K = 1000 (median 11, 99 % within 16; periods 20 (75 %), 5, 4, 8) - so the 51 iterations of chunk 0 that
// k_mm_probe waits for are mostly spent going round the cycle.  Here the probe starts at iteration l0, measures the pair
// (||b'-b||^2, ||b||^2) of EVERY iteration it executes (cyc[j]: iteration l0 + j) and compares the state with two snapshots:
// b_{l0} itself for the first 32 iterations and the state after them from then on, and the state 8 iterations in (kept in
// the row's y, which a dead row does not use).  A snapshot taken before the row reached its cycle is never met again, a later
// one is - and it must stand for a whole period (the first version renewed ONE snapshot after 4, 8, 16 and 32 iterations and met
// the 20-iteration cycles of K = 1000 only through the last one, after 52 iterations); the second snapshot finds the rows whose
// transient is a little longer than l0 (8 % at K = 397, hard) after 8 + p iterations instead of 32 + p.
// State s_{j+1} == snapshot s_js: the trajectory is periodic from iteration l0 + js on
// with period p = j + 1 - js, and the pair of iteration l >= l0 + js is cyc[js + (l - l0 - js) mod p] - every checkpoint of
// the row (l = 50, 100, ...; l0 + 32 <= 50 is checked on the host) is filled and the row is done, exactly as if it had run.
// Head lengths 12 / 14 / 16 measured (dead-row kernels of one call, us; K = 1000 | 397 hard | 100): 17 717 | 7 199 | 1 029,
// 18 680 | 7 232 | 1 059, 19 554 | 7 015 | 1 039 - 12 it is (scripts/gpu_dead_trace.py).
// No cycle within kMaxCycle iterations, or a row whose cache is partly filled (a batch that stopped early): the row goes to
// the list `next_rows`, which the host hands to the old path (k_mm_live<.., true> over the whole chunk from alpha, then
// k_mm_probe) - nothing is assumed about such a row.
#ifndef TCLIP_DEAD_HEAD
#define TCLIP_DEAD_HEAD 12
#endif
template <int E, int G>
__global__ __launch_bounds__(256, (E > 16 ? TCLIP_MM_WAVES_LARGE : TCLIP_MM_WAVES_SMALL)) void k_mm_probe_head(MMArgs a) {
    __shared__ LogTabEntry tab[16];
    __shared__ double cyc[256 / G][kMaxCycle][2];
    __shared__ float lg_queue[4][64 * E];
    load_log_table(tab);
    float* queue = lg_queue[threadIdx.x >> 6];
    const int lane = threadIdx.x & (G - 1);
    const int group = threadIdx.x / G;
    const int groups_per_block = blockDim.x / G;
    const int n = *a.n_rows;
    const int K = a.K;
    for (int i = blockIdx.x * groups_per_block + group; i < n; i += gridDim.x * groups_per_block) {
        const int row = a.rows[i];
        if (a.stop[row / a.rows_per_batch]) continue;
        if (a.cache_len[row] != 0) {                                // partly cached: the old path continues it where it stands
            if (lane == 0) a.next_rows[atomicAdd(a.next_count, 1)] = row;
            continue;
        }
        float* ref = a.beta_dead + (size_t)row * K;                // b_{l0}; replaced by the state 32 iterations on
        float* ref8 = const_cast<float*>(a.y) + (size_t)row * K;   // second snapshot, 8 iterations on: the y row of a dead row is
                                                                   // never read (y = -10) nor written by the M-step statistics
        float beta[E];
        RowY<E, G> yv;
        yv.load(nullptr, lane, K);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = elem_of<E, G>(e, lane);
            beta[e] = d < K ? ref[d] : 0.0f;
        }
        int period = 0, js = 0, base = 0;
        bool have8 = false;
        constexpr unsigned long long kAll = G == 64 ? ~0ull : (1ull << (G & 63)) - 1ull;
        const int shift = (threadIdx.x & 63) & ~(G - 1);                                          // this group's lanes in a ballot
        for (int j = 0; j < kMaxCycle && period == 0; j++) {
            double pn = 0.0, pd = 0.0;
            mm_iterate<E, G>(beta, yv, K, lane, tab, queue, true, pn, pd);
            pn = group_sum_f64_g<G>(pn);
            pd = group_sum_f64_g<G>(pd);
            if (lane == 0) { cyc[group][j][0] = pn; cyc[group][j][1] = pd; }
            bool same = true, same8 = have8;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int d = elem_of<E, G>(e, lane);
                if (d < K) same = same && (beta[e] == ref[d]);
            }
            if (have8) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int d = elem_of<E, G>(e, lane);
                    if (d < K) same8 = same8 && (beta[e] == ref8[d]);
                }
            }
            const bool hit = ((__ballot(same) >> shift) & kAll) == kAll, hit8 = ((__ballot(same8) >> shift) & kAll) == kAll;
            if (hit) {
                period = j + 1 - js;
                base = js;
            } else if (have8 && hit8) {
                period = j + 1 - 8;
                base = 8;
            } else if (j + 1 == 8 || j + 1 == 32) {                                                // a new snapshot: s_{j+1}
                float* dst = j + 1 == 8 ? ref8 : ref;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int d = elem_of<E, G>(e, lane);
                    if (d < K) dst[d] = beta[e];
                }
                if (j + 1 == 8) have8 = true;
                else js = 32;
            }
        }
        if (period && lane == 0) {
            for (int m = 0; m < a.n_checks; m++) {
                const int t = 50 * (m + 1) - a.l0;                  // cyc index of the checkpoint's iteration, were the window long enough
                const int j = base + (t - base) % period;
                double* c = a.cache + ((size_t)row * a.n_checks + m) * 2;
                c[0] = cyc[group][j][0];
                c[1] = cyc[group][j][1];
            }
            a.cache_len[row] = a.n_checks;
        }
        if (!period && lane == 0) a.next_rows[atomicAdd(a.next_count, 1)] = row;
    }
}

// ------------------------------------------------------------------------------------------
// Live rows: the same iteration, with the large-x lgamma queue shared by the whole BLOCK.
// A wave of two rows queues only ~10-20 large arguments per iteration, so the per-wave dense pass
// of mm_iterate still runs ~250 instructions with most lanes idle (measured: 28 % of the MM time
// at K = 100).  Here the 8 rows of a block advance in lockstep: every wave counts its large
// arguments, one barrier publishes the counts (and the fast-domain consensus), the waves write
// their entries at block-wide offsets, and the dense evaluation is spread over all 256 threads.
// Results are identical by construction: the same function is applied to the same arguments.
// Block-wide queue bookkeeping in LDS.  Every wave fills its own slice of the queue and posts its
// entry count for this iteration (count[turn & 1][wave]) before the single barrier that publishes
// both; counts are double-buffered and slices are wave-private outside the dense pass, so a wave
// that runs ahead into the next iteration cannot disturb one that is still picking up results.
// `bad` is raised by a wave that holds an argument outside the fast domain.
// `rowsum` / `psi`: the row sums of the block's rows and digamma of them, double-buffered like the counts:
// one wave evaluates digamma for all rows of the block in a single pass (one row per lane) inside the
// dense-pass window instead of every 32-lane group evaluating its own row's value 32 times over.
struct QueueCtl { int count[2][8]; int bad; float rowsum[2][64]; float psi[2][64]; };
#ifdef TCLIP_COUNT_SMALL
__device__ unsigned long long g_small_count[4];
#endif

// the row sum in torch's order, valid in lane 0 of the row's lane group (K >= 8; shorter rows: in every lane)
template <int E, int G, int KC = 0>
__device__ __forceinline__ float row_sum_torch(const float (&x)[E], int K, int lane) {
    if constexpr (G == kGroup) return group_sum_torch<E, true, sure_registers<E, G, KC>()>(x, K, lane);
    else if constexpr (G == 64) return group_sum_torch_64<E, true, sure_registers<E, G, KC>()>(x, K, lane);
    else return group_sum_torch_g<E, G, true, sure_registers<E, G, KC>()>(x, K, lane);
}

template <int E, int W, int R, int G, int KC = 0>
__device__ __forceinline__ void mm_iterate_block(float (&beta)[R][E], const RowY<E, G> (&yv)[R], int K, int lane,
                                                 const bool (&active)[R], const LogTabEntry* tab, float* queue, QueueCtl* ctl,
                                                 int turn, bool measure, double (&num)[R], double (&den)[R]) {
    constexpr int kGroups = (64 / G) * W;                        // lane groups (= rows per row set) of the block
    const int wave = threadIdx.x >> 6, lane64 = threadIdx.x & 63;
    float s[R];
    bool in_domain = true;
    // phase A: queue the arguments of the expensive lgamma branch in this wave's slice (row set by row set)
    float* slice = queue + wave * (64 * E * R);
    int base[R + 1];
    base[0] = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        s[r] = 16.0f;
        if (active[r]) {
            s[r] = row_sum_torch<E, G, KC>(beta[r], K, lane);
            in_domain = in_domain && (lane != 0 || (fast_range_f32(s[r]) && s[r] <= 0x1p40f));     // the sum lives in lane 0
            uint32_t largest = 0u;
#pragma unroll
            for (int e = 0; e < E; e++) largest = max(largest, f32_bits(beta[r][e]));
            in_domain = in_domain & mm_fast_domain_of_max_bits(largest);
        }
        if (lane == 0) ctl->rowsum[turn & 1][r * kGroups + (threadIdx.x / G)] = s[r];
        int idx = base[r];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const float x1 = beta[r][e] + 1.0f;
            const bool big = active[r] && x1 >= 2.3f;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(big);
            if (big) slice[idx + lanes_below(m)] = x1;
            idx += __popcll(m);
        }
        base[r + 1] = idx;
    }
    const bool wave_ok = __all(in_domain);
    if (lane64 == 0) {
        ctl->count[turn & 1][wave] = base[R];
        if (!wave_ok) ctl->bad = 1;
    }
    __syncthreads();
    int before[W + 1];                                          // entries queued by the waves in front of wave w
    before[0] = 0;
#pragma unroll
    for (int w = 0; w < W; w++) before[w + 1] = before[w] + ctl->count[turn & 1][w];
    const int n_big = before[W];
    const bool bad = ctl->bad != 0;
    if (__builtin_expect(bad, 0)) {                             // NaN / inf / out of range somewhere in the block
        __syncthreads();                                        // everyone has seen the flag
        if (threadIdx.x == 0) ctl->bad = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (!active[r]) continue;
            const float psi_s = digamma_f32(ctl->rowsum[turn & 1][r * kGroups + (threadIdx.x / G)]);   // lane 0's row sum
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float nb = mm_update_generic(beta[r][e], yv[r].get(e), psi_s);
                const bool ok = elem_of<E, G>(e, lane) < K;
                if (measure && ok) {
                    const double df = (double)nb - (double)beta[r][e];
                    num[r] += df * df;
                    den[r] += (double)beta[r][e] * (double)beta[r][e];
                }
                beta[r][e] = ok ? nb : 0.0f;
            }
        }
        __syncthreads();
        return;
    }
    // phase B: dense evaluation, results overwrite the queue.  Usually one pass of one wave covers
    // the whole queue; the wave that takes the first 64 entries rotates with the iteration so that
    // this work spreads over the four SIMDs of the CU (wave w of every block sits on SIMD w % 4).
    for (int start = ((wave + turn) % W) * 64; start < n_big; start += 64 * W) {
        const int j = start + lane64;                          // j-th entry of the block, slices in wave order
        int at = j;
#pragma unroll
        for (int w = 1; w < W; w++) at += j >= before[w] ? 64 * E * R - (before[w] - before[w - 1]) : 0;
        const float v = j < n_big ? queue[at] : 8.0f;
        const float r = lgamma_big_dense(v);
        if (j < n_big) queue[at] = r;
    }
    // digamma of the block's row sums: one lane per row, by the wave after the one that opens the dense pass
    if (wave == (W - (turn % W) + 1) % W && lane64 < kGroups * R)
        ctl->psi[turn & 1][lane64] = digamma_pos_f32(ctl->rowsum[turn & 1][lane64], tab);
    __syncthreads();
    // phase C: per element digamma, cheap lgamma branch, pick-up, algebra
#ifdef TCLIP_COUNT_SMALL
    // design study (scripts/gpu_small_count.py): [0] wavefront-iterations of k_mm_live, [1] those that queued nothing, [2] block-iterations,
    // [3] those in which NO wavefront of the block queued anything
    if (lane64 == 0) {
        atomicAdd(&g_small_count[0], 1ull);
        if (base[R] == 0) atomicAdd(&g_small_count[1], 1ull);
        if (wave == 0) {
            atomicAdd(&g_small_count[2], 1ull);
            if (n_big == 0) atomicAdd(&g_small_count[3], 1ull);
        }
    }
#endif
    if (TCLIP_MM_SMALL_PATH && base[R] == 0) {                   // nothing of this wavefront's rows is in the queue (wave-uniform)
#pragma unroll
        for (int r = 0; r < R; r++)
            if (active[r])
                mm_apply_updates_small<E, G>(beta[r], yv[r], K, lane, ctl->psi[turn & 1][r * kGroups + (threadIdx.x / G)], tab,
                                             measure, num[r], den[r]);
        return;
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        if (active[r])
            mm_apply_updates<E, G>(beta[r], yv[r], K, lane, ctl->psi[turn & 1][r * kGroups + (threadIdx.x / G)], tab, slice, base[r],
                                   measure, num[r], den[r]);
}

#ifndef TCLIP_MM_BLOCK_WAVES
#define TCLIP_MM_BLOCK_WAVES 4        // waves (= pairs of rows) per block of k_mm_live for E <= 8
#endif
#ifndef TCLIP_MM_ROWSETS
#define TCLIP_MM_ROWSETS 2            // rows per 32-lane group of k_mm_live for E <= 8: 16 rows share a block's two barriers and
                                      // its dense lgamma passes (fuller passes: +2 % on the K = 100 bench; 4 rows per group spill)
#endif
// kDead: the listed rows are dead rows whose cache ends at this chunk: y = -10, the iterate lives in
// `beta_dead` (their alpha keeps its value, em_dirichlet.py:224-226) and the stop-test pair goes to the cache.
// G: lanes per row (32; 16 or 8 for short rows, where a 32-lane group would leave lanes idle: K = 100 fills
// 100 of 128 slots as 32 x 4 but 100 of 104 as 8 x 13, with eight rows per wavefront sharing the per-row work).
template <int E, int W, bool kDead, int R, int G = kGroup, int KC = 0>
// wavefronts per SIMD: 4 (128 VGPRs) up to 16 registers per lane - also for K = 257..512 as 32 lanes x 10..16 since round 2
// (K = 397, hard, 1000 tasks: 1.24 -> 1.18 s) - 3 (168 VGPRs) for the 20..28-register kernels
__global__ __launch_bounds__(64 * W, (E > 16 ? TCLIP_MM_WAVES_LARGE : TCLIP_MM_WAVES_SMALL)) void k_mm_live(MMArgs a) {
    __shared__ LogTabEntry tab[16];
    __shared__ float queue[64 * W * E * R];
    __shared__ QueueCtl ctl;
    if (threadIdx.x == 0) ctl.bad = 0;
    load_log_table(tab);
    const int lane = threadIdx.x & (G - 1);
    const int group = threadIdx.x / G;
    constexpr int kGroups = (64 / G) * W, kRows = kGroups * R;
    static_assert(kRows <= 64, "QueueCtl holds 64 row sums");
    int turn = 0;
    const int n = *a.n_rows;
    const int K = KC > 0 ? KC : a.K;               // KC: compiled for this row length (launch_mm)
    for (int first = blockIdx.x * kRows; first < n; first += gridDim.x * kRows) {   // block-uniform trip count
        int row[R];
        bool active[R];
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int i = first + r * kGroups + group;
            row[r] = i < n ? a.rows[i] : 0;
            if constexpr (G == 64) row[r] = __builtin_amdgcn_readfirstlane(row[r]);   // one row per wavefront: a scalar (RowY::fetch_all)
            const bool running = i < n && !a.stop[row[r] / a.rows_per_batch];
            const int have = (kDead && running) ? a.cache_len[row[r]] : 0;
            active[r] = running && (!kDead || have == a.chunk);
            any = any || active[r];
            // hand the row on to the next chunk's list (an active row ends this chunk with chunk + 1 < n_checks pairs)
            if (kDead && a.next_rows && running && have < a.n_checks && lane == 0)
                a.next_rows[atomicAdd(a.next_count, 1)] = row[r];
        }
        if (!__syncthreads_or(any)) continue;                    // e.g. every batch of these rows has stopped
        const float* src = (kDead && a.chunk > 0) ? a.beta_dead : a.alpha;
        float* dst = kDead ? a.beta_dead : a.alpha;
        float beta[R][E];
        RowY<E, G> yv[R];
        double num[R], den[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            yv[r].load(kDead ? nullptr : a.y + (size_t)row[r] * K, lane, K);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int d = elem_of<E, G>(e, lane);
                beta[r][e] = (active[r] && d < K) ? src[(size_t)row[r] * K + d] : 0.0f;
            }
            num[r] = den[r] = 0.0;
        }
        for (int l = a.l0; l <= a.l1; l++)
            mm_iterate_block<E, W, R, G, KC>(beta, yv, K, lane, active, tab, queue, &ctl, turn++, a.has_check && l == a.l1, num, den);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (!active[r]) continue;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int d = elem_of<E, G>(e, lane);
                if (d < K) dst[(size_t)row[r] * K + d] = beta[r][e];
            }
            if (a.work_counter && lane == 0)
                atomicAdd(a.work_counter, (unsigned long long)K * (unsigned long long)(a.l1 - a.l0 + 1));
            if (a.has_check) {
                const double sn = group_sum_f64_g<G>(num[r]), sd = group_sum_f64_g<G>(den[r]);
                if (lane == 0) {
                    double* out = kDead ? a.cache + ((size_t)row[r] * a.n_checks + a.chunk) * 2 : a.rowpart + 2 * (size_t)row[r];
                    out[0] = sn;
                    out[1] = sd;
                    if (kDead) a.cache_len[row[r]] = a.chunk + 1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Class-split MM iteration (live rows, every outer iteration but the first).
// Once the first E-step has run, the parameters of a live row leave the neighbourhood of 1 (measured on the reference's
// alpha at K = 1000: at the start of outer iteration 2, 45 % have a+1 < 2.3, 25 % lie in [2.3, 10), 29 % are >= 10; after
// 20 iterations 8 / 19 / 73 %; K = 100: 35 / 12 / 53 %).  k_mm_live nevertheless runs every element through the nine masked
// steps of digamma's recurrence (no step is taken from 10 on) and through the small-argument lgamma (only used below 2.3):
// a quarter and a fifth of its phase C.  Here an element executes only what its value class needs.  Per iteration the
// wavefront sorts its elements into three dense queues in LDS,
//     A: a+1 < 2.3        recurrence (eight unmasked steps + one) + Sleef's polynomial lgamma, two entries per lane (packed)
//     B: 2.3 <= a+1 < 10  recurrence (eight masked steps) + Sleef's large-argument lgamma
//     C: a+1 >= 10        no recurrence; the large-argument lgamma without its argument shift (a+1 > 7)
// laid out [A | B | C] in two planes of one word per element: the argument, replaced by lgamma(a+1), and digamma(a+1).
// Dense 64-entry passes (class A: 128) evaluate BOTH special functions of the class; phase C picks the two words up
// and runs the update algebra.  The same functions are applied to the same arguments as in k_mm_live, so the results are
// identical bit for bit (tests/test_gpu_round3.py::test_class_split_kernel_is_invisible).
// One wavefront per block: the queues are private to the wavefront and nothing waits at a barrier for another
// wavefront's dense passes (measured against four wavefronts sharing block-wide queues: K = 100 380 -> 369 ms,
// K = 397 453 -> 439 ms, K = 1000 -1 %; k_mm_live, whose single queue is short, is better off sharing).
// The host uses this kernel from the second outer iteration on; in the first one every parameter starts at 1 (class A)
// and the queue traffic would be pure overhead (measured: +15 % there).
#ifndef TCLIP_SPLIT_FROM
#define TCLIP_SPLIT_FROM 1         // first outer iteration (0-based) that runs k_mm_split
#endif
#ifndef TCLIP_SPLIT_MAX_E
#define TCLIP_SPLIT_MAX_E 24       // two words of LDS per element: 8 KB per wavefront at 16 registers per lane (16 wavefronts per CU),
                                   // 12 KB at 24 (12 wavefronts per CU, which is what those kernels' registers allow anyway)
#endif
#ifndef TCLIP_SPLIT_Y_REGS_MAX_E
#define TCLIP_SPLIT_Y_REGS_MAX_E 8     // y of a row in registers up to this many registers per lane; longer rows fetch it at the start of
                                       // phase C (RowY::fetch_all).  Round 4 kept 16 registers of y across the dense passes: the compiler
                                       // spilled them and re-read a pair at a time inside phase C, each re-read followed by a wait for
                                       // everything in flight.  Same-box A/B (profiles/r05_ab_phase_c.txt): K = 1000 -1.1 %, few-shot
                                       // K = 1000 -1.2 %, K = 397 hard -0.9 % per engine call, bit-identical; VGPR spills 59 -> 23
#endif
#ifndef TCLIP_SPLIT_WAVES_SMALL
#define TCLIP_SPLIT_WAVES_SMALL 4  // wavefronts per SIMD requested for up to 8 registers per lane
#endif
#ifndef TCLIP_SPLIT_WAVES_MID
#define TCLIP_SPLIT_WAVES_MID 4    // wavefronts per SIMD requested for 9..16 registers per lane
#endif
#ifndef TCLIP_SPLIT_MIN_E
#define TCLIP_SPLIT_MIN_E 5        // shorter rows fill too little of a dense pass: measured with 16 lanes per row on 1000 tasks,
                                   // split against k_mm_live: K = 10 +42 %, 37 +6 %, 47 +7 %, 64 +1 %, 80 -4.5 %, 96 -5 %, 100 -11 %,
                                   // 128 -10 %, 196 -14 %, 256 -13 %, 300 -15 %, 512 -17 %, 1000 -17 %
#endif

// Sleef's large-argument lgamma for a dense pass of arguments above 7 (class C: >= 10)
__device__ __forceinline__ float lgamma_gt7_dense(float v) {
    bool sure;
    float r = lgamma_sleef_gt7_f64<true>(v, sure);
    if (__builtin_expect(__ballot(!sure) != 0ull, 0)) r = sure ? r : lgamma_sleef_ge23<true>(v);
    return r;
}

// phase C of the split iteration: entry `slot` of the wavefront's planes holds lgamma(a+1) and digamma(a+1)
// kTiny: some parameter of the wavefront's rows may be 1e-11 or less (see pk_mm_update_stage1_core)
// kMeasure: the stop test's iteration (a template parameter so that the other 49 of 50 iterations are ONE basic block per
// register-pair loop: with the flag tested per pair the scheduler could not overlap one pair's chain with the next one's)
template <int E, int G, bool kTiny, bool kMeasure>
__device__ __forceinline__ void split_apply_updates(float (&beta)[E], const RowY<E, G, TCLIP_SPLIT_Y_REGS_MAX_E>& yv, int K, int lane, float psi_s,
                                                    const float* my0, const float* my1, const uint32_t (&slot)[(E + 1) / 2],
                                                    double& num, double& den) {
    constexpr bool measure = kMeasure;
    // y of the row: from the registers of the whole chunk (short rows), or fetched here, all at once, for this phase alone -
    // 16 values per lane that live across the dense passes were 16 values the compiler spilled and re-read one pair at a
    // time, each re-read followed by a wait for EVERYTHING in flight, the next pair's square-root table entries included
    constexpr bool kYLocal = !RowY<E, G, TCLIP_SPLIT_Y_REGS_MAX_E>::kInRegs;
    float yl[kYLocal ? E : 1];
    if constexpr (kYLocal) yv.fetch_all(yl);
    auto y_of = [&](int e) { if constexpr (kYLocal) return yl[e]; else return yv.get(e); };
    const int n_full = full_registers<E, G>(K);
    auto finish = [&](int p, const PkUpdateStage& st) {
        const int e = 2 * p;
        const f2 a{beta[e], beta[e + 1]};
        const f2 nb = pk_mm_update_stage2(st);
        const bool full = e + 1 < n_full;
        const bool ok0 = full || elem_of<E, G>(e, lane) < K, ok1 = full || elem_of<E, G>(e + 1, lane) < K;
        if (measure) {
            const double d0 = (double)nb.x - (double)a.x, d1 = (double)nb.y - (double)a.y;
            if (ok0) { num += d0 * d0; den += (double)a.x * (double)a.x; }
            if (ok1) { num += d1 * d1; den += (double)a.y * (double)a.y; }
        }
        if (full) {
            beta[e] = nb.x;
            beta[e + 1] = nb.y;
        } else {
            beta[e] = ok0 ? nb.x : 0.0f;
            beta[e + 1] = ok1 ? nb.y : 0.0f;
        }
    };
    PkUpdateStage pending;
#pragma unroll
    for (int p = 0; p < E / 2; p++) {
        const int e = 2 * p;
        const f2 a{beta[e], beta[e + 1]};
        const int i0 = (int)(slot[p] & 0xffffu), i1 = (int)(slot[p] >> 16);
        const PkUpdateStage st = pk_mm_update_stage1_given<kTiny>(a, f2{y_of(e), y_of(e + 1)}, psi_s, my1[i0], my1[i1], my0[i0], my0[i1]);
        if (p > 0) finish(p - 1, pending);
        pending = st;
    }
    if (E / 2 > 0) finish(E / 2 - 1, pending);
    if (E & 1) {
        constexpr int e = E - 1;
        const float a = beta[e];
        const int i = (int)(slot[e >> 1] & 0xffffu);
        const float nb = mm_update_algebra(a, y_of(e), psi_s, my1[i], my0[i]);
        const bool ok = elem_of<E, G>(e, lane) < K;
        if (measure && ok) {
            const double df = (double)nb - (double)a;
            num += df * df;
            den += (double)a * (double)a;
        }
        beta[e] = ok ? nb : 0.0f;
    }
}

// Lanes of one wavefront hand data to each other through the wavefront's LDS planes at three places of an iteration
// (phase A's scatter -> the dense passes -> phase C's pick-up -> the next iteration's scatter).  The hardware executes a
// wavefront's LDS operations in order; this keeps the COMPILER from moving a may-alias access across a hand-off
// (wavefront-scope fences generate no instructions).
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Where every element of a wavefront's rows sits in the planes: the place of element e of this lane (16 bits each, two per
// word) and the sizes of the three queues.  A placement stays valid for as long as no element changes its class - from one
// MM iteration to the next that is the rule, not the exception (the parameters move by 1e-4 .. 1e-7 of their value) - so
// the kernel keeps it across the iterations of a chunk (round 5): an iteration scatters its arguments to the places of
// the last one (one add and one LDS store per element), the dense passes check the one class condition their own form
// depends on, and only a wavefront that met a misplaced entry sorts again (the two sweeps below, which every iteration
// ran until round 4: 18 of 169 lane-instructions per update, most of them half-rate compares, v_mbcnt and selects).
#ifndef TCLIP_SPLIT_LAZY
#define TCLIP_SPLIT_LAZY 1
#endif
#ifdef TCLIP_PHASE_CLOCK
// design studies only (scripts/gpu_phase_clock.py): wavefront clocks (s_memtime) spent in the parts of the split iteration,
// summed over all wavefronts: [0] head (row sum, domain test, digamma of the row sum), [1] scatter or sort, [2] class A passes,
// [3] class B passes, [4] class C passes, [5] phase C, [6] iterations, [7] passes abandoned (misplaced entry)
__device__ unsigned long long g_phase_clock[8];
#define TCLIP_CLK(i, t) do { const long long now_ = __builtin_readcyclecounter(); pl.clk[i] += now_ - (t); (t) = now_; } while (0)
#else
#define TCLIP_CLK(i, t) do { } while (0)
#endif
template <int E>
struct SplitPlacement {
    uint32_t slot[(E + 1) / 2];
    int nA, nB, nC;
    bool valid;
    int sorts;          // full placements of this wavefront in the launch (instrumentation)
#ifdef TCLIP_PHASE_CLOCK
    long long clk[8];
#endif
};

// The full placement: two sweeps over the registers.
// First sweep: the sizes of the three classes (one v_cmp per threshold writes the 64-lane mask, the counting runs on the
// scalar unit).  Only the last registers of a lane can hold slots beyond the row (the kernel is instantiated for the
// smallest E that covers K: at most three registers of slack, the fourth for the upper half of the 64-lane layout); their
// slots are masked out of the queues, all other registers are queued whole.
// (Lane groups without a row hold zeros and travel as class A; they only occur in the last block of a list.)
// Second sweep: every element's argument to its place in [A | B | C]; the place is kept (16 bits) for the pick-up.
template <int E, int G, int KC>
__device__ __forceinline__ void split_place(const float (&beta)[E], int K, int lane, float* my0, SplitPlacement<E>& pl) {
    constexpr int kFirstRagged = first_ragged_register<E, G, KC>();
    int nA = 0, nC = 0, nV = kFirstRagged * 64;
#pragma unroll
    for (int e = 0; e < E; e++) {
        float x1 = beta[e] + 1.0f;
        asm volatile("" : "+v"(x1));         // (also keeps the compares inside the rarely taken sort: hoisted out of the caller's loop, all 2 E of
                                             // them ran every iteration and their masks went through v_writelane)
        unsigned long long mA = __builtin_amdgcn_ballot_w64(x1 < 2.3f), mC = __builtin_amdgcn_ballot_w64(x1 >= 10.0f);
        if (e >= kFirstRagged) {
            const bool in_row = elem_of<E, G>(e, lane) < K;
            const unsigned long long mv = __builtin_amdgcn_ballot_w64(in_row);
            mA &= mv;
            mC &= mv;
            nV += __popcll(mv);
        }
        nA += __popcll(mA);
        nC += __popcll(mC);
    }
    const int nB = nV - nA - nC;                                    // NaN compares false twice: class B, in both sweeps
    // The compares are repeated on purpose (the asm keeps the compiler from holding 2 E masks in scalar registers).
    int cA = 0, cB = nA, cC = nA + nB;
#pragma unroll
    for (int e = 0; e < E; e++) {
        float x1 = beta[e] + 1.0f;
        asm volatile("" : "+v"(x1));
        const bool lt = x1 < 2.3f, ge = x1 >= 10.0f;
        unsigned long long mA = __builtin_amdgcn_ballot_w64(lt), mC = __builtin_amdgcn_ballot_w64(ge);
        unsigned long long mB = ~(mA | mC);
        bool in_row = true;
        if (e >= kFirstRagged) {
            in_row = elem_of<E, G>(e, lane) < K;
            const unsigned long long mv = __builtin_amdgcn_ballot_w64(in_row);
            mA &= mv;
            mC &= mv;
            mB &= mv;
        }
        // rank + running base in the two v_mbcnt of each class (their addend operand carries the base)
        const int iA = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mA >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA, (unsigned)cA));
        const int iB = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mB >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB, (unsigned)cB));
        const int iC = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mC >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mC, (unsigned)cC));
        int idx = lt ? iA : iB;
        idx = ge ? iC : idx;
        if (e >= kFirstRagged) {
            idx = in_row ? idx : 0;                                 // a slot beyond the row reads entry 0; its result is discarded
            if (in_row) my0[idx] = x1;
        } else {
            my0[idx] = x1;
        }
        if (e & 1) pl.slot[e >> 1] |= (uint32_t)idx << 16;
        else pl.slot[e >> 1] = (uint32_t)idx;
        cA += __popcll(mA);
        cB += __popcll(mB);
        cC += __popcll(mC);
    }
    pl.nA = nA;
    pl.nB = nB;
    pl.nC = nC;
}

// The arguments of this iteration to the places of the last placement.
template <int E, int G, int KC>
__device__ __forceinline__ void split_scatter(const float (&beta)[E], int K, int lane, float* my0, const SplitPlacement<E>& pl) {
    constexpr int kFirstRagged = first_ragged_register<E, G, KC>();
#pragma unroll
    for (int e = 0; e < E; e++) {
        const float x1 = beta[e] + 1.0f;
        const int idx = (e & 1) ? (int)(pl.slot[e >> 1] >> 16) : (int)(pl.slot[e >> 1] & 0xffffu);
        if (e >= kFirstRagged) {
            if (elem_of<E, G>(e, lane) < K) my0[idx] = x1;
        } else {
            my0[idx] = x1;
        }
    }
}

// my0 / my1: the wavefront's two planes of 64 E words
template <int E, int G, int KC = 0>
__device__ __forceinline__ void mm_iterate_wave_split(float (&beta)[E], const RowY<E, G, TCLIP_SPLIT_Y_REGS_MAX_E>& yv, int K, int lane, bool active,
                                                      const LogTabEntry* tab, float* my0, float* my1, bool measure,
                                                      double& num, double& den, SplitPlacement<E>& pl, bool keep_placement) {
    const int lane64 = threadIdx.x & 63;
#ifdef TCLIP_PHASE_CLOCK
    long long tclk = __builtin_readcyclecounter();
    pl.clk[6] += 1;
#endif
    float s = 16.0f;
    bool in_domain = true;
    if (active) {
        s = row_sum_torch<E, G, KC>(beta, K, lane);
        in_domain = lane != 0 || (fast_range_f32(s) && s <= 0x1p40f);     // the sum lives in lane 0
        uint32_t largest = 0u;
#pragma unroll
        for (int e = 0; e < E; e++) largest = max(largest, f32_bits(beta[e]));
        in_domain = in_domain & mm_fast_domain_of_max_bits(largest);
    }
    s = __shfl(s, 0, G);                                            // the row's sum in every lane of its group
    if (__builtin_expect(!__all(in_domain), 0)) {                   // NaN / inf / out of range somewhere in the wavefront
        if (active) {
            const float psi_s = digamma_f32(s);
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float nb = mm_update_generic(beta[e], yv.get(e), psi_s);
                const bool ok = elem_of<E, G>(e, lane) < K;
                if (measure && ok) {
                    const double df = (double)nb - (double)beta[e];
                    num += df * df;
                    den += (double)beta[e] * (double)beta[e];
                }
                beta[e] = ok ? nb : 0.0f;
            }
        }
        pl.valid = false;                                           // the generic path keeps no placement
        return;
    }
    // Is any parameter 1e-11 or less (the curvature's constant branch, em_dirichlet.py:155)?  Phase C of a wavefront without
    // one runs without that compare and select.  A running minimum over the parameters' bit patterns as signed integers (the
    // order of the non-negative floats; -0 sorts below everything and counts as small), three values per instruction and no
    // lane mask to keep - sixteen ballots lived in scalar registers the kernel does not have and went through v_writelane /
    // v_readlane.  Slots beyond the row hold 0 and do not take part.
    constexpr int kFirstRagged = first_ragged_register<E, G, KC>();
    int32_t smallest = 0x7f800000;
#pragma unroll
    for (int e = 0; e < E; e++) {
        int32_t bits = (int32_t)f32_bits(beta[e]);
        if (e >= kFirstRagged) bits = elem_of<E, G>(e, lane) < K ? bits : 0x7f800000;
        smallest = bits < smallest ? bits : smallest;
    }
    const bool tiny = __builtin_amdgcn_ballot_w64(smallest <= (int32_t)0x2d2febffu) != 0ull;      // 0x2d2febff = 1e-11f
    const float psi_s = digamma_pos_f32(s, tab);                    // the row sums of the wavefront's rows, one evaluation
    // phase A + phase B, until the dense passes have met every entry in a queue whose form is the one for its value
    bool sort_now = !(TCLIP_SPLIT_LAZY && keep_placement && pl.valid);
    TCLIP_CLK(0, tclk);
    for (;;) {
        if (sort_now) {
            split_place<E, G, KC>(beta, K, lane, my0, pl);
            pl.sorts++;
        } else split_scatter<E, G, KC>(beta, K, lane, my0, pl);
        const int nA = pl.nA, nB = pl.nB, nC = pl.nC;
        wave_lds_handoff();
        TCLIP_CLK(1, tclk);
        // phase B: the queues in dense passes; entry i leaves with lgamma(a+1) in plane 0 and digamma(a+1) in plane 1.
        // A lane beyond the end of its queue in a last, partial pass takes the queue's LAST entry along with that entry's own lane:
        // same argument, same results, written to the same two words - so no pass needs an execution mask around its LDS
        // read or its stores (two s_and_saveexec / s_or pairs, a compare and a default value per pass before).
        // Each pass first checks the condition its form depends on - A: x < 2.3 (the polynomial lgamma, eight unmasked
        // recurrence steps), B: x >= 2.3 (its masked recurrence and general lgamma are right from 10 on as well: an entry
        // that has grown past 10 stays until the next sort), C: x >= 10 (no recurrence) - and leaves at once when an entry
        // placed by an earlier iteration has left its class.
        bool misplaced = false;
        int jA = 0;
        for (; jA + 64 < nA; jA += 128) {                           // more than 64 entries left: two per lane on the packed pipe
            const int j = jA;
            const int i0 = j + lane64, i1 = min(i0 + 64, nA - 1);       // i0 < nA: the loop's condition
            const f2 x{my0[i0], my0[i1]};
            if (!sort_now && __builtin_amdgcn_ballot_w64(!(fmaxf(x.x, x.y) < 2.3f)) != 0ull) { misplaced = true; break; }
            f2 xr = x, acc = pk(0.0f);
#pragma unroll
            for (int k = 0; k < 8; k++) {                           // x + 7 < 10: the first eight steps are always taken
                acc = acc - pk_rcp_rn(xr);
                xr = xr + pk(1.0f);
            }
            const f2 m{below10_f32(xr.x), below10_f32(xr.y)};
            acc = pk_fma(-m, pk_rcp_rn(xr), acc);
            xr = xr + m;
            const f2 psi = pk_digamma_after_rec(xr, acc, tab);
            const f2 lg = pk_lgamma_sleef_1_23(x);
            my0[i0] = lg.x;
            my1[i0] = psi.x;
            my0[i1] = lg.y;
            my1[i1] = psi.y;
        }
        for (; !misplaced && jA < nA; jA += 64) {                   // a last pass of up to 64 entries: one per lane (half the instructions)
            const int i = min(jA + lane64, nA - 1);
            const float x = my0[i];
            if (!sort_now && __builtin_amdgcn_ballot_w64(!(x < 2.3f)) != 0ull) { misplaced = true; break; }
            float xr = x, acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                acc = acc - rcp_rn_f32(xr);
                xr = xr + 1.0f;
            }
            const float m = below10_f32(xr);
            acc = __builtin_fmaf(-m, rcp_rn_f32(xr), acc);
            xr = xr + m;
            const float psi = digamma_after_rec_ge10<false>(xr, acc, tab);
            bool sure;
            float lg = lgamma_sleef_1_23_f64(x, sure);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!sure) != 0ull, 0)) lg = sure ? lg : lgamma_sleef_05_23(x);
            my0[i] = lg;
            my1[i] = psi;
        }
        TCLIP_CLK(2, tclk);
        // (two entries per lane in these passes - two independent chains for the scheduler to interleave - measured no
        // different: K = 100 361 against 359 ms, K = 1000 equal; the passes are not latency-bound)
        auto pass_b = [&](int i) -> bool {                          // recurrence (eight masked steps: x + 8 >= 10) + series + general large-argument lgamma
            const float x = my0[i];
            if (!sort_now && __builtin_amdgcn_ballot_w64(!(x >= 2.3f)) != 0ull) return true;
            float xr = x, acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float m = below10_f32(xr);
                acc = __builtin_fmaf(-m, rcp_rn_f32(xr), acc);
                xr += m;
            }
            const float psi = digamma_after_rec_ge10<false>(xr, acc, tab);
            const float lg = lgamma_big_dense(x);
            my0[i] = lg;
            my1[i] = psi;
            return false;
        };
        // When the last, partial passes of B and C fit into one (<= 64 entries together), class C's leftovers ride in B's pass:
        // the masked recurrence takes no step from 10 on and the general lgamma returns what the no-shift form returns (both are
        // RN32 of Sleef's value), so the entries get the same bits for the price of one pass instead of two.
        const int tB = nB & 63, tC = nC & 63;
        const bool merged = tB > 0 && tC > 0 && tB + tC <= 64;
        const int endB = merged ? nB - tB : nB, endC = merged ? nC - tC : nC;
        for (int j = 0; !misplaced && j < endB; j += 64) misplaced = pass_b(nA + min(j + lane64, endB - 1));
        if (merged && !misplaced) {
            const int l = min(lane64, tB + tC - 1);
            misplaced = pass_b(l < tB ? nA + endB + l : nA + nB + endC + (l - tB));
        }
        TCLIP_CLK(3, tclk);
        for (int j = 0; !misplaced && j < endC; j += 64) {
            const int i = nA + nB + min(j + lane64, endC - 1);
            const float x = my0[i];
            if (!sort_now && __builtin_amdgcn_ballot_w64(!(x >= 10.0f)) != 0ull) { misplaced = true; break; }
            const float psi = digamma_after_rec_ge10<true>(x, 0.0f, tab);
            const float lg = lgamma_gt7_dense(x);
            my0[i] = lg;
            my1[i] = psi;
        }
        TCLIP_CLK(4, tclk);
        if (!misplaced) break;
#ifdef TCLIP_PHASE_CLOCK
        pl.clk[7] += 1;
#endif
        sort_now = true;                                            // (the passes of a fresh placement check nothing: this loop runs twice at most)
        wave_lds_handoff();
    }
    pl.valid = true;
    // phase C
    wave_lds_handoff();
    if (__builtin_expect(measure, 0)) {
        if (tiny) split_apply_updates<E, G, true, true>(beta, yv, K, lane, psi_s, my0, my1, pl.slot, num, den);
        else split_apply_updates<E, G, false, true>(beta, yv, K, lane, psi_s, my0, my1, pl.slot, num, den);
    } else if (__builtin_expect(tiny, 0)) split_apply_updates<E, G, true, false>(beta, yv, K, lane, psi_s, my0, my1, pl.slot, num, den);
    else split_apply_updates<E, G, false, false>(beta, yv, K, lane, psi_s, my0, my1, pl.slot, num, den);
    wave_lds_handoff();
    TCLIP_CLK(5, tclk);
}

template <int E, int G, int KC = 0>
__global__ __launch_bounds__(64, (E > 16 ? TCLIP_MM_WAVES_LARGE : (E > 8 ? TCLIP_SPLIT_WAVES_MID : TCLIP_SPLIT_WAVES_SMALL))) void k_mm_split(MMArgs a) {
    static_assert(E <= TCLIP_SPLIT_MAX_E, "LDS: two words per element");
    __shared__ LogTabEntry tab[16];
    __shared__ float plane0[64 * E];
    __shared__ float plane1[64 * E];
    load_log_table(tab);
    const int lane = threadIdx.x & (G - 1);
    const int group = threadIdx.x / G;
    constexpr int kRows = 64 / G;
    const int n = *a.n_rows;
    const int K = KC > 0 ? KC : a.K;
    for (int first = blockIdx.x * kRows; first < n; first += gridDim.x * kRows) {
        const int i = first + group;
        int row = i < n ? a.rows[i] : 0;
        if constexpr (G == 64) row = __builtin_amdgcn_readfirstlane(row);     // one row per wavefront: its y pointer is a scalar register pair
        const bool active = i < n && !a.stop[row / a.rows_per_batch];
        if (!__any(active)) continue;
        float beta[E];
        RowY<E, G, TCLIP_SPLIT_Y_REGS_MAX_E> yv;
        double num = 0.0, den = 0.0;
        yv.load(a.y + (size_t)row * K, lane, K);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = elem_of<E, G>(e, lane);
            beta[e] = (active && d < K) ? a.alpha[(size_t)row * K + d] : 0.0f;
        }
        SplitPlacement<E> pl;
        pl.valid = false;
        pl.sorts = 0;
#ifdef TCLIP_PHASE_CLOCK
        for (int i = 0; i < 8; i++) pl.clk[i] = 0;
#endif
        for (int l = a.l0; l <= a.l1; l++)
            mm_iterate_wave_split<E, G, KC>(beta, yv, K, lane, active, tab, plane0, plane1, a.has_check && l == a.l1, num, den, pl,
                                            a.keep_placement != 0);
        // per-wavefront instrumentation by lane 0 whether or not ITS lane group has a row (in the 16- and 32-lane layouts the
        // wavefront iterates as long as any of its groups does: __any(active) above)
#ifdef TCLIP_PHASE_CLOCK
        if ((threadIdx.x & 63) == 0)
            for (int i = 0; i < 8; i++) atomicAdd(&g_phase_clock[i], (unsigned long long)pl.clk[i]);
#endif
        if (a.work_counter && (threadIdx.x & 63) == 0) {           // [1]: wavefront-iterations of this kernel, [2]: full placements among them
            atomicAdd(a.work_counter + 1, (unsigned long long)(a.l1 - a.l0 + 1));
            atomicAdd(a.work_counter + 2, (unsigned long long)pl.sorts);
        }
        if (!active) continue;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = elem_of<E, G>(e, lane);
            if (d < K) a.alpha[(size_t)row * K + d] = beta[e];
        }
        if (a.work_counter && lane == 0)
            atomicAdd(a.work_counter, (unsigned long long)K * (unsigned long long)(a.l1 - a.l0 + 1));
        if (a.has_check) {
            const double sn = group_sum_f64_g<G>(num), sd = group_sum_f64_g<G>(den);
            if (lane == 0) {
                a.rowpart[2 * (size_t)row] = sn;
                a.rowpart[2 * (size_t)row + 1] = sd;
            }
        }
    }
}

#ifdef TCLIP_ISA_ONLY
// ISA studies (scripts/isa_one.sh): only the K = 1000 MM kernels are instantiated - seconds instead of two minutes per compile.
// Never part of a build of the library.
template __global__ void k_mm_split<16, 64, 1000>(MMArgs);
template __global__ void k_mm_live<16, 4, false, 1, 64, 1000>(MMArgs);
#ifdef TCLIP_ISA_K100
template __global__ void k_mm_split<7, 16, 100>(MMArgs);
template __global__ void k_mm_split<13, 32, 397>(MMArgs);
#endif
}  // namespace tclip
#else
// Batch-global stop test (em_dirichlet.py:169-175), one block per batch:
//   crit = ||b'-b||_F^2 / ||b||_F^2 over all N*K*K entries of the batch;  stop if < 1e-11.
// fp64 accumulation in a fixed order; the final arithmetic follows the reference's fp32 form
// norm()**2 / norm()**2.  Also records the number of MM iterations executed.
// First stage for large batches (K = 1000: 125 000 rows per batch, 2 MB of partials - one block
// took 0.3 ms per checkpoint, on the critical path of every chunk): slice s of n_slices sums its
// rows in a fixed order into dpart[b][s]; k_mm_decide then sums the slices in order.
__global__ __launch_bounds__(256) void k_mm_decide_partial(const double* __restrict__ rowpart, const double* __restrict__ cache,
                                                           const uint8_t* __restrict__ live, int rows_per_batch,
                                                           int n_checks, int chunk, const int32_t* __restrict__ stop,
                                                           double* __restrict__ dpart) {
    const int b = blockIdx.y, slice = blockIdx.x, n_slices = gridDim.x;
    if (stop[b]) return;
    const int per = (rows_per_batch + n_slices - 1) / n_slices;
    const int r0 = slice * per, r1 = r0 + per < rows_per_batch ? r0 + per : rows_per_batch;
    const size_t base = (size_t)b * rows_per_batch;
    double num = 0.0, den = 0.0;
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const size_t row = base + r;
        if (live[row]) {
            num += rowpart[2 * row];
            den += rowpart[2 * row + 1];
        } else {
            num += cache[(row * n_checks + chunk) * 2];
            den += cache[(row * n_checks + chunk) * 2 + 1];
        }
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = num;
    sh[1][threadIdx.x] = den;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dpart[((size_t)b * n_slices + slice) * 2] = sh[0][0];
        dpart[((size_t)b * n_slices + slice) * 2 + 1] = sh[1][0];
    }
}

__global__ __launch_bounds__(1024) void k_mm_decide(const double* __restrict__ rowpart, const double* __restrict__ cache,
                                                    const uint8_t* __restrict__ live, int rows_per_batch, int n_checks,
                                                    int chunk, int has_check, int l1, int is_last, int iter_mm,
                                                    int32_t* __restrict__ stop, int32_t* __restrict__ mm_iters_out /* [B] slot of this outer iteration, stride given */,
                                                    int mm_stride, const double* __restrict__ dpart, int n_slices) {
    const int b = blockIdx.x;
    if (stop[b]) return;
    __shared__ double sh[2][1024];
    double num = 0.0, den = 0.0;
    if (has_check && dpart) {
        if ((int)threadIdx.x < n_slices) {
            num = dpart[((size_t)b * n_slices + threadIdx.x) * 2];
            den = dpart[((size_t)b * n_slices + threadIdx.x) * 2 + 1];
        }
    } else if (has_check) {
        const size_t base = (size_t)b * rows_per_batch;
        for (int r = threadIdx.x; r < rows_per_batch; r += blockDim.x) {
            const size_t row = base + r;
            if (live[row]) {
                num += rowpart[2 * row];
                den += rowpart[2 * row + 1];
            } else {
                num += cache[(row * n_checks + chunk) * 2];
                den += cache[(row * n_checks + chunk) * 2 + 1];
            }
        }
    }
    sh[0][threadIdx.x] = num;
    sh[1][threadIdx.x] = den;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bool stopped = false;
        if (has_check) {
            const float nn = (float)__builtin_sqrt(sh[0][0]), dn = (float)__builtin_sqrt(sh[1][0]);
            const float crit = (nn * nn) / (dn * dn);
            if (crit < 1e-11f) {
                stop[b] = 1;
                mm_iters_out[(size_t)b * mm_stride] = l1 + 1;
                stopped = true;
            }
        }
        if (!stopped && is_last) mm_iters_out[(size_t)b * mm_stride] = iter_mm;
    }
}

// ------------------------------------------------------------------------------------------
// E-step, part 1: per row  lgamma(sum_d alpha) + (-sum_d lgamma(alpha_d))     (em_dirichlet.py:35-36)
template <int E>
__global__ __launch_bounds__(256) void k_row_consts(const float* __restrict__ alpha, const int32_t* __restrict__ rows,
                                                    const int32_t* __restrict__ n_rows, int K,
                                                    float* __restrict__ rowc) {
    const int lane = threadIdx.x & (kGroup - 1);
    const int groups_per_block = blockDim.x / kGroup;
    const int n = *n_rows;
    for (int i = blockIdx.x * groups_per_block + threadIdx.x / kGroup; i < n; i += gridDim.x * groups_per_block) {
        const int row = rows[i];
        float av[E], lg[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = e * kGroup + lane;
            const bool ok = d < K;
            av[e] = ok ? alpha[(size_t)row * K + d] : 0.0f;
            lg[e] = ok ? lgamma_f32(av[e]) : 0.0f;
        }
        const float l1 = lgamma_f32(group_sum_torch<E>(av, K, lane));
        const float l2 = -group_sum_torch<E>(lg, K, lane);
        if (lane == 0) rowc[row] = l1 + l2;
    }
}

// E-step, part 2: logit0[t,q,k] = rowc[t,k] + sum_d (alpha[t,k,d]-1) * logz[t,q,d]   (em_dirichlet.py:37-39)
// A block takes R consecutive entries of the row list (neighbours in the list are rows of one task almost always:
// k_build_rows keeps 64-row spans together) with their alpha in registers; its 8 lane groups sweep the task's queries,
// each log z row loaded once for the R rows.  Blocks b, b + 8, b + 16, ... (one XCD under round-robin dispatch) walk ONE
// contiguous eighth of the list, so that a task's 75 K log z values are pulled into one L2 instead of eight.
// Registers below index logits_full_regs(E) lie inside the row for every K that dispatch_E hands to the instantiation
// (K > 32 x the next smaller size): their loads, products and additions carry no bounds tests.
constexpr int logits_full_regs(int E) {
    constexpr int sizes[] = {0, 1, 2, 3, 4, 6, 8, 10, 13, 16, 20, 24, 28, 32};      // dispatch_E's
    int prev = 0;
    for (int s : sizes) {
        if (s >= E) break;
        prev = s;
    }
    return prev;
}
template <int E, int R>
__global__ __launch_bounds__(256) void k_logits(const float* __restrict__ alpha, const float* __restrict__ logz,
                                                const float* __restrict__ rowc, const int32_t* __restrict__ rows,
                                                const int32_t* __restrict__ n_rows, int Q, int K,
                                                float* __restrict__ logit0, const int32_t* __restrict__ only_if) {
    if (only_if && *only_if == 0) return;           // the initial call: k_init_logits has done the work
    constexpr int kFull = logits_full_regs(E);
    constexpr int kGroups = 256 / kGroup;
    const int lane = threadIdx.x & (kGroup - 1);
    const int group = threadIdx.x / kGroup;
    const int n = *n_rows;
    const int nseg = (n + R - 1) / R, per_xcd = (nseg + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    for (int j = blockIdx.x >> 3; j < per_xcd; j += gridDim.x >> 3) {              // the grid is a multiple of 8 blocks
        const int seg = xcd * per_xcd + j;
        if (seg >= nseg) break;
        int row[R], task[R];
        bool same = true;
        float am1[R][E], rc[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            // (a segment past the end of the list repeats its first row: same value to the same place)
            row[r] = __builtin_amdgcn_readfirstlane(rows[seg * R + r < n ? seg * R + r : seg * R]);
            task[r] = row[r] / K;
            same = same && task[r] == task[0];
            const float* ar = alpha + (size_t)row[r] * K + lane;
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (e < kFull) am1[r][e] = ar[e * kGroup] - 1.0f;
                else am1[r][e] = e * kGroup + lane < K ? ar[e * kGroup] - 1.0f : 0.0f;
            }
            rc[r] = rowc[row[r]];
        }
        auto sweep = [&](int t, auto&& use) {        // the task's queries: log z row q in registers, handed to use(q, lv)
            for (int q = group; q < Q; q += kGroups) {
                const float* lz = logz + ((size_t)t * Q + q) * K + lane;
                float lv[E];
#pragma unroll
                for (int e = 0; e < E; e++) {
                    if (e < kFull) lv[e] = lz[e * kGroup];
                    else lv[e] = lz[e * kGroup + lane < K ? e * kGroup : K - 1 - lane];   // clamped into the row: the load is unconditional
                }
                use(q, lv);
            }
        };
        auto one = [&](int r, int t, int q, const float (&lv)[E]) {
            float pr[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                pr[e] = am1[r][e] * lv[e];
                if (e >= kFull) pr[e] = e * kGroup + lane < K ? pr[e] : 0.0f;
            }
            const float l3 = group_sum_torch<E, true, kFull>(pr, K, lane);
            if (lane == 0) logit0[((size_t)t * Q + q) * K + (row[r] - t * K)] = rc[r] + l3;
        };
        if (same) {
            sweep(task[0], [&](int q, const float (&lv)[E]) {
#pragma unroll
                for (int r = 0; r < R; r++) one(r, task[0], q, lv);
            });
        } else {                                     // the segment straddles two tasks
#pragma unroll
            for (int r = 0; r < R; r++) sweep(task[r], [&](int q, const float (&lv)[E]) { one(r, task[r], q, lv); });
        }
    }
}

// k-means E-step: logit[t,q,k] = temperature * (pre * sum_d (w[t,k,d] - z[t,q,d])^2), the sum in torch's
// last-dim order; SOFT_KMEANS (soft_kmeans.py:105-125): pre = -1/2, temperature = T; HARD_KMEANS
// (hard_kmeans.py:26-35): pre = temperature = 1, i.e. the plain squared distance.
// A block handles kRowsPerBlock consecutive classes of one task: a query row is loaded once and used
// against all of them (the task's feature block would otherwise be re-read from L2 once per class;
// four classes per block made SOFT_KMEANS at K = 397 2.4x faster).  `need` marks the (task, class)
// rows to (re)compute.
template <int E, int kRowsPerBlock>
__global__ __launch_bounds__(256) void k_kmeans_logits_rows(const float* __restrict__ w, const float* __restrict__ z,
                                                            const uint8_t* __restrict__ need, int Q, int K, float pre,
                                                            float temperature, float* __restrict__ logit0) {
    const int lane = threadIdx.x & (kGroup - 1);
    const int group = threadIdx.x / kGroup, groups_per_block = blockDim.x / kGroup;
    const int t = blockIdx.y, k0 = blockIdx.x * kRowsPerBlock;
    bool want[kRowsPerBlock];
    bool any = false;
    float wv[kRowsPerBlock][E];
#pragma unroll
    for (int j = 0; j < kRowsPerBlock; j++) {
        want[j] = k0 + j < K && need[(size_t)t * K + k0 + j];
        any = any || want[j];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = e * kGroup + lane;
            wv[j][e] = (want[j] && d < K) ? w[((size_t)t * K + k0 + j) * K + d] : 0.0f;
        }
    }
    if (!any) return;
    for (int q = group; q < Q; q += groups_per_block) {
        const float* zq = z + ((size_t)t * Q + q) * K;
        float zv[E];
#pragma unroll
        for (int e = 0; e < E; e++) {                                   // unconditional loads (index clamped into the row): one batch in flight
            const int d = e * kGroup + lane;
            zv[e] = zq[d < K ? d : K - 1];
        }
#pragma unroll
        for (int e = 0; e < E; e++) zv[e] = e * kGroup + lane < K ? zv[e] : 0.0f;
#pragma unroll
        for (int j = 0; j < kRowsPerBlock; j++) {
            if (!want[j]) continue;                                  // block-uniform
            float pr[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float df = e * kGroup + lane < K ? wv[j][e] - zv[e] : 0.0f;
                pr[e] = df * df;
            }
            const float ssum = group_sum_torch<E>(pr, K, lane);
            if (lane == 0) logit0[((size_t)t * Q + q) * K + k0 + j] = temperature * (pre * ssum);
        }
    }
}

// The same logits with the roles turned round (round 4): ONE LANE per class, the whole sum over d inside the lane.
// k_kmeans_logits_rows spreads a (query, class) pair over 32 lanes and pays ~25 cross-lane instructions (shuffles through
// the LDS crossbar, DPP adds) beside the 39 arithmetic ones for every 397-element sum, reads each query row once per four
// classes from L2, and stores one float per 32 lanes.  Here a block stages a tile of 64 centroids in LDS once (odd row
// stride: lane k reads word k stride + d, 32 lanes on 32 banks), lane k of every wavefront keeps torch's 32 partial sums
// (accumulator r = 0..3, vector lane j = 0..7: element d = 32 m + 8 r + j belongs to slot 8 r + j at step m) in registers,
// the wavefront's query row is wave-uniform and arrives through the scalar cache as the SGPR operand of the subtraction:
// three VALU instructions and one ds_read per (query, class, d), nothing across lanes, coalesced stores.  Same operations
// in the same order as group_sum_torch (slots sequentially over the steps, whole vectors beyond the 4-way part into
// accumulator 0, a0 + a1 + a2 + a3 per vector lane, the K mod 8 tail first, then the eight vector lanes), hence the same
// bits: tests/test_gpu_round4.py::test_kmeans_tile_kernel_is_invisible runs both kernels on 20 row lengths.
// Rows of 32 .. 511 elements (fewer than 16 steps: torch's cascade never dumps; 64 x 511 floats of LDS = 131 KB).
constexpr int kKmeansTile = 64;
#ifndef TCLIP_KMEANS_TILE_THREADS
#define TCLIP_KMEANS_TILE_THREADS 1024    // the tile's LDS (101 KB at K = 397) allows one block per CU: sixteen wavefronts share it, four per SIMD (512: 74 against 66 ms per 1000-task SOFT_KMEANS call)
#endif
constexpr int kKmeansTileThreads = TCLIP_KMEANS_TILE_THREADS;
#ifndef TCLIP_KMEANS_PREFETCH
#define TCLIP_KMEANS_PREFETCH 1        // 1000 tasks, K = 397, 20 iterations of SOFT_KMEANS: 66.5 ms against 73.7 without
#endif
__global__ __launch_bounds__(TCLIP_KMEANS_TILE_THREADS) void k_kmeans_logits_tile(const float* __restrict__ w, const float* __restrict__ z,
                                                            const uint8_t* __restrict__ need, int Q, int K, int stride, float pre,
                                                            float temperature, float* __restrict__ logit0) {
    extern __shared__ float wt[];                                   // [kKmeansTile][stride]
    const int t = blockIdx.y, k0 = blockIdx.x * kKmeansTile;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
    const int k = k0 + lane;
    const bool ok = k < K && need[(size_t)t * K + k];
    if (!__syncthreads_or(ok)) return;                              // no class of the tile moved
    const int rows = K - k0 < kKmeansTile ? K - k0 : kKmeansTile;
    const float* wsrc = w + ((size_t)t * K + k0) * K;
    {   // eight loads in flight per thread (a loop of load - wait - store took as long as the tile's arithmetic: a block
        // is alone on its CU, nothing else hides the latency); word i of the tile goes to row i / K, the quotient by steps
        const int n = rows * K, step = blockDim.x;
        const int pad = stride - K;                                 // 0 or 1 words per row
        for (int i0 = threadIdx.x; i0 < n; i0 += 8 * step) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = i0 + j * step < n ? wsrc[i0 + j * step] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j * step;
                if (i < n) wt[i + (pad ? i / K : 0)] = v[j];
            }
        }
    }
    __syncthreads();
    const float* wl = wt + (lane < rows ? lane : rows - 1) * stride;     // lanes beyond the last class recompute it; nothing is stored
    const int vec_size = K >> 3, size_ilp = vec_size >> 2, nleft = vec_size - 4 * size_ilp, ntail = K - 8 * vec_size;
    for (int q = wave; q < Q; q += n_waves) {
        const float* zq = z + ((size_t)t * Q + q) * K;              // wave-uniform: scalar loads
        float acc[32];
#pragma unroll
        for (int sl = 0; sl < 32; sl++) acc[sl] = 0.0f;
#if TCLIP_KMEANS_PREFETCH
        // the query row's next 32 values are requested (scalar loads, ~300 cycles from L2) before the current step's
        // arithmetic, not at their first use: with one block per CU there are too few wavefronts to hide that latency
        float zc[32];
#pragma unroll
        for (int sl = 0; sl < 32; sl++) zc[sl] = zq[sl];
        for (int m = 0; m < size_ilp; m++) {
            float zn[32];
            const int nx = m + 1 < size_ilp ? 32 * (m + 1) : 32 * m;   // the last step re-reads its own values (never used)
#pragma unroll
            for (int sl = 0; sl < 32; sl++) zn[sl] = zq[nx + sl];
#pragma unroll
            for (int sl = 0; sl < 32; sl++) {
                const float df = wl[32 * m + sl] - zc[sl];
                acc[sl] += df * df;
            }
#pragma unroll
            for (int sl = 0; sl < 32; sl++) zc[sl] = zn[sl];
        }
#else
        for (int m = 0; m < size_ilp; m++) {
#pragma unroll
            for (int sl = 0; sl < 32; sl++) {
                const float df = wl[32 * m + sl] - zq[32 * m + sl];
                acc[sl] += df * df;
            }
        }
#endif
        int d = 32 * size_ilp;
        for (int i = 0; i < nleft; i++, d += 8) {                   // whole vectors beyond the 4-way part join accumulator 0
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float df = wl[d + j] - zq[d + j];
                acc[j] += df * df;
            }
        }
        float fin = 0.0f;
        for (int i = 0; i < ntail; i++) {                           // the K mod 8 tail first
            const float df = wl[d + i] - zq[d + i];
            fin += df * df;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float p0 = acc[j];
            p0 += acc[8 + j];
            p0 += acc[16 + j];
            p0 += acc[24 + j];
            fin += p0;
        }
        if (ok) logit0[((size_t)t * Q + q) * K + k] = temperature * (pre * fin);
    }
}


// EM_GAUSSIAN_COV E-step (em_gaussian_cov.py:106-129):
//   logit[t,q,k] = -1/2 sum_d ((w[t,k,d] - z[t,q,d])^2 * s[t,k,d]) + 1/2 sum_d log(s[t,k,d] + eps),
// both sums in torch's last-dim order, the log being MKL's vsLn.  Blocking as k_kmeans_logits_rows.
template <int E, int kRowsPerBlock>
__global__ __launch_bounds__(256) void k_cov_logits_rows(const float* __restrict__ w, const float* __restrict__ s,
                                                         const float* __restrict__ z, const uint8_t* __restrict__ need,
                                                         int Q, int K, float* __restrict__ logit0) {
    const int lane = threadIdx.x & (kGroup - 1);
    const int group = threadIdx.x / kGroup, groups_per_block = blockDim.x / kGroup;
    const int t = blockIdx.y, k0 = blockIdx.x * kRowsPerBlock;
    bool want[kRowsPerBlock];
    bool any = false;
    float wv[kRowsPerBlock][E], sv[kRowsPerBlock][E], det[kRowsPerBlock];
#pragma unroll
    for (int j = 0; j < kRowsPerBlock; j++) {
        want[j] = k0 + j < K && need[(size_t)t * K + k0 + j];
        any = any || want[j];
        float lg[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = e * kGroup + lane;
            const bool in = want[j] && d < K;
            wv[j][e] = in ? w[((size_t)t * K + k0 + j) * K + d] : 0.0f;
            sv[j][e] = in ? s[((size_t)t * K + k0 + j) * K + d] : 0.0f;
            lg[e] = in ? log_f32(sv[j][e] + kEpsF) : 0.0f;
        }
        det[j] = 0.5f * group_sum_torch<E>(lg, K, lane);
    }
    if (!any) return;
    for (int q = group; q < Q; q += groups_per_block) {
        const float* zq = z + ((size_t)t * Q + q) * K;
        float zv[E];
#pragma unroll
        for (int e = 0; e < E; e++) {                                   // unconditional loads (index clamped into the row): one batch in flight
            const int d = e * kGroup + lane;
            zv[e] = zq[d < K ? d : K - 1];
        }
#pragma unroll
        for (int e = 0; e < E; e++) zv[e] = e * kGroup + lane < K ? zv[e] : 0.0f;
#pragma unroll
        for (int j = 0; j < kRowsPerBlock; j++) {
            if (!want[j]) continue;                                  // block-uniform
            float pr[E];
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float df = e * kGroup + lane < K ? wv[j][e] - zv[e] : 0.0f;
                pr[e] = (df * df) * sv[j][e];
            }
            const float ssum = group_sum_torch<E>(pr, K, lane);
            if (lane == 0) logit0[((size_t)t * Q + q) * K + k0 + j] = -0.5f * ssum + det[j];
        }
    }
}

// E-step, part 3: u = softmax_k(logit0 + (lambd * v) / Q), torch CPU softmax order
// (max, Sleef expf of the shifted row, 16-lane strided sum + butterfly, one reciprocal).
// One 16-lane group per (task, query) row.  Also argmax (first maximum; pick_min: first minimum,
// hard_kmeans.py:193) and the hard one-hot.  u may be logit0 itself (each entry is read before it
// is written, by the same thread).
__global__ __launch_bounds__(256) void k_softmax(const float* logit0, const float* __restrict__ v, int TQ,
                                                 int Q, int K, float lambd, int hard, int pick_min, float* u,
                                                 int32_t* __restrict__ preds) {
    const int lane = threadIdx.x & 15;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (r >= TQ) return;
    const int t = r / Q;
    const float* x = logit0 + (size_t)r * K;
    const float* vt = v ? v + (size_t)t * K : nullptr;
    float* ur = u + (size_t)r * K;
    const float qf = (float)Q;
    float mx = -__builtin_inff();
    for (int k = lane; k < K; k += 16) {
        const float val = v ? x[k] + (lambd * vt[k]) / qf : x[k];
        ur[k] = val;
        mx = val > mx ? val : mx;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        const float o = __shfl_xor(mx, m, 16);
        mx = o > mx ? o : mx;
    }
    float acc = 0.0f;
    bool first = true;
    for (int k = lane; k < K; k += 16) {
        const float ev = exp_f32_sleef(ur[k] - mx);
        ur[k] = ev;
        acc = first ? ev : acc + ev;
        first = false;
    }
    float total;
    if (K < 16) {   // vec_reduce_all on a partial vector: serial over the elements
        total = __shfl(acc, 0, 16);
        for (int k = 1; k < K; k++) total += __shfl(acc, k, 16);
    } else {
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 16);
        total = acc;
    }
    const float inv = 1.0f / total;
    const float sgn = pick_min ? -1.0f : 1.0f;       // compare sgn * u: softmax values lie in [0, 1]
    float best = -2.0f;
    int best_k = 0x7fffffff, first_nan = 0x7fffffff;
    for (int k = lane; k < K; k += 16) {
        const float uv = ur[k] * inv;
        ur[k] = uv;
        if (sgn * uv > best) { best = sgn * uv; best_k = k; }
        if (uv != uv && k < first_nan) first_nan = k;
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
        const float ob = __shfl_xor(best, m, 16);
        const int ok = __shfl_xor(best_k, m, 16);
        if (ob > best || (ob == best && ok < best_k)) { best = ob; best_k = ok; }
        const int on = __shfl_xor(first_nan, m, 16);
        first_nan = on < first_nan ? on : first_nan;
    }
    // torch.argmax / argmin treat a NaN as the extremum and return the first one (features off the simplex give
    // NaN rows: log of a negative number); the prediction is used as an index later, so it must lie in [0, K)
    if (first_nan < K) best_k = first_nan;
    if (hard)
        for (int k = lane; k < K; k += 16) ur[k] = (k == best_k) ? 1.0f : 0.0f;
    if (lane == 0) preds[r] = best_k;
}

// ------------------------------------------------------------------------------------------
// Convergence record (em_dirichlet.py:236-239): per task ||alpha_old - alpha||_F / ||alpha_old||_F, then
// alpha_old <- alpha, with torch's fp32 norm: `x.norm(dim=(1,2))` is ONE serial pass over the task's K*K elements
// with eight fused multiply-add accumulators by element index mod 8, the accumulators added in order, the n mod 8
// tail (first four as product + add, the rest fused), one correctly rounded square root (probed bit for bit up to
// 10^6 elements, oracle/mathcheck.cpp::mc_norm8).  One wavefront per task; lane j (and its seven copies 8 i + j) owns
// accumulator j.  The chain of n/8 dependent FMAs per accumulator is inherent; everything else is kept off it (main loop
// below).  (A variant with four tasks per wavefront and DPP operands had a quarter of the wavefronts and was slower.)
#ifndef TCLIP_CRITERION_PACKED
#define TCLIP_CRITERION_PACKED 1
#endif
#ifndef TCLIP_CRITERION_DEPTH
#define TCLIP_CRITERION_DEPTH 8
#endif
__global__ __launch_bounds__(64) void k_criterion(const float* __restrict__ alpha, float* __restrict__ alpha_old, int K, int T,
                                                  float* __restrict__ ratio) {
    const int t = blockIdx.x, lane = threadIdx.x, j = lane & 7;
    if (t >= T) return;
    const size_t n = (size_t)K * K, base = (size_t)t * n, nv = n & ~(size_t)7;
    float acc_d = 0.0f, acc_o = 0.0f;                         // meaningful in lanes 0..7
    auto step = [&](float o, float c, int groups) {           // `groups` whole groups of eight elements, in order
        const float d = o - c;
        for (int i = 0; i < groups; i++) {
            const float vd = __shfl(d, 8 * i + j, 64), vo = __shfl(o, 8 * i + j, 64);
            acc_d = __builtin_fmaf(vd, vd, acc_d);
            acc_o = __builtin_fmaf(vo, vo, acc_o);
        }
    };
    size_t s0 = 0;
    // Main loop, 512 elements at a time: the wavefront loads them coalesced, writes alpha_old, and transposes the pairs (d = o - c, o)
    // through LDS so that lane j finds the 64 operand pairs of accumulator j (elements j, 8 + j, ..., 504 + j of the block) contiguous
    // in its row; the rows are read back as 16-byte vectors and feed the two FMA chains (one packed FMA per pair), while the next
    // block's global loads are in flight.  (Round 2 fetched every operand with a cross-lane shuffle: 16 per 64 elements in front of 16 dependent FMAs,
    // ~400 cycles per 64 elements where the chains need ~100; one wavefront per task has nothing else to hide that behind,
    // which the few-shot runs - 33 tasks per stream - paid 20 times per run.)
    // kDepth x 64 elements per block.  Round 6 measured what bounds a lone wavefront here (few-shot: 25-50 tasks per launch, 1.2 ms per
    // call): not the look-ahead - blocks of 1024 / 2048 elements (one block's chain then lasts as long as a trip to HBM) were 3 % / 11 %
    // SLOWER - and not the packed form - the two chains as interleaved plain v_fma_f32 were 12 % slower: the cadence of the dependent
    // chain itself (~22 clocks per element), which is the reference's order (profiles/r06_ab_criterion.txt).
    constexpr int kDepth = TCLIP_CRITERION_DEPTH, kRow = 8 * kDepth + 4;    // rows of 8 kDepth (d, o) pairs, padded by 4: the transposing 8-byte writes of a half-wavefront hit 32 different bank pairs
    __shared__ __attribute__((aligned(16))) float2 sdo[8 * kRow];
    if (s0 + 64 * kDepth <= nv) {
        const int wi = lane >> 3;                             // this lane holds operand 8 k + wi of accumulator j in chunk k
        float o[kDepth], c[kDepth];
#pragma unroll
        for (int k = 0; k < kDepth; k++) {
            o[k] = alpha_old[base + s0 + 64 * k + lane];
            c[k] = alpha[base + s0 + 64 * k + lane];
        }
        f2 acc{0.0f, 0.0f};                                   // {acc_d, acc_o}: the two chains share packed FMAs
        for (bool more = true; more;) {
#pragma unroll
            for (int k = 0; k < kDepth; k++) {
                alpha_old[base + s0 + 64 * k + lane] = c[k];
                sdo[j * kRow + 8 * k + wi] = float2{o[k] - c[k], o[k]};
            }
            s0 += 64 * kDepth;
            more = s0 + 64 * kDepth <= nv;
            if (more) {
#pragma unroll
                for (int k = 0; k < kDepth; k++) {
                    o[k] = alpha_old[base + s0 + 64 * k + lane];
                    c[k] = alpha[base + s0 + 64 * k + lane];
                }
            }
            __syncthreads();                                  // one wavefront per block: orders the LDS writes before the reads
#pragma unroll
            for (int m = 0; m < 4 * kDepth; m++) {
                const float4 v = *reinterpret_cast<const float4*>(&sdo[j * kRow + 2 * m]);
#if TCLIP_CRITERION_PACKED
                const f2 p0{v.x, v.y}, p1{v.z, v.w};
                acc = pk_fma(p0, p0, acc);
                acc = pk_fma(p1, p1, acc);
#else
                // the two chains as plain FMAs, interleaved (measured slower, see above)
                acc.x = __builtin_fmaf(v.x, v.x, acc.x);
                acc.y = __builtin_fmaf(v.y, v.y, acc.y);
                acc.x = __builtin_fmaf(v.z, v.z, acc.x);
                acc.y = __builtin_fmaf(v.w, v.w, acc.y);
#endif
            }
            __syncthreads();                                  // ... and the reads before the next block's writes
        }
        acc_d = acc.x;
        acc_o = acc.y;
    }
    for (; s0 + 64 <= nv; s0 += 64) {
        const float o = alpha_old[base + s0 + lane], c = alpha[base + s0 + lane];
        alpha_old[base + s0 + lane] = c;
        step(o, c, 8);
    }
    {   // the last, partial step: the rest of the whole 8-element groups and the n mod 8 tail (< 64 elements together)
        const size_t i0 = s0 + lane;
        const bool in = i0 < n;
        const float o = in ? alpha_old[base + i0] : 0.0f, c = in ? alpha[base + i0] : 0.0f;
        const float d = o - c;
        if (in) alpha_old[base + i0] = c;
        const int groups = (int)((nv - s0) >> 3);             // wave-uniform, 0..7
        step(o, c, groups);
        float bd = __shfl(acc_d, 0, 64), bo = __shfl(acc_o, 0, 64);
#pragma unroll
        for (int l = 1; l < 8; l++) {
            bd += __shfl(acc_d, l, 64);
            bo += __shfl(acc_o, l, 64);
        }
        const int ntail = (int)(n - nv), tail0 = 8 * groups;  // the tail sits in lanes tail0 .. tail0 + ntail - 1
        for (int k = 0; k < ntail; k++) {
            const float vd = __shfl(d, tail0 + k, 64), vo = __shfl(o, tail0 + k, 64);
            if (ntail >= 4 && k < 4) {                        // the vectorised epilogue: product, then add
                bd = bd + vd * vd;
                bo = bo + vo * vo;
            } else {
                bd = __builtin_fmaf(vd, vd, bd);
                bo = __builtin_fmaf(vo, vo, bo);
            }
        }
        if (lane == 0) ratio[t] = __builtin_sqrtf(bd) / __builtin_sqrtf(bo);
    }
}

__global__ void k_criterion_mean(const float* __restrict__ ratio, int N, int force_zero, float* __restrict__ out, int stride) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const float* r = ratio + (size_t)b * N;
    const float s = dsum_inner_serial(N, [&](int i) { return r[i]; });
    out[(size_t)b * stride] = force_zero ? 0.0f : s / (float)N;
}

// PADDLE prototype initialisation (few_shot/paddle.py:127-140): class means of the support set.
__global__ void k_div_rows(const float* __restrict__ num, const float* __restrict__ den, size_t n, int K, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = num[i] / den[i / K];
}

// ---- BD-CSPN (few_shot/bdcspn.py)
// torch's x.norm(p=2, dim=-1) on the AVX-512 host (probed bit for bit, 12 900 rows of width 1..1000):
// eight accumulators, element d goes to accumulator d mod 8 with one fused multiply-add, for the
// d < 8*floor(K/8); the accumulators are added in order 0..7; the remaining K mod 8 elements
// follow, the first four of them (if there are that many) as rounded product + add, the rest
// fused; one correctly rounded square root.  Eight consecutive lanes (j = 0..7) share a row;
// `get(d)` yields element d.  The result is valid in all eight lanes.
template <typename F>
__device__ __forceinline__ float row_norm_torch(int K, int j, F get) {
    const int nv = K & ~7;
    float acc = 0.0f;
    for (int d = j; d < nv; d += 8) {
        const float v = get(d);
        acc = __builtin_fmaf(v, v, acc);
    }
    float b = __shfl(acc, 0, 8);
#pragma unroll
    for (int l = 1; l < 8; l++) b += __shfl(acc, l, 8);
    int d = nv;
    if (K - d >= 4) {
        for (int k = 0; k < 4; k++, d++) {
            const float v = get(d);
            b = b + v * v;
        }
    }
    for (; d < K; d++) {
        const float v = get(d);
        b = __builtin_fmaf(v, v, b);
    }
    return __builtin_sqrtf(b);
}

// out[t,c] = mean_r x[t,r,c] = (torch outer sum over the R rows) / R      (bdcspn.py:165 train_mean, :127 eta)
__global__ void k_col_mean(const float* __restrict__ x, int T, int R, int K, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * K) return;
    const int t = i / K, c = i % K;
    const float* xt = x + (size_t)t * R * K;
    out[i] = dsum_outer(R, c, K, [&](int r) { return xt[(size_t)r * K + c]; }) / (float)R;
}

// Feature normalisation (bdcspn.py:77-100) and get_logits' own (:50-51), eight lanes per row:
//   mode 0 (UN) copy; 1 (L2N) x / ||x||; 2 (CL2N) (x - mean_t) / ||x - mean_t||.
// With `shift` (rows >= shift_from of every task get + shift[t,:] first; the augmented set of
// proto_rectification, :128-131) the source rows come from two arrays: rows < R0 from x, the rest from x2.
__global__ __launch_bounds__(256) void k_bdcspn_normalize(const float* __restrict__ x, const float* __restrict__ x2, int R0,
                                                          int R, int K, int mode, const float* __restrict__ mean,
                                                          const float* __restrict__ shift, int n_rows,
                                                          float* __restrict__ out) {
    const int j = threadIdx.x & 7;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (row >= n_rows) return;
    const int t = row / R, r = row % R;
    const float* src = r < R0 ? x + ((size_t)t * R0 + r) * K : x2 + ((size_t)t * (R - R0) + (r - R0)) * K;
    const float* mt = mean ? mean + (size_t)t * K : nullptr;
    const float* sh = (shift && r >= R0) ? shift + (size_t)t * K : nullptr;
    auto get = [&](int d) {
        float v = src[d];
        if (sh) v = v + sh[d];
        if (mode == 2) v = v - mt[d];
        return v;
    };
    float* o = out + (size_t)row * K;
    if (mode == 0) {
        for (int d = j; d < K; d += 8) o[d] = get(d);
        return;
    }
    const float nrm = row_norm_torch(K, j, get);
    for (int d = j; d < K; d += 8) o[d] = get(d) / nrm;
}

// eta[t,c] = mean_s zs[t,s,c] - mean_q zq[t,q,c]                                              (bdcspn.py:127)
__global__ void k_bdcspn_eta(const float* __restrict__ zs, const float* __restrict__ zq, int T, int S, int Q, int K,
                             float* __restrict__ eta) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * K) return;
    const int t = i / K, c = i % K;
    const float* a = zs + (size_t)t * S * K;
    const float* b = zq + (size_t)t * Q * K;
    const float ms = dsum_outer(S, c, K, [&](int r) { return a[(size_t)r * K + c]; }) / (float)S;
    const float mq = dsum_outer(Q, c, K, [&](int r) { return b[(size_t)r * K + c]; }) / (float)Q;
    eta[i] = ms - mq;
}

// KL_KMEANS (kl_kmeans.py:123-189).  Centroids: w = (u^T z) / max(sum_q u, 1), zero for empty clusters.
// torch's bmm is MKL sgemm, which for these shapes accumulates every output as one chain of fused
// multiply-adds over the queries in ascending order (probed against torch on soft and one-hot u:
// identical on every entry), so a thread keeps kMstatsRows such chains for its feature column.
// Below 400 multiply-adds per matrix (Q * K * K < 400, i.e. K = 2 at 75 queries) ATen does not call MKL but
// its own triple loop (baddbmm_cpu_kernel), whose accumulation is a rounded product plus an add.
__global__ __launch_bounds__(64) void k_kl_centroids(const float* __restrict__ u, const float* __restrict__ z,
                                                     const float* __restrict__ cs, int Q, int K, float* __restrict__ w) {
    const int t = blockIdx.z, k0 = blockIdx.y * kMstatsRows;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= K) return;
    const float* ut = u + (size_t)t * Q * K;
    const float* zt = z + (size_t)t * Q * K + d;
    float acc[kMstatsRows];
#pragma unroll
    for (int j = 0; j < kMstatsRows; j++) acc[j] = 0.0f;
    const bool fused = (long)Q * K * K >= 400;
    for (int q = 0; q < Q; q++) {
        const float zv = zt[(size_t)q * K];
#pragma unroll
        for (int j = 0; j < kMstatsRows; j++) {
            const int k = k0 + j < K ? k0 + j : K - 1;
            const float uv = ut[(size_t)q * K + k];
            acc[j] = fused ? __builtin_fmaf(uv, zv, acc[j]) : acc[j] + uv * zv;
        }
    }
#pragma unroll
    for (int j = 0; j < kMstatsRows; j++) {
        if (k0 + j >= K) break;
        const size_t row = (size_t)t * K + k0 + j;
        const float c = cs[row];
        w[row * K + d] = (acc[j] / (c < 1.0f ? 1.0f : c)) * (c > 0.0f ? 1.0f : 0.0f);
    }
}

// divs[t,q,k] = sum_d P log(P / Q), P = z[t,q,d] + eps, Q = w[t,k,d] + eps, the sum in torch's last-dim
// order (kl_kmeans.py:123-127); same blocking as k_kmeans_logits_rows.
template <int E, int kRowsPerBlock>
__global__ __launch_bounds__(256) void k_kl_divergences(const float* __restrict__ w, const float* __restrict__ z, int Q, int K,
                                                        float* __restrict__ divs) {
    const int lane = threadIdx.x & (kGroup - 1);
    const int group = threadIdx.x / kGroup, groups_per_block = blockDim.x / kGroup;
    const int t = blockIdx.y, k0 = blockIdx.x * kRowsPerBlock;
    float qv[kRowsPerBlock][E];
#pragma unroll
    for (int j = 0; j < kRowsPerBlock; j++) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const int d = e * kGroup + lane;
            qv[j][e] = (k0 + j < K && d < K) ? w[((size_t)t * K + k0 + j) * K + d] + kEpsF : 1.0f;
        }
    }
    __shared__ uint32_t s_buckets[64];
    __shared__ float s_t1[32], s_t2[32];
    if (threadIdx.x < 64) s_buckets[threadIdx.x] = kRcp14Buckets.e[threadIdx.x];
    if (threadIdx.x < 32) { s_t1[threadIdx.x] = kSlnT1[threadIdx.x]; s_t2[threadIdx.x] = kSlnT2[threadIdx.x]; }
    __syncthreads();
    // probability features keep P, Q and P / Q inside the range of the short exact quotient and of the
    // restated MKL log's main path; anything else (checked per wave) takes the IEEE operator and log_f32
    bool q_in = true;
#pragma unroll
    for (int j = 0; j < kRowsPerBlock; j++)
#pragma unroll
        for (int e = 0; e < E; e++) q_in = q_in && fast_range_f32(qv[j][e]);
    const bool q_ok = __all(q_in);
    for (int q = group; q < Q; q += groups_per_block) {
        const float* zq = z + ((size_t)t * Q + q) * K;
        float pv[E];
        bool p_in = true;
#pragma unroll
        for (int e = 0; e < E; e++) {                                   // unconditional loads (index clamped): one batch in flight
            const int d = e * kGroup + lane;
            pv[e] = zq[d < K ? d : K - 1];
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            pv[e] = e * kGroup + lane < K ? pv[e] + kEpsF : 1.0f;
            p_in = p_in && fast_range_f32(pv[e]);
        }
        const bool fast = q_ok && __all(p_in);                       // wave-uniform
#pragma unroll
        for (int j = 0; j < kRowsPerBlock; j++) {
            if (k0 + j >= K) break;                                  // block-uniform
            float pr[E];
            if (fast) {
#pragma unroll
                for (int e = 0; e < E; e++)
                    pr[e] = e * kGroup + lane < K ? pv[e] * log_mkl_inrange_tab(div_rn_inrange_f32(pv[e], qv[j][e]), s_buckets, s_t1, s_t2) : 0.0f;
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) pr[e] = e * kGroup + lane < K ? pv[e] * log_f32(pv[e] / qv[j][e]) : 0.0f;
            }
            const float ssum = group_sum_torch<E>(pr, K, lane);
            if (lane == 0) divs[((size_t)t * Q + q) * K + k0 + j] = ssum;
        }
    }
}

// KL divergences with one lane per class (round 4), the structure of k_kmeans_logits_tile: the tile holds Q = w + eps of 64
// centroids, the wavefront's query row P = z + eps is wave-uniform (scalar loads), a lane keeps torch's 32 partial sums of
// sum_d P log(P / Q) for its class.  Per (query, class, d): the short exact quotient, the restated MKL logarithm (its three
// tables in LDS), one multiply, one add - and nothing across lanes (k_kl_divergences: ~25 cross-lane instructions per 32-lane
// sum, each query row re-read per four classes).  A step whose 32 elements are not all inside the range of the fast forms
// (wave-uniform test) takes the IEEE quotient and log_f32, which agree with the fast forms wherever those are defined.
// z + 1e-15 lies inside the fast forms' range (positive normal, |exponent| <= 60) whenever z itself is +0 .. 2^60: a test on
// the bits of the wave-uniform z, i.e. on the scalar unit (sufficient, not necessary: anything else takes the generic forms)
__device__ __forceinline__ bool kl_p_surely_fast(float z) { return f32_bits(z) <= 0x5d800000u; }
__global__ __launch_bounds__(TCLIP_KMEANS_TILE_THREADS) void k_kl_divergences_tile(const float* __restrict__ w, const float* __restrict__ z,
                                                                                  int Q, int K, int stride, float* __restrict__ divs) {
    extern __shared__ float wt[];                                   // [kKmeansTile][stride]: w + eps
    __shared__ uint32_t s_buckets[64];
    __shared__ float s_t1[32], s_t2[32];
    const int t = blockIdx.y, k0 = blockIdx.x * kKmeansTile;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
    const int k = k0 + lane;
    if (threadIdx.x < 64) s_buckets[threadIdx.x] = kRcp14Buckets.e[threadIdx.x];
    if (threadIdx.x < 32) { s_t1[threadIdx.x] = kSlnT1[threadIdx.x]; s_t2[threadIdx.x] = kSlnT2[threadIdx.x]; }
    const int rows = K - k0 < kKmeansTile ? K - k0 : kKmeansTile;
    const float* wsrc = w + ((size_t)t * K + k0) * K;
    bool q_in = true;
    {
        const int n = rows * K, step = blockDim.x;
        const int pad = stride - K;
        for (int i0 = threadIdx.x; i0 < n; i0 += 8 * step) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = i0 + j * step < n ? wsrc[i0 + j * step] + kEpsF : 1.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j * step;
                q_in = q_in && fast_range_f32(v[j]);
                if (i < n) wt[i + (pad ? i / K : 0)] = v[j];
            }
        }
    }
    const bool q_ok = __syncthreads_and(q_in);                      // every Q of the tile inside the fast forms' range
    const float* wl = wt + (lane < rows ? lane : rows - 1) * stride;
    const int vec_size = K >> 3, size_ilp = vec_size >> 2, nleft = vec_size - 4 * size_ilp, ntail = K - 8 * vec_size;
    auto term_fast = [&](float pz, float qv) {
        const float P = pz + kEpsF;
        return P * log_mkl_inrange_tab(div_rn_inrange_f32(P, qv), s_buckets, s_t1, s_t2);
    };
    auto term_any = [&](float pz, float qv) {
        const float P = pz + kEpsF;
        return P * log_f32(P / qv);
    };
    auto term = [&](float pz, float qv, bool fast) { return fast ? term_fast(pz, qv) : term_any(pz, qv); };
    for (int q = wave; q < Q; q += n_waves) {
        const float* zq = z + ((size_t)t * Q + q) * K;              // wave-uniform: scalar loads
        float acc[32];
#pragma unroll
        for (int sl = 0; sl < 32; sl++) acc[sl] = 0.0f;
        for (int m = 0; m < size_ilp; m++) {
            bool p_in = true;
#pragma unroll
            for (int sl = 0; sl < 32; sl++) p_in = p_in && kl_p_surely_fast(zq[32 * m + sl]);
            if (q_ok && p_in) {                                     // wave-uniform (the row is): one straight block of fast forms
#pragma unroll
                for (int sl = 0; sl < 32; sl++) acc[sl] += term_fast(zq[32 * m + sl], wl[32 * m + sl]);
            } else {
#pragma unroll
                for (int sl = 0; sl < 32; sl++) acc[sl] += term_any(zq[32 * m + sl], wl[32 * m + sl]);
            }
        }
        int d = 32 * size_ilp;
        for (int i = 0; i < nleft; i++, d += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += term(zq[d + j], wl[d + j], q_ok && kl_p_surely_fast(zq[d + j]));
        }
        float fin = 0.0f;
        for (int i = 0; i < ntail; i++) fin += term(zq[d + i], wl[d + i], q_ok && kl_p_surely_fast(zq[d + i]));
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float p0 = acc[j];
            p0 += acc[8 + j];
            p0 += acc[16 + j];
            p0 += acc[24 + j];
            fin += p0;
        }
        if (k < K) divs[((size_t)t * Q + q) * K + k] = fin;
    }
}

// labels[r] = first index of the smallest of the K values of row r (torch.argmin).
__global__ void k_argmin_rows(const float* __restrict__ x, int rows, int K, int32_t* __restrict__ labels) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * K;
    float best = xr[0];
    int best_k = 0;
    for (int k = 1; k < K; k++)
        if (xr[k] < best) { best = xr[k]; best_k = k; }
    labels[r] = best_k;
}

// labels[r] = torch.argmax of row r: first maximum, a NaN counts as the maximum (inductive_clip.py:47).
__global__ void k_argmax_rows(const float* __restrict__ x, long rows, int K, int32_t* __restrict__ labels) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * K;
    float best = xr[0];
    int best_k = 0;
    for (int k = 1; k < K; k++) {
        const float v = xr[k];
        if (best == best && (v > best || v != v)) { best = v; best_k = k; }
    }
    labels[r] = best_k;
}

// HARD_KMEANS helpers.  Centroids of empty clusters are zero (hard_kmeans.py:149-152).
__global__ void k_zero_dead_rows(const uint8_t* __restrict__ live, int TK, int K, float* __restrict__ w) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)TK * K; i += (size_t)gridDim.x * blockDim.x)
        if (!live[i / K]) w[i] = 0.0f;
}

// Per task ||u_old - one_hot(labels)||_F (hard_kmeans.py:197), then u <- one_hot(labels).
// One block per task, fp64 accumulation.
__global__ __launch_bounds__(256) void k_hard_assign(const int32_t* __restrict__ labels, int Q, int K, float* __restrict__ u,
                                                     float* __restrict__ change) {
    const int t = blockIdx.x;
    const size_t n = (size_t)Q * K, base = (size_t)t * n;
    double a = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float hot = ((int)(i % K) == labels[(size_t)t * Q + i / K]) ? 1.0f : 0.0f;
        const double d = (double)u[base + i] - (double)hot;
        a += d * d;
        u[base + i] = hot;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sh[threadIdx.x] += sh[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) change[t] = (float)__builtin_sqrt(sh[0]);
}

// ------------------------------------------------------------------------------------------
// Accuracy tail helpers.
__global__ void k_one_hot(const int32_t* __restrict__ preds, size_t TQ, int K, float* __restrict__ hot) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < TQ * K; i += (size_t)gridDim.x * blockDim.x)
        hot[i] = ((int)(i % K) == preds[i / K]) ? 1.0f : 0.0f;
}

// Clusters present in preds[t,:] in first-appearance order (utils.py:387-394) and their
// prototype rows copied out of the dense [T,K,K] statistics.
__global__ void k_gather_prototypes(const int32_t* __restrict__ preds, const float* __restrict__ dense, int Q, int K,
                                    int Cmax, int32_t* __restrict__ n_clusters, int32_t* __restrict__ cluster_ids,
                                    float* __restrict__ protos) {
    const int t = blockIdx.x;
    __shared__ int ids[1024];
    __shared__ int cnt;
    if (threadIdx.x == 0) {
        int c = 0;
        for (int q = 0; q < Q; q++) {
            int p = preds[(size_t)t * Q + q];
            p = p < 0 ? 0 : (p >= K ? K - 1 : p);          // never index outside the task (the host side rejects such labels)
            bool seen = false;
            for (int i = 0; i < c; i++) seen |= ids[i] == p;
            if (!seen) ids[c++] = p;
        }
        cnt = c;
        n_clusters[t] = c;
        for (int i = 0; i < Cmax; i++) cluster_ids[(size_t)t * Cmax + i] = i < c ? ids[i] : -1;
    }
    __syncthreads();
    const int c = cnt;
    for (int i = threadIdx.x; i < c * K; i += blockDim.x) {
        const int ci = i / K, d = i % K;
        protos[((size_t)t * Cmax + ci) * K + d] = dense[((size_t)t * K + ids[ci]) * K + d];
    }
}


// Range check of device-resident index tensors (tclip_check_task_indices): raises `bit` in *flag when some value lies
// outside [0, limit) - what torch's own `table[idx]` turns into an IndexError (eval_zero_shot.py:160-163).
template <typename I>
__global__ void k_check_range(const I* __restrict__ v, size_t n, int64_t limit, int32_t* __restrict__ flag, int bit) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int64_t x = (int64_t)v[i];
        bad = bad || x < 0 || x >= limit;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

__global__ void k_gather_rows(const float* __restrict__ table, int64_t n_rows, int K, const int64_t* __restrict__ idx,
                              int64_t n_out, float* __restrict__ out) {
    const int64_t r = blockIdx.x;
    if (r >= n_out) return;
    const int64_t src = idx[r];
    if (src < 0 || src >= n_rows) return;
    for (int d = threadIdx.x; d < K; d += blockDim.x) out[r * K + d] = table[src * K + d];
}

// ------------------------------------------------------------------------------------------
// Probability-feature front-end (reference: src/utils.py:287-290): for every image embedding f,
//   z = softmax_k( (T * f/||f||) . text_k ),  text_k unit-norm class text embeddings.
// One block per image; the scaled embedding is staged in LDS, each thread owns classes
// k = tid, tid+256, ... and sweeps the text matrix (K x D, L2-resident) with 16-byte loads.
__global__ __launch_bounds__(256) void k_probability_features(const float* __restrict__ f, const float* __restrict__ text,
                                                              int D, int K, float temperature, float* __restrict__ z) {
    extern __shared__ float sh[];                 // D floats: scaled embedding; then K floats: logits
    float* emb = sh;
    float* logit = sh + D;
    __shared__ float red[8];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* fr = f + (size_t)n * D;
    float ss = 0.0f;
    for (int d = tid; d < D; d += blockDim.x) ss += fr[d] * fr[d];
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float nrm = __builtin_sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int d = tid; d < D; d += blockDim.x) emb[d] = temperature * (fr[d] / nrm);
    __syncthreads();
    float mx = -__builtin_inff();
    for (int k = tid; k < K; k += blockDim.x) {
        const float* tr = text + (size_t)k * D;
        float acc = 0.0f;
        int d = 0;
        if ((D & 3) == 0) {
            for (; d < D; d += 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(tr + d);
                acc = __builtin_fmaf(emb[d], t4.x, acc);
                acc = __builtin_fmaf(emb[d + 1], t4.y, acc);
                acc = __builtin_fmaf(emb[d + 2], t4.z, acc);
                acc = __builtin_fmaf(emb[d + 3], t4.w, acc);
            }
        }
        for (; d < D; d++) acc = __builtin_fmaf(emb[d], tr[d], acc);
        logit[k] = acc;
        mx = acc > mx ? acc : mx;
    }
    for (int m = 32; m >= 1; m >>= 1) { const float o = __shfl_xor(mx, m, 64); mx = o > mx ? o : mx; }
    __syncthreads();
    if (lane == 0) red[4 + wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    float sum = 0.0f;
    for (int k = tid; k < K; k += blockDim.x) {
        const float e = exp_f32_sleef(logit[k] - mx);
        logit[k] = e;
        sum += e;
    }
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
    __syncthreads();
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int k = tid; k < K; k += blockDim.x) z[(size_t)n * K + k] = logit[k] * inv;
}

// ------------------------------------------------------------------------------------------
// Device self-test (tclip_selftest_primitives).
//   counters 0..5: mismatches of the fast exact primitives / the branch-free MM update against
//                  the compiler's IEEE operators and the generic routines;
//   counters 6..13: checksums of the restated library routines (tclip_selftest_inputs.h), to be
//                  compared with the host build's checksums.
__global__ void k_selftest(unsigned long long* out) {
    __shared__ LogTabEntry tab[16];
    load_log_table(tab);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0;
    // (0) reciprocal: every float of the binade [1,2), at exponents -60, 0 and 60; (5) two-step form
    for (uint32_t m = tid; m < (1u << 23); m += nth) {
        const float x = bits_f32(0x3f800000u | m);
        b0 += rcp_rn_f32(x) != 1.0f / x;
        b5 += rcp_rn2_f32(x) != 1.0f / x;
        const float xs = x * 0x1p-60f, xl = x * 0x1p60f;
        b0 += rcp_rn_f32(xs) != 1.0f / xs;
        b0 += rcp_rn_f32(xl) != 1.0f / xl;
    }
    // (1) quotient: 2^28 pseudo-random pairs with exponents in [-40, 40], both signs
    for (uint32_t i = tid; i < (1u << 28); i += nth) {
        const float a = rand_float(i, 1u, -40, 81), b = rand_float(i, 2u, -40, 81);
        b1 += div_rn_inrange_f32(a, b) != a / b;
        b1 += div_rn_inrange_f32(-a, b) != (-a) / b;
    }
    // (2,3) fused digamma/lgamma of a+1 against the generic routines; (4) whole update
    for (uint32_t i = tid; i < (1u << 24); i += nth) {
        const float a = rand_float(i, 3u, -45, 70);                    // 2^-45 .. 2^25
        float p, l;
        digamma_lgamma_xp1(a, tab, p, l);
        b2 += p != digamma_f32(a + 1.0f);
        b2 += digamma_pos_f32(a, tab) != digamma_f32(a);
        b3 += l != lgamma_f32(a + 1.0f);
        const uint32_t h = mix32(i * 7u + 3u);
        const float y = -0.001f - 40.0f * (float)(h & 0xffffu) / 65536.0f;
        const float ps = 0.5f + 14.0f * (float)(h >> 16) / 65536.0f;
        const float uf = mm_update(a, y, ps, tab), ug = mm_update_generic(a, y, ps);
        const bool differ = !(uf == ug || (uf != uf && ug != ug));    // both NaN (0/0 after exact cancellation) is agreement
        b4 += differ;
        if (differ) { out[14] = f32_bits(a); out[15] = f32_bits(y); out[16] = f32_bits(ps); out[17] = f32_bits(uf); out[18] = f32_bits(ug); }
        // the packed form, on (a, a second argument concentrated on 2^-12 .. 2^6)
        const float a2 = rand_float(i, 4u, -12, 18);
        const f2 av{a, a2};
        const f2 lgb{a + 1.0f >= 2.3f ? lgamma_sleef_ge23<true>(a + 1.0f) : 0.0f,
                     a2 + 1.0f >= 2.3f ? lgamma_sleef_ge23<true>(a2 + 1.0f) : 0.0f};
        const f2 up = pk_mm_update(av, f2{y, y}, pk(ps), lgb, tab);
        const float ug2 = mm_update_generic(a2, y, ps);
        const bool differ0 = !(up.x == ug || (up.x != up.x && ug != ug));
        const bool differ1 = !(up.y == ug2 || (up.y != up.y && ug2 != ug2));
        b4 += differ0;
        b4 += differ1;
        if (differ0) { out[14] = f32_bits(a); out[15] = f32_bits(y); out[16] = f32_bits(ps); out[17] = f32_bits(up.x); out[18] = f32_bits(ug); }
        if (differ1) { out[14] = f32_bits(a2); out[15] = f32_bits(y); out[16] = f32_bits(ps); out[17] = f32_bits(up.y); out[18] = f32_bits(ug2); }
        // the split kernel's form: digamma and lgamma of a+1 handed over per component; without the small-parameter select
        // where it may be dropped
        float p2, l2;
        digamma_lgamma_xp1(a2, tab, p2, l2);
        const f2 yy{y, y};
        const f2 us = pk_mm_update_stage2(pk_mm_update_stage1_given<true>(av, yy, ps, p, p2, l, l2));
        b4 += !(us.x == ug || (us.x != us.x && ug != ug)) || !(us.y == ug2 || (us.y != us.y && ug2 != ug2));
        if (a > 1e-11f && a2 > 1e-11f) {
            const f2 un = pk_mm_update_stage2(pk_mm_update_stage1_given<false>(av, yy, ps, p, p2, l, l2));
            b4 += f32_bits(un.x) != f32_bits(us.x) || f32_bits(un.y) != f32_bits(us.y);
        }
    }
    // (4, continued) total cancellation beside a slot beyond the row: component x has lgamma(a+1) == digamma(a+1) * a handed in, so
    // t == 0, curvature 0 and deno == 0 (the reference's quotient is nume * inf: +inf or NaN); component y is what a ragged
    // register pair holds beyond the row, a == 0, whose curvature without the small-parameter select is NaN.  The form without
    // the select must still give component x what the form with it gives (round 4 tested deno.x * deno.y == 0: NaN hid the zero).
    for (uint32_t i = tid; i < (1u << 18); i += nth) {
        const float a = rand_float(i, 5u, -30, 34);                    // 2^-30 .. 2^4 > 1e-11
        const uint32_t h = mix32(i * 11u + 5u);
        const float y = -0.001f - 40.0f * (float)(h & 0xffffu) / 65536.0f;
        const float ps = 0.5f + 14.0f * (float)(h >> 16) / 65536.0f;
        const float p = ((i & 1u) ? 1.0f : -1.0f) * rand_float(i, 6u, -8, 14);
        const float l = p * a;
        float p0, l0;
        digamma_lgamma_xp1(0.0f, tab, p0, l0);
        const f2 av{a, 0.0f}, yy{y, 0.0f};
        const f2 us = pk_mm_update_stage2(pk_mm_update_stage1_given<true>(av, yy, ps, p, p0, l, l0));
        const f2 un = pk_mm_update_stage2(pk_mm_update_stage1_given<false>(av, yy, ps, p, p0, l, l0));
        b4 += !(f32_bits(un.x) == f32_bits(us.x) || (un.x != un.x && us.x != us.x));
        b4 += !(us.x != us.x || __builtin_fabsf(us.x) == __builtin_inff());   // nume * inf
        const f2 vs = pk_mm_update_stage2(pk_mm_update_stage1_given<false>(f2{0.0f, a}, f2{0.0f, y}, ps, p0, p, l0, l));   // the other order
        b4 += !(f32_bits(vs.y) == f32_bits(us.x) || (vs.y != vs.y && us.x != us.x));
    }
    // (3, continued) the fp64 form of lgamma on every float of [1, 2.3): where it is sure it must agree
    for (uint32_t b = f32_bits(1.0f) + tid; b < f32_bits(2.3f); b += nth) {
        const float x = bits_f32(b);
        bool sure;
        const float fast = lgamma_sleef_1_23_f64(x, sure);
        b3 += sure && f32_bits(fast) != f32_bits(lgamma_sleef_05_23(x));
    }
    // the fp64 form of the large-argument lgamma on every float of [2.3, 2^41]; above 7 also its no-shift specialisation
    for (uint32_t b = f32_bits(2.3f) + tid; b <= f32_bits(0x1p41f); b += nth) {
        const float x = bits_f32(b);
        bool sure;
        const float fast = lgamma_sleef_ge23_f64<true>(x, sure);
        b3 += sure && f32_bits(fast) != f32_bits(lgamma_sleef_ge23<true>(x));
        if (x > 7.0f) {
            bool sure7;
            const float fast7 = lgamma_sleef_gt7_f64<true>(x, sure7);
            b3 += sure7 && f32_bits(fast7) != f32_bits(lgamma_sleef_ge23<true>(x));
        }
    }
    // (2, continued) digamma in the pieces the class-split kernel uses, on every float of [1, 16): the closed form of where
    // the recurrence leaves x against the loop, the packed forms against the scalar ones, the whole against digamma_xp1
    for (uint32_t b = f32_bits(1.0f) + tid; b < f32_bits(16.0f); b += nth) {
        const float x1 = bits_f32(b);
        const float xc = digamma_rec_x(x1), acc = digamma_rec_acc(x1);
        b2 += f32_bits(xc) != f32_bits(digamma_rec_x_loop(x1));
        b2 += f32_bits(digamma_after_rec(xc, acc, tab)) != f32_bits(digamma_pos_f32(x1, tab));
        b2 += f32_bits(digamma_after_rec_ge10<false>(xc, acc, tab)) != f32_bits(digamma_pos_f32(x1, tab));
        if (x1 >= 10.0f) b2 += f32_bits(digamma_after_rec_ge10<true>(x1, 0.0f, tab)) != f32_bits(digamma_pos_f32(x1, tab));
        const f2 xp{x1, bits_f32(b ^ 0x00400000u)};           // a second argument from the other half of the binade
        const f2 xcp = pk_digamma_rec_x(xp), accp = pk_digamma_rec_acc(xp), psip = pk_digamma_after_rec(xcp, accp, tab);
        b2 += f32_bits(xcp.x) != f32_bits(xc) || f32_bits(accp.x) != f32_bits(acc);
        if (x1 < 2.3f && xp.y < 2.3f) {
            const f2 a8 = pk_digamma_rec_acc_lt23(xp);
            b2 += f32_bits(a8.x) != f32_bits(accp.x) || f32_bits(a8.y) != f32_bits(accp.y);
        }
        if (x1 >= 2.3f) b2 += f32_bits(digamma_rec_acc_ge23(x1)) != f32_bits(acc);
        b2 += f32_bits(psip.x) != f32_bits(digamma_pos_f32(xp.x, tab)) || f32_bits(psip.y) != f32_bits(digamma_pos_f32(xp.y, tab));
    }
    // (4, continued) torch.sqrt on the derived table (whose estimate is not VRSQRT14PS's at the exact powers of 4) against the
    // restatement on the instruction's own values: every float of [1, 4) - all table entries, both parities - at three
    // scales, and every power of two of the range
    for (uint32_t m = tid; m < (1u << 24); m += nth) {
        const float x = bits_f32(0x3f800000u + m);
        const f2 r = pk_sqrt_torch_inrange(f2{x, x * 0x1p-60f});
        b4 += f32_bits(r.x) != f32_bits(sqrt_torch_inrange_f32(x)) || f32_bits(r.y) != f32_bits(sqrt_torch_inrange_f32(x * 0x1p-60f));
        b4 += f32_bits(sqrt_torch_inrange_dev(x * 0x1p61f)) != f32_bits(sqrt_torch_inrange_f32(x * 0x1p61f));
    }
    for (int e = -100 + (int)tid; e <= 100; e += (int)nth) {
        const float x = bits_f32((uint32_t)(127 + e) << 23);
        b4 += f32_bits(sqrt_torch_inrange_dev(x)) != f32_bits(sqrt_torch_inrange_f32(x));
        b4 += f32_bits(pk_sqrt_torch_inrange(pk(x)).y) != f32_bits(sqrt_torch_inrange_f32(x));
    }
    atomicAdd(&out[0], b0); atomicAdd(&out[1], b1); atomicAdd(&out[2], b2);
    atomicAdd(&out[3], b3); atomicAdd(&out[4], b4); atomicAdd(&out[5], b5);
    for (int f = 0; f < kSelfTestFunctions; f++) {
        unsigned long long c = 0;
        for (uint32_t i = tid; i < kSelfTestCount; i += nth) c += selftest_term(f, i, tab);
        atomicAdd(&out[6 + f], c);
    }
}

// ------------------------------------------------------------------------------------------
// host side
thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

#define TCLIP_HIP(call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) return fail(TCLIP_ERR_HIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Optional instrumentation (bench.py): HIP events around every k_mm_live launch and a device
// counter of the element-updates it executes.  Thread-local, off by default, never touched otherwise.
struct Profile {
    bool on = false;
    std::vector<hipEvent_t> ev;      // start/stop pairs, reused across collections
    std::vector<uint8_t> kind;       // per pair: 0 = k_mm_live, 1 = k_mm_split
    size_t used = 0;
    unsigned long long* counter = nullptr;      // [4]: element-updates executed by k_mm_live / k_mm_split; k_mm_split's wavefront-iterations / full placements
    // per-kernel figures of the last collection (tclip_profile_last_kernels)
    double last_busy[2] = {0, 0}, last_sum[2] = {0, 0};
    int64_t last_launches[2] = {0, 0}, last_updates[2] = {0, 0};
    int64_t last_split_iterations = 0, last_split_sorts = 0;     // k_mm_split: wavefront-iterations, and how many of them sorted their queues anew
};
static thread_local bool g_last_mm_was_split = false;     // which kernel the last launch_mm(kMMSplit / kMMLive) started
thread_local Profile g_prof;
static int g_probe_chunks = TCLIP_PROBE_CHUNKS;     // tclip_debug_set_probe_chunks
static int g_dead_head = TCLIP_DEAD_HEAD;           // tclip_debug_set_dead_head: iterations a fresh dead row runs before the early probe; 0 = no early probe
static int g_rowset_min_rows = -1;                  // tclip_debug_set_rowset_min_rows; negative: the default rule
static int g_mm_split = -1;                         // tclip_debug_set_mm_split: 0 never, 1 always, negative: from the second outer iteration on
static int g_split_keep_placement = 1;              // tclip_debug_set_split_keep_placement: 0 = k_mm_split sorts its queues in every iteration

static hipEvent_t prof_event() {
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_prof.ev.push_back(e);
    }
    return g_prof.ev[g_prof.used++];
}

constexpr int kDecideSlices = 64;     // blocks per batch in the first stage of the stop test (large batches)
struct Layout {
    size_t logz, y, alpha_old, beta_dead, sup, cnt, cs, live, rowc, logit0, cache, cache_len, rowpart, mm_rows,
        mm_rows2, mm_rows3, dead_counts, cls, n_live, live_rows, counts, flags, stop, ratio, dpart, total;
    int n_checks;
};

static int n_chunks_of(int iter_mm) { return iter_mm <= 51 ? 1 : 1 + (iter_mm - 51 + 49) / 50; }
static int n_checks_of(int iter_mm) { return iter_mm <= 50 ? 0 : (iter_mm - 1) / 50; }

static Layout make_layout(const tclip_problem& p) {
    Layout L;
    const size_t T = (size_t)p.n_batches * p.tasks_per_batch, K = p.n_class, Q = p.n_query;
    const bool zs = p.n_support == 0;
    L.n_checks = n_checks_of(p.iter_mm);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return r; };
    L.logz = take(T * Q * K * 4);
    L.y = take(T * K * K * 4);
    L.alpha_old = take(T * K * K * 4);
    L.beta_dead = take(zs ? T * K * K * 4 : 0);
    L.sup = take(zs ? 0 : T * K * K * 4);
    L.cnt = take(zs ? 0 : T * K * 4);
    L.cs = take(T * K * 4);
    L.live = take(T * K);
    L.rowc = take(T * K * 4);
    L.logit0 = take(T * Q * K * 4);
    L.cache = take(zs ? T * K * (size_t)L.n_checks * 16 : 0);
    L.cache_len = take(T * K * 4);
    L.rowpart = take(T * K * 16);
    L.mm_rows = take(T * K * 4);
    L.mm_rows2 = take(zs ? T * K * 4 : 0);
    L.mm_rows3 = take(zs ? T * K * 4 : 0);                    // rows the early probe hands back to the old dead-row path
    L.dead_counts = take(((size_t)n_chunks_of(p.iter_mm) + 3) * 4);    // [chunk] + one spare + [n_chunks + 2]: length of mm_rows3
    L.cls = take(zs ? T * K * 2 : 0);                         // compacted lists of live classes (k_live_class_lists)
    L.n_live = take(zs ? T * 4 : 0);
    L.live_rows = take(T * K * 4);
    L.counts = take(256);
    L.flags = take(256);
    L.stop = take((size_t)p.n_batches * 4);
    L.ratio = take(T * 4);
    L.dpart = take((size_t)p.n_batches * kDecideSlices * 2 * sizeof(double));
    L.total = o;
    return L;
}

static int check_problem(const tclip_problem* p) {
    if (!p) return fail(TCLIP_ERR_ARG, "problem is null");
    if (p->n_batches < 1 || p->tasks_per_batch < 1 || p->n_query < 1 || p->iters < 0 || p->iter_mm < 1 ||
        p->n_support < 0)
        return fail(TCLIP_ERR_ARG, "non-positive size in tclip_problem");
    if (p->n_class < 2 || p->n_class > 1024) return fail(TCLIP_ERR_ARG, "n_class must be in 2..1024");
    if (p->n_support > 16000) return fail(TCLIP_ERR_ARG, "n_support must be <= 16000 (member lists are staged in LDS)");
    if ((long long)p->n_batches * p->tasks_per_batch > 65535) return fail(TCLIP_ERR_ARG, "n_batches*tasks_per_batch must be <= 65535 per call");
    if ((size_t)p->n_batches * p->tasks_per_batch * p->n_class > 0x7fffffffu)
        return fail(TCLIP_ERR_ARG, "n_batches*tasks_per_batch*n_class must fit in int32");
    return TCLIP_OK;
}

template <template <int> class Launcher, typename... Args>
static void dispatch_E(int K, Args... args) {
    const int need = (K + kGroup - 1) / kGroup;
    if (need <= 1) Launcher<1>::run(args...);
    else if (need <= 2) Launcher<2>::run(args...);
    else if (need <= 3) Launcher<3>::run(args...);
    else if (need <= 4) Launcher<4>::run(args...);
    else if (need <= 6) Launcher<6>::run(args...);
    else if (need <= 8) Launcher<8>::run(args...);
    else if (need <= 10) Launcher<10>::run(args...);
    else if (need <= 13) Launcher<13>::run(args...);
    else if (need <= 16) Launcher<16>::run(args...);
    else if (need <= 20) Launcher<20>::run(args...);
    else if (need <= 24) Launcher<24>::run(args...);
    else if (need <= 28) Launcher<28>::run(args...);
    else Launcher<32>::run(args...);
}

// The MM kernels: lanes per row and registers per lane from the row length.  Rows of up to 256 elements are spread
// over 16 lanes (4 rows per wavefront, E = ceil(K / 16) registers), which fills the lanes (K = 100: 89 % as 16 x 7
// instead of 78 % as 32 x 4; K = 10: 62 % instead of 31 %) and shares the per-row work among more rows.  Measured on
// 1000 tasks (MM loop, 32 -> 16 lanes): K = 10 95 -> 71 ms, K = 37 186 -> 151, K = 47 217 -> 174, K = 100 476 -> 411
// (460 with round 1's two rows per 32-lane group), K = 196 974 -> 895.  g_rowset_min_rows == 0 (test hook) forces the
// 32-lane layout, which must give the same bits.
#ifndef TCLIP_LOGITS_GRID
#define TCLIP_LOGITS_GRID 16384
#endif
#ifndef TCLIP_G8_MAX_K
#define TCLIP_G8_MAX_K 0              // 8 lanes per row: measured slower than 16 at 1000 tasks (too few wavefronts: K = 10 / 37 / 47:
                                      // 81 / 166 / 211 ms against 71 / 151 / 174), 2 % faster at 3000 tasks of K = 100; not compiled by default
#endif
#ifndef TCLIP_G16_MAX_K
#define TCLIP_G16_MAX_K 256
#endif
#ifndef TCLIP_MM_LAUNCH_WAVES
#define TCLIP_MM_LAUNCH_WAVES 4
#endif
enum MMKind { kMMLive = 0, kMMDead = 1, kMMProbe = 2, kMMSplit = 3, kMMProbeHead = 4 };
template <int E, int G, int KC = 0>
static void launch_mm_EG(int dead, int rows, hipStream_t st, const MMArgs& a) {
    constexpr int kWaves = TCLIP_MM_LAUNCH_WAVES, kRowsPerBlock = (64 / G) * kWaves;
    int grid = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    if (grid > 256 * 16) grid = 256 * 16;
    if (dead == kMMSplit) {                        // live rows through the class-split kernel where it exists (TCLIP_SPLIT_MIN_E <= E <= TCLIP_SPLIT_MAX_E)
        if constexpr (E <= TCLIP_SPLIT_MAX_E && E >= TCLIP_SPLIT_MIN_E) {
            constexpr int kSplitRows = 64 / G;           // one wavefront per block
            int sgrid = (rows + kSplitRows - 1) / kSplitRows;
            if (sgrid > 256 * 64) sgrid = 256 * 64;
            hipLaunchKernelGGL((k_mm_split<E, G, KC>), dim3(sgrid), dim3(64), 0, st, a);
            g_last_mm_was_split = true;
            return;
        }
        dead = kMMLive;
    }
    if (dead == kMMProbe) hipLaunchKernelGGL((k_mm_probe<E, G>), dim3(grid), dim3(256), 0, st, a);
    else if (dead == kMMProbeHead) {
        if constexpr (KC == 0) hipLaunchKernelGGL((k_mm_probe_head<E, G>), dim3(grid), dim3(256), 0, st, a);      // (launch_mm: probes have no fixed-K build)
    } else if (dead) hipLaunchKernelGGL((k_mm_live<E, kWaves, true, 1, G, KC>), dim3(grid), dim3(64 * kWaves), 0, st, a);
    else hipLaunchKernelGGL((k_mm_live<E, kWaves, false, 1, G, KC>), dim3(grid), dim3(64 * kWaves), 0, st, a);
}
template <int G>
static void launch_mm_G(int need, int dead, int rows, hipStream_t st, const MMArgs& a) {
    if (G < 32 || need <= 8) {
        if (need <= 1) return launch_mm_EG<1, G>(dead, rows, st, a);
        if (need <= 2) return launch_mm_EG<2, G>(dead, rows, st, a);
        if (need <= 3) return launch_mm_EG<3, G>(dead, rows, st, a);
        if (need <= 4) return launch_mm_EG<4, G>(dead, rows, st, a);
        if (need <= 5) return launch_mm_EG<5, G>(dead, rows, st, a);
        if (need <= 6) return launch_mm_EG<6, G>(dead, rows, st, a);
        if (need <= 7) return launch_mm_EG<7, G>(dead, rows, st, a);
        if (need <= 8) return launch_mm_EG<8, G>(dead, rows, st, a);
    }
    if (need <= 10) return launch_mm_EG<10, G>(dead, rows, st, a);
    if (need <= 13) return launch_mm_EG<13, G>(dead, rows, st, a);
    if (need <= 16) return launch_mm_EG<16, G>(dead, rows, st, a);
    if constexpr (G == 32) {
        if (need <= 20) return launch_mm_EG<20, G>(dead, rows, st, a);
        if (need <= 24) return launch_mm_EG<24, G>(dead, rows, st, a);
        if (need <= 28) return launch_mm_EG<28, G>(dead, rows, st, a);
        return launch_mm_EG<32, G>(dead, rows, st, a);
    }
}
// registers per lane launch_mm_G<G> picks for `need`
static int mm_regs_of(int need, int G) {
    if ((G < 32 || need <= 8) && need <= 8) return need;
    if (need <= 10) return 10;
    if (need <= 13) return 13;
    if (need <= 16) return 16;
    if (need <= 20) return 20;
    if (need <= 24) return 24;
    return need <= 28 ? 28 : 32;
}
// does launch_mm(kMMSplit, K, ..) start k_mm_split (true) or fall back to k_mm_live (no instantiation for this row length)?
static bool mm_has_split(int K) {
    const bool wide = g_rowset_min_rows == 0;
    int E;
    if (K >= TCLIP_G64_MIN_K && K >= 512 && !wide && TCLIP_G64_MIN_K > 0) E = 16;
    else if (K <= TCLIP_G8_MAX_K && !wide) E = mm_regs_of((K + 7) / 8, 8);            // launch_mm's order of tests
    else if (K <= TCLIP_G16_MAX_K && !wide) E = mm_regs_of((K + 15) / 16, 16);
    else E = mm_regs_of((K + 31) / 32, 32);
    return E <= TCLIP_SPLIT_MAX_E && E >= TCLIP_SPLIT_MIN_E;
}
// Row lengths the MM kernels are also compiled for as constants - the class counts of the reference's datasets that BASELINE.json
// runs (ImageNet 1000, SUN397 397, Caltech101-sized 100): same code, same layout as the run-time-K kernel of the row length's
// bucket, with every mask and tail of a ragged last register resolved by the compiler (round 4: the run-time forms kept
// 2 x 4 lane masks per row sum in spilled scalar registers).  g_fixed_k_kernels == 0 (tclip_debug_set_fixed_k_kernels, tests):
// the run-time-K kernels for every row length, which must give the same bits.
#ifndef TCLIP_FIXED_K_DEFAULT
#define TCLIP_FIXED_K_DEFAULT 1
#endif
static int g_fixed_k_kernels = TCLIP_FIXED_K_DEFAULT;
static void launch_mm(int dead, int K, int rows, hipStream_t st, const MMArgs& a) {
    const bool wide = g_rowset_min_rows == 0;              // test hook: the 32-lane layout for every row length
    if (!wide && g_fixed_k_kernels && dead != kMMProbe && dead != kMMProbeHead) {
#if TCLIP_G64_MIN_K > 0 && TCLIP_G64_MIN_K <= 1000
        if (K == 1000) return launch_mm_EG<16, 64, 1000>(dead, rows, st, a);
#endif
#if TCLIP_G16_MAX_K < 397
        if (K == 397) return launch_mm_EG<13, 32, 397>(dead, rows, st, a);
#endif
#if TCLIP_G16_MAX_K >= 100 && TCLIP_G8_MAX_K < 100
        if (K == 100) return launch_mm_EG<7, 16, 100>(dead, rows, st, a);
#endif
    }
#if TCLIP_G64_MIN_K > 0
    if (K >= TCLIP_G64_MIN_K && K >= 512 && !wide) return launch_mm_EG<16, 64>(dead, rows, st, a);   // 512: the cascade's first dump
#endif
#if TCLIP_G8_MAX_K > 0
    if (K <= TCLIP_G8_MAX_K && !wide) return launch_mm_G<8>((K + 7) / 8, dead, rows, st, a);
#endif
    if (K <= TCLIP_G16_MAX_K && !wide) launch_mm_G<16>((K + 15) / 16, dead, rows, st, a);
    else launch_mm_G<32>((K + 31) / 32, dead, rows, st, a);
}
template <int E> struct LaunchRowConsts {
    static void run(int grid, hipStream_t st, const float* alpha, const int32_t* rows, const int32_t* n, int K, float* rowc) {
        hipLaunchKernelGGL(k_row_consts<E>, dim3(grid), dim3(256), 0, st, alpha, rows, n, K, rowc);
    }
};
template <int E> struct LaunchLogits {
    static void run(int grid, hipStream_t st, const float* alpha, const float* logz, const float* rowc,
                    const int32_t* rows, const int32_t* n, int Q, int K, float* logit0, const int32_t* only_if) {
        constexpr int kRows = E <= 16 ? 4 : 2;                       // registers: kRows x E for the rows' alpha
        hipLaunchKernelGGL((k_logits<E, kRows>), dim3((grid + 7) / 8 * 8), dim3(256), 0, st, alpha, logz, rowc, rows, n, Q, K, logit0,
                           only_if);
    }
};

template <int E> struct LaunchKlDivergences {
    static void run(int T, hipStream_t st, const float* w, const float* z, int Q, int K, float* divs) {
        constexpr int kRows = E <= 16 ? 4 : 2;
        hipLaunchKernelGGL((k_kl_divergences<E, kRows>), dim3((K + kRows - 1) / kRows, T), dim3(256), 0, st, w, z, Q, K, divs);
    }
};
template <int E> struct LaunchCovLogitsRows {
    static void run(int T, hipStream_t st, const float* w, const float* s, const float* z, const uint8_t* need, int Q, int K,
                    float* logit0) {
        constexpr int kRows = E <= 8 ? 4 : (E <= 16 ? 2 : 1);        // registers: 2 x kRows x E for centroids and covariances
        hipLaunchKernelGGL((k_cov_logits_rows<E, kRows>), dim3((K + kRows - 1) / kRows, T), dim3(256), 0, st, w, s, z, need, Q, K,
                           logit0);
    }
};
template <int E> struct LaunchKmeansLogitsRows {
    static void run(int T, hipStream_t st, const float* w, const float* z, const uint8_t* need, int Q, int K, float pre,
                    float temperature, float* logit0) {
        constexpr int kRows = E <= 16 ? 4 : 2;                       // registers: kRows x E for the centroids
        hipLaunchKernelGGL((k_kmeans_logits_rows<E, kRows>), dim3((K + kRows - 1) / kRows, T), dim3(256), 0, st, w, z, need, Q, K,
                           pre, temperature, logit0);
    }
};

static int g_kmeans_tile = -1;       // tclip_debug_set_kmeans_tile: 0 = k_kmeans_logits_rows for every K, negative: the default rule
// squared distances to the centroids: one lane per class where the row length allows it (k_kmeans_logits_tile), else 32 lanes per class
// the tile kernels' dynamic LDS goes beyond the 64 KB a kernel gets without asking (once per kernel and process)
static bool kmeans_tile_lds_raised(const void* kernel) {
    struct Seen { const void* kernel; int device; bool ok; };
    static std::mutex mu;                                     // entry points are re-entrant across host threads
    static std::vector<Seen> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : seen)
        if (e.kernel == kernel && e.device == dev) return e.ok;
    const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kKmeansTile * 511 * (int)sizeof(float)) == hipSuccess;
    seen.push_back(Seen{kernel, dev, ok});
    return ok;
}
static void launch_kmeans_logits(int T, hipStream_t st, const float* w, const float* z, const uint8_t* need, int Q, int K, float pre,
                                 float temperature, float* logit0) {
    if (g_kmeans_tile != 0 && K >= 32 && K <= 511) {
        const int stride = K | 1;
        const size_t lds = (size_t)kKmeansTile * stride * sizeof(float);
        const bool raised = kmeans_tile_lds_raised((const void*)k_kmeans_logits_tile);
        if (raised) {
            hipLaunchKernelGGL(k_kmeans_logits_tile, dim3((K + kKmeansTile - 1) / kKmeansTile, T), dim3(kKmeansTileThreads), lds, st, w, z, need, Q, K,
                               stride, pre, temperature, logit0);
            return;
        }
    }
    dispatch_E<LaunchKmeansLogitsRows>(K, T, st, w, z, need, Q, K, pre, temperature, logit0);
}

// XCD-aware block -> (task, tile) mapping of the kernels whose blocks share a task's operands (round 6).  The hardware deals
// consecutive workgroups to the eight XCDs in turn and every XCD has its own L2: with the task as the slowest grid dimension
// the (K/64)^2 blocks of one task landed on all eight XCDs and each L2 fetched the task's u and f (2 x 119 KB at K = 397)
// for itself.  Here the grid is one-dimensional: block L runs on XCD L % 8 and works on task 8 (L / 8 / tiles) + L % 8, tile
// (L / 8) % tiles of it - the blocks of ONE task go to ONE XCD, whose L2 holds the operands once.
// TCLIP_XCD_MAP=0 (A/B builds): the plain order, task slowest.
#ifndef TCLIP_XCD_MAP
#define TCLIP_XCD_MAP 1
#endif
constexpr int kXcds = 8;
struct TaskTile { int t, bx, by; };
__device__ __forceinline__ TaskTile task_tile_of_block(int nx, int ny) {
    const int per_task = nx * ny;
#if TCLIP_XCD_MAP
    const int slot = blockIdx.x / kXcds, xcd = blockIdx.x % kXcds;
    const int t = (slot / per_task) * kXcds + xcd, inner = slot % per_task;
#else
    const int t = blockIdx.x / per_task, inner = blockIdx.x % per_task;
#endif
    return TaskTile{t, inner % nx, inner / nx};                 // the caller returns at once when t >= T
}
static unsigned task_tile_grid(int nx, int ny, int T) { return (unsigned)((long)((T + kXcds - 1) / kXcds) * kXcds * nx * ny); }

// M-step statistics / centroids / prototypes: rows whose K columns all lie in torch's cascade region
// go through the 8-rows-per-thread kernel, the last few rows through the one-row kernel.
// The same statistics with the task's feature columns staged ONCE per block (round 4), for the reference's 75 queries: a block
// owns 64 columns d of one task - f[t, 0..74, d0..d0+63] in 19 KB of LDS - and a range of classes; each of its four
// wavefronts walks its share of the classes in chunks of eight, lane = column; u[t, q, k .. k+7] is wave-uniform and arrives
// as one scalar load per (q, chunk).  k_mstats_rows reads the task's feature block once per eight classes from L2 (K = 397:
// 49 times, 6.6 GB per call, which is what bounded it: 1.65 ms for 1 000 tasks); here it is read once per `rows_per_block`
// classes and the kernel is left with its 2 x 75 VALU instructions per output.
// Same operations in the same order: products u f added in query order, a0 dumped into a1 after every 16 queries
// (dsum_cascade with n = 75: four dumps, eleven leftovers), a0 + a1 + a2 + a3 with a2 = a3 = +0 (the zero additions
// decide the sign of a zero sum).  Rows of the cascade region only, as k_mstats_rows.
// kCov: EM_GAUSSIAN_COV's inverse variances, y = cs / max(sum_q (wc - f)^2 u, eps), as k_mstats_rows<true>.
constexpr int kColsQ = 75, kColsChunk = 8, kColsWaves = 4;
template <bool kCov>
__global__ __launch_bounds__(64 * kColsWaves) void k_mstats_cols75(const float* __restrict__ u, const float* __restrict__ f,
                                                                   const float* __restrict__ cs, const uint8_t* __restrict__ live,
                                                                   const float* __restrict__ sup, const float* __restrict__ cnt, int K,
                                                                   int k_rows, int rows_per_block, float* __restrict__ y, int paddle,
                                                                   const float* __restrict__ wc, int T, int nx, int ny) {
    __shared__ float zt[kColsQ * 64];
    const TaskTile tt = task_tile_of_block(nx, ny);
    if (tt.t >= T) return;
    const int t = tt.t, d0 = tt.bx * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kb = tt.by * rows_per_block;
    const int ke = kb + rows_per_block < k_rows ? kb + rows_per_block : k_rows;
    if (kb >= ke) return;
    {   // nothing to do for a block without a live class (block-uniform)
        bool any = false;
        for (int k = kb; k < ke; k++) any = any || live[(size_t)t * K + k];
        if (!any) return;
    }
    const float* ft = f + (size_t)t * kColsQ * K;
    for (int i = threadIdx.x; i < kColsQ * 64; i += 64 * kColsWaves) {
        const int q = i >> 6, dc = d0 + (i & 63);
        zt[i] = ft[(size_t)q * K + (dc < K ? dc : K - 1)];
    }
    __syncthreads();
    const int d = d0 + lane;
    const float* ut = u + (size_t)t * kColsQ * K;
    const float* zl = zt + lane;
    for (int kc = kb + wave * kColsChunk; kc < ke; kc += kColsWaves * kColsChunk) {
        const int k0 = kc + kColsChunk <= k_rows ? kc : k_rows - kColsChunk;   // the last chunk overlaps its predecessor (same values again)
        bool any = false;
#pragma unroll
        for (int j = 0; j < kColsChunk; j++) any = any || live[(size_t)t * K + k0 + j];
        if (!any) continue;
        // two classes per packed instruction, spelled out (the build runs without the SLP vectoriser, which used to find these
        // pairs: 1.29 against 1.44 ms per call at K = 397): u of classes 2 j2, 2 j2 + 1 is an aligned pair of scalar registers
        f2 a0[kColsChunk / 2], a1[kColsChunk / 2];
#pragma unroll
        for (int j = 0; j < kColsChunk / 2; j++) a0[j] = a1[j] = pk(0.0f);
        f2 wcv[kColsChunk / 2];
#pragma unroll
        for (int j = 0; j < kColsChunk / 2; j++)
            wcv[j] = kCov ? f2{wc[((size_t)t * K + k0 + 2 * j) * K + (d < K ? d : K - 1)], wc[((size_t)t * K + k0 + 2 * j + 1) * K + (d < K ? d : K - 1)]} : pk(0.0f);
        auto term = [&](int j, f2 uv, f2 fv) {
            if (kCov) {
                const f2 df = wcv[j] - fv;
                return (df * df) * uv;
            }
            return uv * fv;
        };
        const float* uk = ut + k0;                                  // indexed, not walked: a pointer that moves through the loop
                                                                    // turned these wave-uniform loads into per-lane ones
#pragma unroll 1
        for (int g = 0; g < kColsQ / 8; g++) {                      // nine groups of eight queries; a dump after every second group
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int q = 8 * g + i;
                const f2 zq = pk(zl[q * 64]);
#pragma unroll
                for (int j = 0; j < kColsChunk / 2; j++) a0[j] = a0[j] + term(j, f2{uk[(size_t)q * K + 2 * j], uk[(size_t)q * K + 2 * j + 1]}, zq);
            }
            if (g & 1) {
#pragma unroll
                for (int j = 0; j < kColsChunk / 2; j++) { a1[j] = a1[j] + a0[j]; a0[j] = pk(0.0f); }
            }
        }
#pragma unroll
        for (int q = 8 * (kColsQ / 8); q < kColsQ; q++) {          // the last three
            const f2 zq = pk(zl[q * 64]);
#pragma unroll
            for (int j = 0; j < kColsChunk / 2; j++) a0[j] = a0[j] + term(j, f2{uk[(size_t)q * K + 2 * j], uk[(size_t)q * K + 2 * j + 1]}, zq);
        }
        if (d >= K) continue;
#pragma unroll
        for (int j = 0; j < kColsChunk; j++) {
            const size_t row = (size_t)t * K + k0 + j;
            if (!live[row]) continue;
            float s = (j & 1) ? a0[j >> 1].y : a0[j >> 1].x;
            s += (j & 1) ? a1[j >> 1].y : a1[j >> 1].x;
            s += 0.0f;                                  // a2, a3 of the cascade: never filled with 75 terms, but added
            s += 0.0f;
            const float c = cs[row];
            if (kCov) {
                y[row * K + d] = c / (s < kEpsF ? kEpsF : s);
            } else if (paddle == 2) {
                y[row * K + d] = s / c;
            } else if (sup && paddle) {
                y[row * K + d] = (s + sup[row * K + d]) / (c + cnt[row]);
            } else if (sup) {
                const float w = 1.0f / (cnt[row] + c);
                y[row * K + d] = w * (sup[row * K + d] + s);
            } else {
                y[row * K + d] = s / (c < kEpsF ? kEpsF : c);
            }
        }
    }
}

// Compacted list of the live classes of every task (EM-Dirichlet's M-step statistics, zero-shot): cls[t][0 .. n_live[t]) = the
// classes k < k_rows with live[t, k], ascending; one block per task.  After the first outer iteration ~5 % of a task's classes
// are alive (47 of 1000 on the bench's tasks, 28-43 of 397 in hard mode) and k_mstats_cols75, which skips dead classes by
// chunks of eight, still walked a third to a half of its chunks; k_mstats_tile75 takes this list and works on n_live / 64
// tiles of classes.
__global__ __launch_bounds__(256) void k_live_class_lists(const uint8_t* __restrict__ live, int K, int k_rows,
                                                          int16_t* __restrict__ cls, int32_t* __restrict__ n_live) {
    __shared__ int wave_count[4];
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* lt = live + (size_t)t * K;
    int16_t* out = cls + (size_t)t * K;
    int base = 0;
    for (int k0 = 0; k0 < k_rows; k0 += 256) {                     // block-uniform trip count
        const int k = k0 + threadIdx.x;
        const bool alive = k < k_rows && lt[k] != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(alive);
        if (lane == 0) wave_count[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            before += w < wave ? wave_count[w] : 0;
            total += wave_count[w];
        }
        if (alive) out[base + before + lanes_below(m)] = (int16_t)k;
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_live[t] = base;
}

// The same statistics as a register-tiled product (round 6).
// k_mstats_cols75 hands u to the arithmetic as wave-uniform scalar loads, eight dwords per query and eight packed
// instructions that use them: the counters of round 5 showed its wavefronts waiting 68 % of their cycles on those loads
// (s_load returns out of order, so every wait is for ALL of them, and the LDS reads of the feature column share the
// counter), VALU 55 % busy.  Here a wavefront owns a 32 x 32 tile of outputs, lane (jr, cc) the 4 classes x 4 columns at
// (4 jr, 4 cc): per query one 16-byte LDS read of u[q, 4 classes] and one of f[q, 4 columns] feed sixteen packed
// instructions (two reads per 16 where the column kernel needs one scalar load and one read per 8), both operands
// arrive through the in-order LDS queue, and nothing is wave-uniform.  A block of four wavefronts stages
// u[t, 0..74, 64 classes] and f[t, 0..74, 64 columns] (38 KB) once.
// Same operations in the same order per output as k_mstats_cols75 / k_mstats_rows: products u f added in query order, a0
// dumped into a1 after every 16 queries, a0 + a1 + 0 + 0.  Rows of the cascade region only (k < k_rows).
// It does the work of a whole tile for any live class in it: the host uses it over all classes where clusters do not die in
// numbers (SOFT_KMEANS, the first outer iteration of EM-Dirichlet, few-shot) and over the compacted list of the live classes
// (cls / n_live, k_live_class_lists) for the later outer iterations of zero-shot EM-Dirichlet; EM_GAUSSIAN keeps one cluster
// per task alive and stays with the column kernel, which skips by eight.
// Blocks of one task share an XCD (task_tile_of_block).
constexpr int kTileQ = 75, kTileBlock = 64;
#ifndef TCLIP_TILE_STAGE_BOTH
#define TCLIP_TILE_STAGE_BOTH 1
#endif
#ifndef TCLIP_TILE_PREFETCH
#define TCLIP_TILE_PREFETCH 1
#endif
#ifndef TCLIP_TILE_WAVES
#define TCLIP_TILE_WAVES 3
#endif
__global__ __launch_bounds__(256, TCLIP_TILE_WAVES) void k_mstats_tile75(const float* __restrict__ u, const float* __restrict__ f,
                                                       const float* __restrict__ cs, const uint8_t* __restrict__ live,
                                                       const float* __restrict__ sup, const float* __restrict__ cnt, int K, int T,
                                                       int k_rows, int tiles_d, int tiles_k, float* __restrict__ y, int paddle,
                                                       const int16_t* __restrict__ cls, const int32_t* __restrict__ n_live) {
    __shared__ __attribute__((aligned(16))) float zt[kTileQ * kTileBlock];
    __shared__ __attribute__((aligned(16))) float ut[kTileQ * kTileBlock];
    __shared__ int slot_class[kTileBlock];                      // the class behind slot kb + c of this block (-1: none)
    const TaskTile tt = task_tile_of_block(tiles_d, tiles_k);
    if (tt.t >= T) return;
    const int t = tt.t, d0 = tt.bx * kTileBlock, kb = tt.by * kTileBlock;
    // cls: the block's 64 class slots are entries kb .. kb + 63 of the task's list of live classes (k_live_class_lists), which
    // has n_slots entries; without a list slot i is class i and there are k_rows of them
    const int n_slots = cls ? n_live[t] : k_rows;
    if (kb >= n_slots) return;                                  // block-uniform: a tile of classes beyond the list
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {   // staging: thread (wave, lane) moves column `lane` of queries wave, wave + 4, ... of both operands
        int kc = kb + lane < n_slots ? kb + lane : n_slots - 1;  // slots beyond the list repeat its last class (their outputs are skipped)
        if (cls) kc = cls[(size_t)t * K + kc];
        if (wave == 0) slot_class[lane] = kb + lane < n_slots ? kc : -1;
        const int dc = d0 + lane < K ? d0 + lane : K - 1;
        const float* fp = f + ((size_t)t * kTileQ + wave) * K + dc;
        const float* upg = u + ((size_t)t * kTileQ + wave) * K + kc;
#if TCLIP_TILE_STAGE_BOTH
        float fr[19], ur[19];
#pragma unroll
        for (int j = 0; j < 19; j++) {
            const bool in = j < 18 || wave < kTileQ - 72;
            fr[j] = in ? fp[(size_t)4 * j * K] : 0.0f;
            ur[j] = in ? upg[(size_t)4 * j * K] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 19; j++) {
            if (j < 18 || wave < kTileQ - 72) {
                zt[(4 * j + wave) * kTileBlock + lane] = fr[j];
                ut[(4 * j + wave) * kTileBlock + lane] = ur[j];
            }
        }
#else
        // one operand at a time: nineteen loads in flight, then their nineteen LDS writes (both operands at once were 38 registers
        // that the accumulators' 32 and the double-buffered reads did not leave room for at four wavefronts per SIMD)
        auto stage = [&](const float* src, float* dst) {
            float r[19];
#pragma unroll
            for (int j = 0; j < 19; j++) r[j] = (j < 18 || wave < kTileQ - 72) ? src[(size_t)4 * j * K] : 0.0f;   // 4 * 18 + wave < 75
#pragma unroll
            for (int j = 0; j < 19; j++)
                if (j < 18 || wave < kTileQ - 72) dst[(4 * j + wave) * kTileBlock + lane] = r[j];
        };
        stage(fp, zt);
        stage(upg, ut);
#endif
    }
    __syncthreads();
    const int kw = kb + 32 * (wave >> 1), dw = d0 + 32 * (wave & 1);       // this wavefront's 32 x 32 tile (kw: its first slot)
    if (kw >= n_slots || dw >= K) return;
    const int jr = lane >> 3, cc = lane & 7;
    const float* up = ut + 32 * (wave >> 1) + 4 * jr;
    const float* zp = zt + 32 * (wave & 1) + 4 * cc;
    f2 a0[4][2], a1[4][2];                                                  // [class][column pair]
#pragma unroll
    for (int i = 0; i < 4; i++) a0[i][0] = a0[i][1] = a1[i][0] = a1[i][1] = pk(0.0f);
    // the operands of query q + 1 are requested before the sixteen packed instructions of query q (the LDS answers in order,
    // so the wait before them is for the older pair only)
    auto fetch = [&](int q, float4& uv, float4& fv) {
        uv = *(const float4*)(up + q * kTileBlock);
        fv = *(const float4*)(zp + q * kTileBlock);
    };
    auto apply = [&](const float4& uv, const float4& fv) {
        const f2 f01{fv.x, fv.y}, f23{fv.z, fv.w};
        const float uu[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            a0[i][0] = a0[i][0] + pk(uu[i]) * f01;
            a0[i][1] = a0[i][1] + pk(uu[i]) * f23;
        }
    };
    float4 un, fn;
    fetch(0, un, fn);
#pragma unroll 1
    for (int g = 0; g < kTileQ / 16; g++) {                                 // four groups of sixteen queries, a dump after each
#pragma unroll
        for (int i = 0; i < 16; i++) {
#if TCLIP_TILE_PREFETCH
            const float4 uc = un, fc = fn;
            fetch(16 * g + i + 1, un, fn);                                  // (query 64 after the last group: the leftovers' first)
            apply(uc, fc);
#else
            float4 uc, fc;
            fetch(16 * g + i, uc, fc);
            apply(uc, fc);
#endif
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            a1[i][0] = a1[i][0] + a0[i][0]; a0[i][0] = pk(0.0f);
            a1[i][1] = a1[i][1] + a0[i][1]; a0[i][1] = pk(0.0f);
        }
    }
#pragma unroll
    for (int q = 16 * (kTileQ / 16); q < kTileQ; q++) {                     // the last eleven
#if TCLIP_TILE_PREFETCH
        const float4 uc = un, fc = fn;
        if (q + 1 < kTileQ) fetch(q + 1, un, fn);
        apply(uc, fc);
#else
        float4 uc, fc;
        fetch(q, uc, fc);
        apply(uc, fc);
#endif
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int k = slot_class[32 * (wave >> 1) + 4 * jr + i];
        if (k < 0) continue;
        const size_t row = (size_t)t * K + k;
        if (!live[row]) continue;
        const float c = cs[row];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int d = dw + 4 * cc + j;
            if (d >= K) continue;
            float s = (j & 1) ? a0[i][j >> 1].y : a0[i][j >> 1].x;
            s += (j & 1) ? a1[i][j >> 1].y : a1[i][j >> 1].x;
            s += 0.0f;                                  // a2, a3 of the cascade: never filled with 75 terms, but added
            s += 0.0f;
            if (paddle == 2) {
                y[row * K + d] = s / c;
            } else if (sup && paddle) {
                y[row * K + d] = (s + sup[row * K + d]) / (c + cnt[row]);
            } else if (sup) {
                const float w = 1.0f / (cnt[row] + c);
                y[row * K + d] = w * (sup[row * K + d] + s);
            } else {
                y[row * K + d] = s / (c < kEpsF ? kEpsF : c);
            }
        }
    }
}

// KL_KMEANS's centroids (k_kl_centroids) in the same blocking: one chain of fused multiply-adds per output over the 75
// queries in ascending order (what MKL's sgemm does for these shapes), 64 feature columns staged once per block, u as scalar
// loads.  Q * K * K >= 400 here (K >= 8), i.e. always the fused form.
__global__ __launch_bounds__(64 * kColsWaves) void k_kl_centroids_cols75(const float* __restrict__ u, const float* __restrict__ z,
                                                                         const float* __restrict__ cs, int K, int rows_per_block,
                                                                         float* __restrict__ w, int T, int nx, int ny) {
    __shared__ float zt[kColsQ * 64];
    const TaskTile tt = task_tile_of_block(nx, ny);
    if (tt.t >= T) return;
    const int t = tt.t, d0 = tt.bx * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kb = tt.by * rows_per_block;
    const int ke = kb + rows_per_block < K ? kb + rows_per_block : K;
    if (kb >= ke) return;
    const float* ft = z + (size_t)t * kColsQ * K;
    for (int i = threadIdx.x; i < kColsQ * 64; i += 64 * kColsWaves) {
        const int q = i >> 6, dc = d0 + (i & 63);
        zt[i] = ft[(size_t)q * K + (dc < K ? dc : K - 1)];
    }
    __syncthreads();
    const int d = d0 + lane;
    const float* ut = u + (size_t)t * kColsQ * K;
    const float* zl = zt + lane;
    for (int kc = kb + wave * kColsChunk; kc < ke; kc += kColsWaves * kColsChunk) {
        const int k0 = kc + kColsChunk <= K ? kc : K - kColsChunk;   // the last chunk overlaps its predecessor (same values again)
        f2 acc[kColsChunk / 2];                                     // two classes per packed fma, as in k_mstats_cols75
#pragma unroll
        for (int j = 0; j < kColsChunk / 2; j++) acc[j] = pk(0.0f);
        const float* uk = ut + k0;
#pragma unroll 1
        for (int g = 0; g < kColsQ / 8; g++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int q = 8 * g + i;
                const f2 zq = pk(zl[q * 64]);
#pragma unroll
                for (int j = 0; j < kColsChunk / 2; j++) acc[j] = pk_fma(f2{uk[(size_t)q * K + 2 * j], uk[(size_t)q * K + 2 * j + 1]}, zq, acc[j]);
            }
        }
#pragma unroll
        for (int q = 8 * (kColsQ / 8); q < kColsQ; q++) {
            const f2 zq = pk(zl[q * 64]);
#pragma unroll
            for (int j = 0; j < kColsChunk / 2; j++) acc[j] = pk_fma(f2{uk[(size_t)q * K + 2 * j], uk[(size_t)q * K + 2 * j + 1]}, zq, acc[j]);
        }
        if (d >= K) continue;
#pragma unroll
        for (int j = 0; j < kColsChunk; j++) {
            const size_t row = (size_t)t * K + k0 + j;
            const float c = cs[row];
            const float a = (j & 1) ? acc[j >> 1].y : acc[j >> 1].x;
            w[row * K + d] = (a / (c < 1.0f ? 1.0f : c)) * (c > 0.0f ? 1.0f : 0.0f);
        }
    }
}

static int g_mstats_cols = -1;          // tclip_debug_set_kmeans_tile also switches this kernel (0: k_mstats_rows for every shape)
#ifndef TCLIP_MSTATS_TILE
#define TCLIP_MSTATS_TILE -1
#endif
static int g_mstats_tile = TCLIP_MSTATS_TILE;          // tclip_debug_set_kmeans_tile switches this kernel too (0: never)
template <bool kCov>
static void launch_mstats_mode(hipStream_t st, const float* u, const float* f, const float* cs, const uint8_t* live,
                               const float* sup, const float* cnt, int T, int Q, int K, float* y, int paddle, const float* wc,
                               bool dense = false, int16_t* cls = nullptr, int32_t* n_live = nullptr) {
    const long ncols = (long)K * K;
    const int full_rows = ncols >= 8 ? (int)(((ncols / 32) * 32) / K) : 0;      // rows 0 .. full_rows-1 are all-cascade
    int groups = full_rows / kMstatsRows;
    if (!kCov && (dense || cls) && Q == kTileQ && full_rows >= 32 && g_mstats_tile != 0) {
        // 32 x 32 register tiles, both operands through LDS: over all classes where (nearly) every class is alive (the caller's
        // word), over the compacted list of the live ones where the caller provides room for it (cls, n_live)
        const int tiles_d = (K + kTileBlock - 1) / kTileBlock, tiles_k = (full_rows + kTileBlock - 1) / kTileBlock;
        if (cls) hipLaunchKernelGGL(k_live_class_lists, dim3(T), dim3(256), 0, st, live, K, full_rows, cls, n_live);
        hipLaunchKernelGGL(k_mstats_tile75, dim3(task_tile_grid(tiles_d, tiles_k, T)), dim3(256), 0, st,
                           u, f, cs, live, sup, cnt, K, T, full_rows, tiles_d, tiles_k, y, paddle, (const int16_t*)cls, (const int32_t*)n_live);
        if (full_rows < K)
            hipLaunchKernelGGL(k_mstats<kCov>, dim3((K + 63) / 64, K - full_rows, T), dim3(64), 0, st, u, f, cs, live, sup, cnt, Q, K, y,
                               paddle, full_rows, wc);
        return;
    }
    if (Q == kColsQ && full_rows >= kColsChunk && g_mstats_cols != 0) {
        // the column kernel takes every row of the cascade region; enough blocks to fill the machine, at least 32 rows each
        const int dtiles = (K + 63) / 64;
        int splits = (int)((8192 + (long)T * dtiles - 1) / ((long)T * dtiles));
        if (splits > full_rows / (kColsWaves * kColsChunk)) splits = full_rows / (kColsWaves * kColsChunk);
        if (splits < 1) splits = 1;
        int rows_per_block = ((full_rows + splits - 1) / splits + kColsChunk - 1) / kColsChunk * kColsChunk;
        splits = (full_rows + rows_per_block - 1) / rows_per_block;
        hipLaunchKernelGGL(k_mstats_cols75<kCov>, dim3(task_tile_grid(dtiles, splits, T)), dim3(64 * kColsWaves), 0, st, u, f, cs, live, sup, cnt, K,
                           full_rows, rows_per_block, y, paddle, wc, T, dtiles, splits);
        groups = full_rows / kMstatsRows;
        const int k_first = full_rows;
        if (k_first < K)
            hipLaunchKernelGGL(k_mstats<kCov>, dim3((K + 63) / 64, K - k_first, T), dim3(64), 0, st, u, f, cs, live, sup, cnt, Q, K, y,
                               paddle, k_first, wc);
        return;
    }
    if (groups > 0)
        hipLaunchKernelGGL(k_mstats_rows<kCov>, dim3((K + 63) / 64, groups, T), dim3(64), 0, st, u, f, cs, live, sup, cnt, Q, K, y,
                           paddle, wc);
    const int k_first = groups * kMstatsRows;
    if (k_first < K)
        hipLaunchKernelGGL(k_mstats<kCov>, dim3((K + 63) / 64, K - k_first, T), dim3(64), 0, st, u, f, cs, live, sup, cnt, Q, K, y,
                           paddle, k_first, wc);
}
// dense: the caller expects (nearly) every class of every task to be alive - the register-tiled kernel
// cls / n_live: room for the compacted lists of live classes ([T, K] int16, [T] int32) - the tile kernel over those lists
static void launch_mstats(hipStream_t st, const float* u, const float* f, const float* cs, const uint8_t* live,
                          const float* sup, const float* cnt, int T, int Q, int K, float* y, int paddle, bool dense = false,
                          int16_t* cls = nullptr, int32_t* n_live = nullptr) {
    launch_mstats_mode<false>(st, u, f, cs, live, sup, cnt, T, Q, K, y, paddle, nullptr, dense, cls, n_live);
}
// EM_GAUSSIAN_COV: s = cs / max(sum_q (w - z_q)^2 u, eps) for the rows `live` marks
static void launch_cov_stats(hipStream_t st, const float* u, const float* z, const float* cs, const uint8_t* live,
                             const float* w, int T, int Q, int K, float* s) {
    launch_mstats_mode<true>(st, u, z, cs, live, nullptr, nullptr, T, Q, K, s, 0, w);
}

static int ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace tclip

using namespace tclip;

extern "C" {

int tclip_abi_version(void) { return TCLIP_ABI_VERSION; }
const char* tclip_last_error(void) { return g_err; }

}  // extern "C"

namespace tclip {

// Enqueues the whole loop for `p.n_batches` consecutive batches on stream `st`.  `crit_stride` is
// the row stride (= iters of the full problem) of criterions / mm_iters.
static int enqueue_batches(const tclip_problem& p, const RowSrc& q_src, const RowSrc& s_src, const int64_t* y_s, float* u,
                           float* v, float* alpha, int32_t* preds, float* criterions, int32_t* mm_iters,
                           char* ws, hipStream_t st) {
    const bool zs = p.n_support == 0;
    const Layout L = make_layout(p);
    const int B = p.n_batches, N = p.tasks_per_batch, Q = p.n_query, K = p.n_class, S = p.n_support;
    const int T = B * N, TK = T * K;
    const size_t TQK = (size_t)T * Q * K, TKK = (size_t)T * K * K;
    float* logz = (float*)(ws + L.logz);
    float* y = (float*)(ws + L.y);
    float* alpha_old = (float*)(ws + L.alpha_old);
    float* beta_dead = zs ? (float*)(ws + L.beta_dead) : nullptr;
    float* sup = zs ? nullptr : (float*)(ws + L.sup);
    float* cnt = zs ? nullptr : (float*)(ws + L.cnt);
    float* cs = (float*)(ws + L.cs);
    uint8_t* live = (uint8_t*)(ws + L.live);
    float* rowc = (float*)(ws + L.rowc);
    float* logit0 = (float*)(ws + L.logit0);
    double* cache = zs ? (double*)(ws + L.cache) : nullptr;
    int32_t* cache_len = (int32_t*)(ws + L.cache_len);
    double* rowpart = (double*)(ws + L.rowpart);
    int32_t* mm_rows = (int32_t*)(ws + L.mm_rows);
    int32_t* dead_list[2] = {mm_rows, zs ? (int32_t*)(ws + L.mm_rows2) : mm_rows};   // chunk c sweeps dead_list[c & 1]
    int32_t* dead_counts = (int32_t*)(ws + L.dead_counts);                        // [chunk] length of that list
    int32_t* head_back = zs ? (int32_t*)(ws + L.mm_rows3) : nullptr;              // rows the early probe could not finish
    int32_t* live_rows = (int32_t*)(ws + L.live_rows);
    int32_t* counts = (int32_t*)(ws + L.counts);
    int32_t* flags = (int32_t*)(ws + L.flags);          // [0]: some log z is not finite
    int32_t* stop = (int32_t*)(ws + L.stop);
    float* ratio = (float*)(ws + L.ratio);
    double* dpart = (double*)(ws + L.dpart);
    const int n_chunks = n_chunks_of(p.iter_mm);
    const int n_checks = zs ? L.n_checks : 0;   // few-shot has no dead rows: nothing to cache

    // ---- initialisation (em_dirichlet.py:195-211)
    TCLIP_HIP(hipMemsetAsync(flags, 0, 256, st));
    if (q_src.idx) {            // rows of a feature table through the task-batch loop's index tensor: gather, copy and log in one pass
        const size_t rows = (size_t)T * Q;
        hipLaunchKernelGGL(k_gather_log_features, dim3((unsigned)(rows > 65536 ? 65536 : rows)), dim3(256), 0, st, q_src, Q, K, rows, u, logz, flags);
    } else {
        hipLaunchKernelGGL(k_log_features, dim3(ew_grid(TQK)), dim3(256), 0, st, q_src.base, logz, TQK, flags);
        hipLaunchKernelGGL(k_copy, dim3(ew_grid(TQK)), dim3(256), 0, st, q_src.base, u, TQK);
    }
    hipLaunchKernelGGL(k_fill, dim3(ew_grid(TKK)), dim3(256), 0, st, alpha, 1.0f, TKK);
    hipLaunchKernelGGL(k_fill, dim3(ew_grid(TKK)), dim3(256), 0, st, alpha_old, 1.0f, TKK);
    hipLaunchKernelGGL(k_fill, dim3(ew_grid(TK)), dim3(256), 0, st, v, 0.0f, (size_t)TK);
    TCLIP_HIP(hipMemsetAsync(cache_len, 0, (size_t)TK * 4, st));
    if (!zs) {
        hipLaunchKernelGGL(k_support_stats, dim3(K, T), dim3(128), (size_t)S * sizeof(int), st, s_src, y_s, S, K, 1, sup, cnt);
    }
    // E-step terms of the initial alpha = 1 for every row (rows that never come alive keep them).  Every
    // row is the same all-ones vector, so lgamma(sum) - sum lgamma is evaluated for ONE row and copied
    // (at K = 1000 the generic lgamma over all T*K*K ones took a second per 417 tasks); the
    // contraction with log z runs over the full-row list 0..TK-1 once.
    {
        TCLIP_HIP(hipMemsetAsync(counts, 0, 256, st));
        hipLaunchKernelGGL(k_one_row_list, dim3(1), dim3(1), 0, st, mm_rows, counts + 2);
        dispatch_E<LaunchRowConsts>(K, 1, st, (const float*)alpha, (const int32_t*)mm_rows, (const int32_t*)(counts + 2), K, rowc);
        hipLaunchKernelGGL(k_broadcast_first, dim3(ew_grid((size_t)TK)), dim3(256), 0, st, rowc, (size_t)TK);
        // live_rows <- identity, counts[1] <- TK
        TCLIP_HIP(hipMemsetAsync(live, 1, (size_t)TK, st));
        hipLaunchKernelGGL(k_build_rows, dim3((TK + 255) / 256), dim3(256), 0, st, live, cache_len, TK, 0, mm_rows,
                           live_rows, counts, counts + 1);
        hipLaunchKernelGGL(k_init_logits, dim3(ew_grid(TQK)), dim3(256), 0, st, (const float*)rowc, (const int32_t*)flags, Q, K, TQK, logit0);
        dispatch_E<LaunchLogits>(K, TK > 262144 ? 262144 : TK, st, (const float*)alpha, (const float*)logz, (const float*)rowc,
                                 (const int32_t*)live_rows, (const int32_t*)(counts + 1), Q, K, logit0, (const int32_t*)flags);
    }

    for (int it = 0; it < p.iters; it++) {
        // ---- M-step statistics
        hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, zs ? 1 : 0,
                           cs, live, v, cache_len);
        // zero-shot: every class alive in the first outer iteration, a few per cent afterwards - the tile kernel over all classes,
        // then over the lists of the live ones; few-shot: every class alive throughout
        launch_mstats(st, (const float*)u, (const float*)logz, (const float*)cs, (const uint8_t*)live, (const float*)sup, (const float*)cnt, T, Q, K, y, 0,
                      !zs || it == 0, (zs && it > 0) ? (int16_t*)(ws + L.cls) : nullptr, (zs && it > 0) ? (int32_t*)(ws + L.n_live) : nullptr);
        TCLIP_HIP(hipMemsetAsync(counts, 0, 256, st));
        TCLIP_HIP(hipMemsetAsync(dead_counts, 0, ((size_t)n_chunks + 3) * 4, st));
        TCLIP_HIP(hipMemsetAsync(stop, 0, (size_t)B * 4, st));
        hipLaunchKernelGGL(k_build_rows, dim3((TK + 255) / 256), dim3(256), 0, st, (const uint8_t*)live,
                           (const int32_t*)cache_len, TK, n_checks, dead_list[0], live_rows, dead_counts, counts + 1);
        // ---- MM fixed point in chunks aligned with the stop-test checkpoints
        for (int c = 0; c < n_chunks; c++) {
            MMArgs a;
            a.alpha = alpha; a.beta_dead = beta_dead; a.y = y; a.live = live; a.cache_len = cache_len;
            a.cache = cache; a.rowpart = rowpart; a.rows = mm_rows; a.n_rows = counts; a.stop = stop;
            a.next_rows = nullptr; a.next_count = nullptr;
            a.K = K; a.rows_per_batch = N * K; a.chunk = c;
            a.keep_placement = g_split_keep_placement;
            a.l0 = c == 0 ? 0 : 50 * c + 1;
            a.l1 = 50 * (c + 1) < p.iter_mm - 1 ? 50 * (c + 1) : p.iter_mm - 1;
            a.has_check = (a.l1 > 0 && a.l1 % 50 == 0) ? 1 : 0;
            a.n_checks = n_checks > 0 ? n_checks : 1;
            // class-split kernel once the parameters have moved away from their start at 1 (k_mm_split)
            const bool split = g_mm_split < 0 ? it >= TCLIP_SPLIT_FROM : (g_mm_split >= 100 ? it >= g_mm_split - 100 : g_mm_split != 0);
            const bool split_runs = split && mm_has_split(K);
            a.work_counter = g_prof.on ? g_prof.counter + (split_runs ? 1 : 0) : nullptr;
            hipEvent_t e0 = g_prof.on ? prof_event() : nullptr, e1 = g_prof.on ? prof_event() : nullptr;
            if (e0 && e1) TCLIP_HIP(hipEventRecord(e0, st));
            a.rows = live_rows; a.n_rows = counts + 1;
            g_last_mm_was_split = false;
            launch_mm(split ? kMMSplit : kMMLive, K, TK, st, a);
            if (e0 && e1) {                                     // the instrumentation covers k_mm_live / k_mm_split only
                TCLIP_HIP(hipEventRecord(e1, st));
                if (g_prof.kind.size() < g_prof.used / 2) g_prof.kind.resize(g_prof.used / 2);
                g_prof.kind[g_prof.used / 2 - 1] = g_last_mm_was_split ? 1 : 0;
            }
            if (split_runs != g_last_mm_was_split) return fail(TCLIP_ERR_ARG, "internal: mm_has_split disagrees with launch_mm");
            if (zs && a.has_check) {          // dead rows only matter through their stop-test terms
                a.rows = dead_list[c & 1]; a.n_rows = dead_counts + c; a.work_counter = nullptr;
                if (c == 0 && g_dead_head > 0 && g_probe_chunks > 0) {
                    // rows that have just died: TCLIP_DEAD_HEAD iterations, then the early probe (k_mm_probe_head), which finishes
                    // every row it finds on its cycle; what is left over takes the path below from the start
                    MMArgs h = a;
                    h.l0 = 0; h.l1 = g_dead_head - 1; h.has_check = 0; h.next_rows = nullptr; h.next_count = nullptr;
                    launch_mm(kMMDead, K, TK, st, h);
                    h.l0 = g_dead_head; h.next_rows = head_back; h.next_count = dead_counts + n_chunks + 2;
                    launch_mm(kMMProbeHead, K, TK, st, h);
                    a.rows = head_back; a.n_rows = dead_counts + n_chunks + 2;
                }
                // the probe needs the row to be ON its cycle already; rows that were still approaching
                // it after chunk 0 get a few more chances before they are left to iterate every chunk
                const bool more = c + 1 < a.n_checks, probe = c < g_probe_chunks && more;
                int32_t* next_rows = more ? dead_list[(c + 1) & 1] : nullptr;
                int32_t* next_count = more ? dead_counts + c + 1 : nullptr;
                a.next_rows = probe ? nullptr : next_rows; a.next_count = probe ? nullptr : next_count;
                launch_mm(kMMDead, K, TK, st, a);
                a.next_rows = next_rows; a.next_count = next_count;
                if (probe) launch_mm(kMMProbe, K, TK, st, a);
            }
            const bool two_stage = a.has_check && N * K > 16384;
            if (two_stage)
                hipLaunchKernelGGL(k_mm_decide_partial, dim3(kDecideSlices, B), dim3(256), 0, st, (const double*)rowpart,
                                   (const double*)cache, (const uint8_t*)live, N * K, a.n_checks, c, (const int32_t*)stop, dpart);
            hipLaunchKernelGGL(k_mm_decide, dim3(B), dim3(1024), 0, st, (const double*)rowpart, (const double*)cache,
                               (const uint8_t*)live, N * K, a.n_checks, c, a.has_check, a.l1, c == n_chunks - 1 ? 1 : 0,
                               p.iter_mm, stop, mm_iters + it, p.iters, (const double*)(two_stage ? dpart : nullptr),
                               kDecideSlices);
        }
        // ---- E-step for the rows whose alpha changed, softmax over all classes
        {
            const int g8 = (TK + 7) / 8 > 65535 * 8 ? 65535 * 8 : (TK + 7) / 8;
            dispatch_E<LaunchRowConsts>(K, g8 > 4096 ? 4096 : g8, st, (const float*)alpha, (const int32_t*)live_rows,
                                        (const int32_t*)(counts + 1), K, rowc);
            dispatch_E<LaunchLogits>(K, TK > TCLIP_LOGITS_GRID ? TCLIP_LOGITS_GRID : TK, st, (const float*)alpha, (const float*)logz, (const float*)rowc,
                                     (const int32_t*)live_rows, (const int32_t*)(counts + 1), Q, K, logit0, (const int32_t*)nullptr);
        }
        hipLaunchKernelGGL(k_softmax, dim3((T * Q * 16 + 255) / 256), dim3(256), 0, st, (const float*)logit0, (const float*)v,
                           T * Q, Q, K, (float)p.lambd, p.hard, 0, u, preds);
        // ---- convergence record
        hipLaunchKernelGGL(k_criterion, dim3(T), dim3(64), 0, st, (const float*)alpha, alpha_old, K, T, ratio);
        hipLaunchKernelGGL(k_criterion_mean, dim3(B), dim3(64), 0, st, (const float*)ratio, N, (!zs && p.hard) ? 1 : 0,
                           criterions + it, p.iters);
    }
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

// Independent batches are spread over up to three streams (the caller's and two internal ones,
// forked from and joined to the caller's stream with events): every MM launch is a barrier for the
// batches it covers, and with equal-length rows its last round of waves leaves SIMDs idle; kernels
// of another group fill them.
constexpr int kMaxGroups = 16;
struct StreamPool {
    hipStream_t s[kMaxGroups] = {};
    hipEvent_t fork = nullptr, join[kMaxGroups] = {};
    bool ready = false;
    int device = -1;
};
thread_local StreamPool g_pool;

// Streams and join events are created on first use, only as many as a call needs (group 0 is the caller's stream).
static int pool_init(int groups) {
    int dev = 0;
    TCLIP_HIP(hipGetDevice(&dev));
    if (g_pool.ready && g_pool.device != dev) {                       // the calling thread moved to another device
        for (int i = 0; i < kMaxGroups; i++) {
            if (g_pool.s[i]) (void)hipStreamDestroy(g_pool.s[i]);
            if (g_pool.join[i]) (void)hipEventDestroy(g_pool.join[i]);
            g_pool.s[i] = nullptr;
            g_pool.join[i] = nullptr;
        }
        (void)hipEventDestroy(g_pool.fork);
        g_pool.fork = nullptr;
        g_pool.ready = false;
    }
    g_pool.device = dev;
    if (!g_pool.fork) TCLIP_HIP(hipEventCreateWithFlags(&g_pool.fork, hipEventDisableTiming));
    for (int i = 1; i < groups && i < kMaxGroups; i++) {
        if (g_pool.s[i]) continue;
        TCLIP_HIP(hipStreamCreateWithFlags(&g_pool.s[i], hipStreamNonBlocking));
        TCLIP_HIP(hipEventCreateWithFlags(&g_pool.join[i], hipEventDisableTiming));
    }
    g_pool.ready = true;
    return TCLIP_OK;
}

// Group 0 runs on the caller's stream, the others on pool streams.  Measured on the K=100 bench
// (10 batches): 1 group 685 ms per step, 2 or 3 groups 606 ms, 4 groups 685 ms again with HIP's
// default of four hardware queues per process (streams beyond the queues serialise behind each
// other; GPU_MAX_HW_QUEUES=8 brings 4 groups to 619 ms).  Small problems are launch-bound and
// run best on one stream (K=10, 10 batches of 10 tasks: 53 ms against 114 ms on four).
static int stream_groups() {
    static const int n = [] {
        const char* e = getenv("TCLIP_STREAM_GROUPS");           // tuning knob, 1..16
        const int v = e ? atoi(e) : 3;
        return v < 1 ? 1 : (v > kMaxGroups ? kMaxGroups : v);
    }();
    return n;
}
// TCLIP_GROUP_SIZES="5,3,2" (tuning knob): the batches of a call that has exactly their sum, dealt to the groups in these sizes
struct GroupSizes { int n = 0, size[kMaxGroups] = {}, total = 0; };
static const GroupSizes& group_sizes_override() {
    static const GroupSizes gs = [] {
        GroupSizes r;
        const char* e = getenv("TCLIP_GROUP_SIZES");
        while (e && *e && r.n < kMaxGroups) {
            const int v = atoi(e);
            if (v < 1) { r = GroupSizes(); break; }
            r.size[r.n++] = v;
            r.total += v;
            while (*e && *e != ',') e++;
            if (*e == ',') e++;
        }
        return r;
    }();
    return gs;
}
static int n_groups_of(const tclip_problem& p) {
    const GroupSizes& o = group_sizes_override();
    if (o.n > 0 && o.total == p.n_batches) return o.n;
    const long long rows = (long long)p.n_batches * p.tasks_per_batch * p.n_class;
    long long g = rows / 8192;                                   // a group should fill the machine once
    if (g > stream_groups()) g = stream_groups();
    if (g > p.n_batches) g = p.n_batches;
    return g < 1 ? 1 : (int)g;
}

static tclip_problem group_problem(const tclip_problem& p, int g, int* first_batch) {
    tclip_problem q = p;
    const GroupSizes& o = group_sizes_override();
    if (o.n > 0 && o.total == p.n_batches) {
        int b0 = 0;
        for (int i = 0; i < g; i++) b0 += o.size[i];
        q.n_batches = o.size[g];
        *first_batch = b0;
        return q;
    }
    const int G = n_groups_of(p), base = p.n_batches / G, extra = p.n_batches % G;
    q.n_batches = base + (g < extra ? 1 : 0);
    *first_batch = g * base + (g < extra ? g : extra);
    return q;
}

}  // namespace tclip

extern "C" {

size_t tclip_workspace_bytes(const tclip_problem* p) {
    if (check_problem(p) != TCLIP_OK) return 0;
    size_t total = 0;
    for (int g = 0; g < n_groups_of(*p); g++) {
        int b0;
        total += make_layout(group_problem(*p, g, &b0)).total;
    }
    return total;
}

}  // extern "C"

namespace tclip {

// Splits the call's batches into stream groups and enqueues each group (see StreamPool).
static int run_em_dirichlet(const tclip_problem& p, const RowSrc& q_src, const RowSrc& s_src, const int64_t* y_s, float* u,
                            float* v, float* alpha, int32_t* preds, float* criterions, int32_t* mm_iters,
                            void* workspace, size_t workspace_bytes, void* stream) {
    const bool zs = p.n_support == 0;
    if (workspace_bytes < tclip_workspace_bytes(&p)) return fail(TCLIP_ERR_WORKSPACE, "workspace smaller than tclip_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(TCLIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t caller = (hipStream_t)stream;
    const int G = n_groups_of(p);
    if (G > 1) {
        if (int rc = pool_init(G)) return rc;
        TCLIP_HIP(hipEventRecord(g_pool.fork, caller));
    }
    const size_t N = p.tasks_per_batch, Q = p.n_query, K = p.n_class, S = p.n_support;
    size_t ws_off[kMaxGroups + 1] = {0};
    for (int g = 0; g < G; g++) {
        int b0;
        ws_off[g + 1] = ws_off[g] + make_layout(group_problem(p, g, &b0)).total;
    }
    // the rows of the tasks from t0 on: a dense tensor moves its base, an indexed source its index (and column) rows
    auto from_task = [&](const RowSrc& src, size_t t0, size_t rows_per_task) {
        if (!src.base) return src;
        if (src.idx) return RowSrc{src.base, src.idx + t0 * rows_per_task, src.cols ? src.cols + t0 * K : nullptr};
        return RowSrc{src.base + t0 * rows_per_task * K, nullptr, nullptr};
    };
    for (int i = 0; i < G; i++) {
        const int g = (i + 1) % G;                       // pool streams first, the caller's stream last
        int b0;
        const tclip_problem q = group_problem(p, g, &b0);
        const size_t t0 = (size_t)b0 * N;
        hipStream_t st = g == 0 ? caller : g_pool.s[g];
        if (g != 0) TCLIP_HIP(hipStreamWaitEvent(st, g_pool.fork, 0));
        tclip_problem qs = q;
        qs.iters = p.iters;
        if (int rc = enqueue_batches(qs, from_task(q_src, t0, Q), from_task(s_src, t0, S), zs ? nullptr : y_s + t0 * S,
                                     u + t0 * Q * K, v + t0 * K, alpha + t0 * K * K, preds + t0 * Q,
                                     criterions + (size_t)b0 * p.iters, mm_iters + (size_t)b0 * p.iters,
                                     (char*)workspace + ws_off[g], st)) {
            // kernels already enqueued on the internal streams still use the caller's buffers: let them finish
            // before the error reaches a caller who may free them (the caller's own stream is ordered anyway)
            for (int h = 1; h < G; h++) (void)hipStreamSynchronize(g_pool.s[h]);
            return rc;
        }
        if (g != 0) TCLIP_HIP(hipEventRecord(g_pool.join[g], st));
    }
    for (int g = 1; g < G; g++) TCLIP_HIP(hipStreamWaitEvent(caller, g_pool.join[g], 0));
    return TCLIP_OK;
}

}  // namespace tclip

extern "C" {

int tclip_em_dirichlet_run(const tclip_problem* pp, const float* x_q, const float* x_s, const int64_t* y_s, float* u,
                           float* v, float* alpha, int32_t* preds, float* criterions, int32_t* mm_iters,
                           void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    const bool zs = pp->n_support == 0;
    if (!x_q || !u || !v || !alpha || !preds || !criterions || !mm_iters || !workspace)
        return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (zs != (x_s == nullptr) || zs != (y_s == nullptr))
        return fail(TCLIP_ERR_ARG, "x_s and y_s must be given exactly when n_support > 0");
    return run_em_dirichlet(*pp, dense_rows(x_q), dense_rows(x_s), y_s, u, v, alpha, preds, criterions, mm_iters, workspace,
                            workspace_bytes, stream);
}

int tclip_em_dirichlet_run_tasks(const tclip_problem* pp, const tclip_task_source* src, const int64_t* y_s, float* u,
                                 float* v, float* alpha, int32_t* preds, float* criterions, int32_t* mm_iters,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    const bool zs = pp->n_support == 0;
    if (!src || !src->table_q || !src->q_idx || !u || !v || !alpha || !preds || !criterions || !mm_iters || !workspace)
        return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (zs != (src->table_s == nullptr) || zs != (src->s_idx == nullptr) || zs != (y_s == nullptr))
        return fail(TCLIP_ERR_ARG, "table_s, s_idx and y_s must be given exactly when n_support > 0");
    return run_em_dirichlet(*pp, RowSrc{src->table_q, src->q_idx, src->cols}, RowSrc{src->table_s, src->s_idx, src->cols}, y_s,
                            u, v, alpha, preds, criterions, mm_iters, workspace, workspace_bytes, stream);
}

// ---- SOFT_KMEANS (BASELINE config 3's second method)
static size_t kmeans_ws_parts(const tclip_problem& p, size_t* o_cs, size_t* o_live, size_t* o_ones, size_t* o_logit,
                              size_t* o_rows, size_t* o_scratch_rows, size_t* o_counts) {
    const size_t T = (size_t)p.n_batches * p.tasks_per_batch, K = p.n_class, Q = p.n_query;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return r; };
    *o_cs = take(T * K * 4);
    *o_live = take(T * K);
    *o_ones = take(T * K);
    *o_logit = take(T * Q * K * 4);
    *o_rows = take(T * K * 4);
    *o_scratch_rows = take(T * K * 4);
    *o_counts = take(256);
    return o;
}

size_t tclip_soft_kmeans_workspace_bytes(const tclip_problem* p) {
    if (check_problem(p) != TCLIP_OK) return 0;
    size_t a, b, c, d, e, f, g;
    return kmeans_ws_parts(*p, &a, &b, &c, &d, &e, &f, &g);
}

// SOFT_KMEANS (v == nullptr) and EM_GAUSSIAN (v given: the class-proportion term lambd * v / Q in the
// softmax and the v update of em_gaussian.py:129-143) share everything else.
static int soft_kmeans_core(const tclip_problem& p, const float* x_q, float temperature, float* v, float* u, float* w,
                            int32_t* preds, void* workspace, size_t workspace_bytes, void* stream, const char* who) {
    if (!x_q || !u || !w || !preds || !workspace) return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (p.n_support != 0) return fail(TCLIP_ERR_ARG, "%s is a zero-shot method: n_support must be 0", who);
    size_t o_cs, o_live, o_ones, o_logit, o_rows, o_scratch, o_counts;
    const size_t total = kmeans_ws_parts(p, &o_cs, &o_live, &o_ones, &o_logit, &o_rows, &o_scratch, &o_counts);
    if (workspace_bytes < total) return fail(TCLIP_ERR_WORKSPACE, "workspace smaller than tclip_soft_kmeans_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(TCLIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int Q = p.n_query, K = p.n_class, T = p.n_batches * p.tasks_per_batch, TK = T * K;
    const size_t TQK = (size_t)T * Q * K;
    float* cs = (float*)(ws + o_cs);
    uint8_t* live = (uint8_t*)(ws + o_live);
    uint8_t* ones = (uint8_t*)(ws + o_ones);
    float* logit0 = (float*)(ws + o_logit);
    hipLaunchKernelGGL(k_copy, dim3(ew_grid(TQK)), dim3(256), 0, st, x_q, u, TQK);          // u = z
    TCLIP_HIP(hipMemsetAsync(ones, 1, (size_t)TK, st));
    if (v) hipLaunchKernelGGL(k_fill, dim3(ew_grid(TK)), dim3(256), 0, st, v, 0.0f, (size_t)TK);
    // w_init: every centroid = u^T z / clamp(sum u)                             (soft_kmeans.py:137-149)
    hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs, live,
                       (float*)nullptr, (int32_t*)nullptr);
    launch_mstats(st, (const float*)u, (const float*)x_q, (const float*)cs, (const uint8_t*)ones, (const float*)nullptr, (const float*)nullptr, T, Q, K, w, 0, true);
    // SOFT_KMEANS keeps its clusters alive (every query spreads its responsibility over all of them); EM_GAUSSIAN's class-proportion
    // term leaves a handful per task after two iterations (profiles/r05_kmeans_live_clusters.txt): only the former is "dense"
    const bool dense = v == nullptr;
    for (int it = 0; it < p.iters; it++) {
        // w_update: live clusters get the new mean, empty ones keep their centroid   (:151-168)
        // EM_GAUSSIAN: the same pass over u also yields v of the previous iteration's v_update (v stays 0 before the first)
        hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs,
                           live, it > 0 ? v : (float*)nullptr, (int32_t*)nullptr);
        launch_mstats(st, (const float*)u, (const float*)x_q, (const float*)cs, (const uint8_t*)live, (const float*)nullptr, (const float*)nullptr, T, Q, K, w, 0, dense);
        // distances only for centroids that moved (all of them in the first iteration)
        launch_kmeans_logits(T, st, (const float*)w, x_q, (const uint8_t*)(it == 0 ? ones : live), Q, K, -0.5f,
                                           temperature, logit0);
        hipLaunchKernelGGL(k_softmax, dim3((T * Q * 16 + 255) / 256), dim3(256), 0, st, (const float*)logit0,
                           (const float*)v, T * Q, Q, K, (float)p.lambd, 0, 0, u, preds);
    }
    if (v)      // the last v_update
        hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs, live,
                           v, (int32_t*)nullptr);
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

int tclip_soft_kmeans_run(const tclip_problem* pp, const float* x_q, float temperature, float* u, float* w,
                          int32_t* preds, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    return soft_kmeans_core(*pp, x_q, temperature, nullptr, u, w, preds, workspace, workspace_bytes, stream, "SOFT_KMEANS");
}

int tclip_em_gaussian_run(const tclip_problem* pp, const float* x_q, float temperature, float* u, float* v, float* w,
                          int32_t* preds, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    if (!v) return fail(TCLIP_ERR_ARG, "null pointer argument");
    return soft_kmeans_core(*pp, x_q, temperature, v, u, w, preds, workspace, workspace_bytes, stream, "EM_GAUSSIAN");
}

// ---- EM_GAUSSIAN_COV: EM_GAUSSIAN with a diagonal inverse covariance per cluster; no temperature
int tclip_em_gaussian_cov_run(const tclip_problem* pp, const float* x_q, float* u, float* v, float* w, float* s,
                              int32_t* preds, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    const tclip_problem p = *pp;
    if (!x_q || !u || !v || !w || !s || !preds || !workspace) return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (p.n_support != 0) return fail(TCLIP_ERR_ARG, "EM_GAUSSIAN_COV is a zero-shot method: n_support must be 0");
    size_t o_cs, o_live, o_ones, o_logit, o_rows, o_scratch, o_counts;
    const size_t total = kmeans_ws_parts(p, &o_cs, &o_live, &o_ones, &o_logit, &o_rows, &o_scratch, &o_counts);
    if (workspace_bytes < total) return fail(TCLIP_ERR_WORKSPACE, "workspace smaller than tclip_soft_kmeans_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(TCLIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int Q = p.n_query, K = p.n_class, T = p.n_batches * p.tasks_per_batch, TK = T * K;
    const size_t TQK = (size_t)T * Q * K;
    float* cs = (float*)(ws + o_cs);
    uint8_t* live = (uint8_t*)(ws + o_live);
    uint8_t* ones = (uint8_t*)(ws + o_ones);
    float* logit0 = (float*)(ws + o_logit);
    hipLaunchKernelGGL(k_copy, dim3(ew_grid(TQK)), dim3(256), 0, st, x_q, u, TQK);          // u = z
    TCLIP_HIP(hipMemsetAsync(ones, 1, (size_t)TK, st));
    hipLaunchKernelGGL(k_fill, dim3(ew_grid(TK)), dim3(256), 0, st, v, 0.0f, (size_t)TK);
    // w_init, s_init: every cluster                                                (em_gaussian_cov.py:146-180)
    hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs, live,
                       (float*)nullptr, (int32_t*)nullptr);
    launch_mstats(st, (const float*)u, x_q, (const float*)cs, (const uint8_t*)ones, (const float*)nullptr, (const float*)nullptr, T, Q, K, w, 0);
    launch_cov_stats(st, (const float*)u, x_q, (const float*)cs, (const uint8_t*)ones, (const float*)w, T, Q, K, s);
    for (int it = 0; it < p.iters; it++) {
        // w_update, s_update: non-empty clusters move, empty ones keep w and s        (:160-193); v of the previous v_update
        hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs,
                           live, it > 0 ? v : (float*)nullptr, (int32_t*)nullptr);
        launch_mstats(st, (const float*)u, x_q, (const float*)cs, (const uint8_t*)live, (const float*)nullptr, (const float*)nullptr, T, Q, K, w, 0);
        launch_cov_stats(st, (const float*)u, x_q, (const float*)cs, (const uint8_t*)live, (const float*)w, T, Q, K, s);
        // u_update: Mahalanobis distances + log-determinants of the clusters that moved, softmax with lambd v / Q   (:106-129)
        dispatch_E<LaunchCovLogitsRows>(K, T, st, (const float*)w, (const float*)s, x_q, (const uint8_t*)(it == 0 ? ones : live), Q, K,
                                        logit0);
        hipLaunchKernelGGL(k_softmax, dim3((T * Q * 16 + 255) / 256), dim3(256), 0, st, (const float*)logit0,
                           (const float*)v, T * Q, Q, K, (float)p.lambd, 0, 0, u, preds);
    }
    hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs, live, v,
                       (int32_t*)nullptr);                                              // the last v_update
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

size_t tclip_hard_kmeans_workspace_bytes(const tclip_problem* p) {
    if (check_problem(p) != TCLIP_OK) return 0;
    size_t a, b, c, d, e, f, g;
    return kmeans_ws_parts(*p, &a, &b, &c, &d, &e, &f, &g) + align_up((size_t)p->n_batches * p->tasks_per_batch * 4);
}

int tclip_hard_kmeans_run(const tclip_problem* pp, const float* x_q, float* u, float* w, int32_t* preds,
                          float* criterions, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    const tclip_problem p = *pp;
    if (!x_q || !u || !w || !preds || !criterions || !workspace) return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (p.n_support != 0) return fail(TCLIP_ERR_ARG, "HARD_KMEANS is a zero-shot method: n_support must be 0");
    size_t o_cs, o_live, o_ones, o_logit, o_rows, o_scratch, o_counts;
    const size_t o_change = kmeans_ws_parts(p, &o_cs, &o_live, &o_ones, &o_logit, &o_rows, &o_scratch, &o_counts);
    if (workspace_bytes < tclip_hard_kmeans_workspace_bytes(pp)) return fail(TCLIP_ERR_WORKSPACE, "workspace smaller than tclip_hard_kmeans_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(TCLIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int Q = p.n_query, K = p.n_class, B = p.n_batches, N = p.tasks_per_batch, T = B * N, TK = T * K;
    const size_t TQK = (size_t)T * Q * K;
    float* cs = (float*)(ws + o_cs);
    uint8_t* live = (uint8_t*)(ws + o_live);
    uint8_t* ones = (uint8_t*)(ws + o_ones);
    float* logit0 = (float*)(ws + o_logit);
    float* change = (float*)(ws + o_change);
    hipLaunchKernelGGL(k_copy, dim3(ew_grid(TQK)), dim3(256), 0, st, x_q, u, TQK);          // u = z
    TCLIP_HIP(hipMemsetAsync(ones, 1, (size_t)TK, st));                                     // every centroid moves every iteration
    for (int it = 0; it < p.iters; it++) {
        // w_update: mean of the members, zero for empty clusters                            (hard_kmeans.py:138-152)
        hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs,
                           live, (float*)nullptr, (int32_t*)nullptr);
        launch_mstats(st, (const float*)u, (const float*)x_q, (const float*)cs, (const uint8_t*)live, (const float*)nullptr, (const float*)nullptr, T, Q, K, w, 0);
        hipLaunchKernelGGL(k_zero_dead_rows, dim3(ew_grid((size_t)TK * K)), dim3(256), 0, st, (const uint8_t*)live, TK, K, w);
        // u_update + hard assignment: softmax of the squared distances, first minimum    (:128-136, :193-195)
        launch_kmeans_logits(T, st, (const float*)w, x_q, (const uint8_t*)ones, Q, K, 1.0f, 1.0f, logit0);
        hipLaunchKernelGGL(k_softmax, dim3((T * Q * 16 + 255) / 256), dim3(256), 0, st, (const float*)logit0,
                           (const float*)nullptr, T * Q, Q, K, 0.0f, 0, 1, logit0, preds);
        // criterion mean_n ||u_old - u||_F, u <- one-hot                                       (:197-199)
        hipLaunchKernelGGL(k_hard_assign, dim3(T), dim3(256), 0, st, (const int32_t*)preds, Q, K, u, change);
        hipLaunchKernelGGL(k_criterion_mean, dim3(B), dim3(64), 0, st, (const float*)change, N, 0, criterions + it, p.iters);
    }
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

// ---- PADDLE: few-shot soft k-means with the class-proportion penalty
static size_t paddle_ws_parts(const tclip_problem& p, size_t* o_sup, size_t* o_cnt, size_t* o_cs, size_t* o_live,
                              size_t* o_logit, size_t* o_rows, size_t* o_scratch_rows, size_t* o_counts) {
    const size_t T = (size_t)p.n_batches * p.tasks_per_batch, K = p.n_class, Q = p.n_query;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return r; };
    *o_sup = take(T * K * K * 4);
    *o_cnt = take(T * K * 4);
    *o_cs = take(T * K * 4);
    *o_live = take(T * K);
    *o_logit = take(T * Q * K * 4);
    *o_rows = take(T * K * 4);
    *o_scratch_rows = take(T * K * 4);
    *o_counts = take(256);
    return o;
}

size_t tclip_paddle_workspace_bytes(const tclip_problem* p) {
    if (check_problem(p) != TCLIP_OK) return 0;
    size_t a, b, c, d, e, f, g, h;
    return paddle_ws_parts(*p, &a, &b, &c, &d, &e, &f, &g, &h);
}

int tclip_paddle_run(const tclip_problem* pp, const float* x_q, const float* x_s, const int64_t* y_s, float lambd,
                     float* u, float* v, float* w, int32_t* preds, void* workspace, size_t workspace_bytes,
                     void* stream) {
    if (int rc = check_problem(pp)) return rc;
    const tclip_problem p = *pp;
    if (!x_q || !x_s || !y_s || !u || !v || !w || !preds || !workspace) return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (p.n_support < 1) return fail(TCLIP_ERR_ARG, "PADDLE is a few-shot method: n_support must be positive");
    size_t o_sup, o_cnt, o_cs, o_live, o_logit, o_rows, o_scratch, o_counts;
    const size_t total = paddle_ws_parts(p, &o_sup, &o_cnt, &o_cs, &o_live, &o_logit, &o_rows, &o_scratch, &o_counts);
    if (workspace_bytes < total) return fail(TCLIP_ERR_WORKSPACE, "workspace smaller than tclip_paddle_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(TCLIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int Q = p.n_query, K = p.n_class, S = p.n_support, T = p.n_batches * p.tasks_per_batch, TK = T * K;
    float* sup = (float*)(ws + o_sup);
    float* cnt = (float*)(ws + o_cnt);
    float* cs = (float*)(ws + o_cs);
    uint8_t* live = (uint8_t*)(ws + o_live);
    float* logit0 = (float*)(ws + o_logit);
    // init (paddle.py:180-197): v = 0, prototypes = class means of the support set; every centroid moves every iteration
    hipLaunchKernelGGL(k_fill, dim3(ew_grid(TK)), dim3(256), 0, st, v, 0.0f, (size_t)TK);
    hipLaunchKernelGGL(k_support_stats, dim3(K, T), dim3(128), (size_t)S * sizeof(int), st, dense_rows(x_s), y_s, S, K, 0, sup, cnt);
    hipLaunchKernelGGL(k_div_rows, dim3(ew_grid((size_t)TK * K)), dim3(256), 0, st, (const float*)sup, (const float*)cnt,
                       (size_t)TK * K, K, w);
    TCLIP_HIP(hipMemsetAsync(live, 1, (size_t)TK, st));
    for (int it = 0; it < p.iters; it++) {
        // u_update (:105-116): softmax_k(-1/2 ||w_k - z_q||^2 + lambd v_k / Q)
        launch_kmeans_logits(T, st, (const float*)w, x_q, (const uint8_t*)live, Q, K, -0.5f, 1.0f, logit0);
        hipLaunchKernelGGL(k_softmax, dim3((T * Q * 16 + 255) / 256), dim3(256), 0, st, (const float*)logit0, (const float*)v,
                           T * Q, Q, K, lambd, 0, 0, u, preds);
        // v_update (:118-124) and w_update (:142-158)
        hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 0, cs, live,
                           v, (int32_t*)nullptr);
        launch_mstats(st, (const float*)u, (const float*)x_q, (const float*)cs, (const uint8_t*)live, (const float*)sup, (const float*)cnt, T, Q, K, w, 1);
    }
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

// ---- BD-CSPN: one pass, no loop
struct BdcspnWs { size_t zs, zq, zqn, mean, eta, sup, cnt, wn, aug, logit, cs, live, dummy, total; };
static BdcspnWs bdcspn_ws(const tclip_problem& p) {
    const size_t T = (size_t)p.n_batches * p.tasks_per_batch, K = p.n_class, Q = p.n_query, S = p.n_support, R = S + Q;
    BdcspnWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return r; };
    w.zs = take(T * S * K * 4);
    w.zq = take(T * Q * K * 4);
    w.zqn = take(T * Q * K * 4);
    w.mean = take(T * K * 4);
    w.eta = take(T * K * 4);
    w.sup = take(T * K * K * 4);
    w.cnt = take(T * K * 4);
    w.wn = take(T * K * K * 4);
    w.aug = take(T * R * K * 4);
    w.logit = take(T * R * K * 4);
    w.cs = take(T * K * 4);
    w.live = take(T * K);
    w.dummy = take(T * R * 4);
    w.total = o;
    return w;
}

size_t tclip_bdcspn_workspace_bytes(const tclip_problem* p) {
    if (check_problem(p) != TCLIP_OK) return 0;
    return bdcspn_ws(*p).total;
}

int tclip_bdcspn_run(const tclip_problem* pp, const float* x_q, const float* x_s, const int64_t* y_s, float temp,
                     int32_t norm_type, float* prototypes, float* u, int32_t* preds, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    const tclip_problem p = *pp;
    if (!x_q || !x_s || !y_s || !prototypes || !u || !preds || !workspace) return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (p.n_support < 1) return fail(TCLIP_ERR_ARG, "BDCSPN is a few-shot method: n_support must be positive");
    if (norm_type < 0 || norm_type > 2) return fail(TCLIP_ERR_ARG, "norm_type must be 0 (UN), 1 (L2N) or 2 (CL2N)");
    const BdcspnWs o = bdcspn_ws(p);
    if (workspace_bytes < o.total) return fail(TCLIP_ERR_WORKSPACE, "workspace smaller than tclip_bdcspn_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(TCLIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int Q = p.n_query, K = p.n_class, S = p.n_support, R = S + Q, T = p.n_batches * p.tasks_per_batch, TK = T * K;
    float* zs = (float*)(ws + o.zs);
    float* zq = (float*)(ws + o.zq);
    float* zqn = (float*)(ws + o.zqn);
    float* mean = (float*)(ws + o.mean);
    float* eta = (float*)(ws + o.eta);
    float* sup = (float*)(ws + o.sup);
    float* cnt = (float*)(ws + o.cnt);
    float* wn = (float*)(ws + o.wn);
    float* aug = (float*)(ws + o.aug);
    float* logit = (float*)(ws + o.logit);
    float* cs = (float*)(ws + o.cs);
    uint8_t* live = (uint8_t*)(ws + o.live);
    int32_t* dummy = (int32_t*)(ws + o.dummy);
    auto rows_grid = [](int n_rows) { return dim3((unsigned)(((size_t)n_rows * 8 + 255) / 256)); };
    auto normalize = [&](const float* x, const float* x2, int R0, int Rr, int mode, const float* mn, const float* sh, float* out) {
        hipLaunchKernelGGL(k_bdcspn_normalize, rows_grid(T * Rr), dim3(256), 0, st, x, x2, R0, Rr, K, mode, mn, sh, T * Rr, out);
    };
    // normalization (bdcspn.py:77-100, :165-166): train_mean = support.mean(1); CL2N / L2N / none
    if (norm_type == 2) hipLaunchKernelGGL(k_col_mean, dim3((TK + 255) / 256), dim3(256), 0, st, x_s, T, S, K, mean);
    normalize(x_s, x_s, S, S, norm_type, (const float*)mean, (const float*)nullptr, zs);
    normalize(x_q, x_q, Q, Q, norm_type, (const float*)mean, (const float*)nullptr, zq);
    // initial prototypes: support class means (:117-120), L2-normalised for get_logits (:50)
    hipLaunchKernelGGL(k_support_stats, dim3(K, T), dim3(128), (size_t)S * sizeof(int), st, dense_rows(zs), y_s, S, K, 0, sup, cnt);
    hipLaunchKernelGGL(k_div_rows, dim3(ew_grid((size_t)TK * K)), dim3(256), 0, st, (const float*)sup, (const float*)cnt,
                       (size_t)TK * K, K, prototypes);
    normalize((const float*)prototypes, (const float*)prototypes, K, K, 1, (const float*)nullptr, (const float*)nullptr, wn);
    // augmented set: support rows, then query rows shifted by eta = mean(support) - mean(query); normalised (:127-131, :51, :137)
    hipLaunchKernelGGL(k_bdcspn_eta, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)zs, (const float*)zq, T, S, Q, K, eta);
    normalize((const float*)zs, (const float*)zq, S, R, 1, (const float*)nullptr, (const float*)eta, aug);
    // soft assignment of the augmented set to the initial prototypes (:133-134)
    TCLIP_HIP(hipMemsetAsync(live, 1, (size_t)TK, st));
    launch_kmeans_logits(T, st, (const float*)wn, (const float*)aug, (const uint8_t*)live, R, K, -0.5f, temp, logit);
    hipLaunchKernelGGL(k_softmax, dim3((T * R * 16 + 255) / 256), dim3(256), 0, st, (const float*)logit, (const float*)nullptr,
                       T * R, R, K, 0.0f, 0, 0, logit, dummy);
    // rectified prototypes = assignment-weighted means of the normalised augmented set (:137-141)
    hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)logit, T, R, K, 0, cs, live,
                       (float*)nullptr, (int32_t*)nullptr);
    launch_mstats(st, (const float*)logit, (const float*)aug, (const float*)cs, (const uint8_t*)live, (const float*)nullptr,
                  (const float*)nullptr, T, R, K, prototypes, 2);
    // prediction (:190-193): softmax(temp * get_logits(prototypes, query)), argmax
    normalize((const float*)prototypes, (const float*)prototypes, K, K, 1, (const float*)nullptr, (const float*)nullptr, wn);
    normalize((const float*)zq, (const float*)zq, Q, Q, 1, (const float*)nullptr, (const float*)nullptr, zqn);
    launch_kmeans_logits(T, st, (const float*)wn, (const float*)zqn, (const uint8_t*)live, Q, K, -0.5f, temp, logit);
    hipLaunchKernelGGL(k_softmax, dim3((T * Q * 16 + 255) / 256), dim3(256), 0, st, (const float*)logit, (const float*)nullptr,
                       T * Q, Q, K, 0.0f, 0, 0, u, preds);
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

int tclip_kl_kmeans_run(const tclip_problem* pp, const float* x_q, float* u, float* w, int32_t* preds,
                        float* criterions, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_problem(pp)) return rc;
    const tclip_problem p = *pp;
    if (!x_q || !u || !w || !preds || !criterions || !workspace) return fail(TCLIP_ERR_ARG, "null pointer argument");
    if (p.n_support != 0) return fail(TCLIP_ERR_ARG, "KL_KMEANS is a zero-shot method: n_support must be 0");
    size_t o_cs, o_live, o_ones, o_logit, o_rows, o_scratch, o_counts;
    const size_t o_change = kmeans_ws_parts(p, &o_cs, &o_live, &o_ones, &o_logit, &o_rows, &o_scratch, &o_counts);
    if (workspace_bytes < tclip_hard_kmeans_workspace_bytes(pp)) return fail(TCLIP_ERR_WORKSPACE, "workspace smaller than tclip_hard_kmeans_workspace_bytes()");
    if (((uintptr_t)workspace & 255) != 0) return fail(TCLIP_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int Q = p.n_query, K = p.n_class, B = p.n_batches, N = p.tasks_per_batch, T = B * N, TK = T * K;
    const size_t TQK = (size_t)T * Q * K;
    float* cs = (float*)(ws + o_cs);
    uint8_t* live = (uint8_t*)(ws + o_live);
    float* divs = (float*)(ws + o_logit);
    float* change = (float*)(ws + o_change);
    hipLaunchKernelGGL(k_copy, dim3(ew_grid(TQK)), dim3(256), 0, st, x_q, u, TQK);          // u = z
    for (int it = 0; it < p.iters; it++) {
        hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)u, T, Q, K, 1, cs,
                           live, (float*)nullptr, (int32_t*)nullptr);
        if (Q == kColsQ && K >= kColsChunk && g_mstats_cols != 0) {
            const int dtiles = (K + 63) / 64;
            int splits = (int)((8192 + (long)T * dtiles - 1) / ((long)T * dtiles));
            if (splits > K / (kColsWaves * kColsChunk)) splits = K / (kColsWaves * kColsChunk);
            if (splits < 1) splits = 1;
            const int rows_per_block = ((K + splits - 1) / splits + kColsChunk - 1) / kColsChunk * kColsChunk;
            const int ksplits = (K + rows_per_block - 1) / rows_per_block;
            hipLaunchKernelGGL(k_kl_centroids_cols75, dim3(task_tile_grid(dtiles, ksplits, T)), dim3(64 * kColsWaves), 0,
                               st, (const float*)u, x_q, (const float*)cs, K, rows_per_block, w, T, dtiles, ksplits);
        } else {
            hipLaunchKernelGGL(k_kl_centroids, dim3((K + 63) / 64, (K + kMstatsRows - 1) / kMstatsRows, T), dim3(64), 0, st,
                               (const float*)u, x_q, (const float*)cs, Q, K, w);
        }
        if (g_kmeans_tile != 0 && K >= 32 && K <= 511 && kmeans_tile_lds_raised((const void*)k_kl_divergences_tile)) {
            const int stride = K | 1;
            hipLaunchKernelGGL(k_kl_divergences_tile, dim3((K + kKmeansTile - 1) / kKmeansTile, T), dim3(kKmeansTileThreads),
                               (size_t)kKmeansTile * stride * sizeof(float), st, (const float*)w, x_q, Q, K, stride, divs);
        } else {
            dispatch_E<LaunchKlDivergences>(K, T, st, (const float*)w, x_q, Q, K, divs);
        }
        hipLaunchKernelGGL(k_argmin_rows, dim3((T * Q + 255) / 256), dim3(256), 0, st, (const float*)divs, T * Q, K, preds);
        hipLaunchKernelGGL(k_hard_assign, dim3(T), dim3(256), 0, st, (const int32_t*)preds, Q, K, u, change);
        hipLaunchKernelGGL(k_criterion_mean, dim3(B), dim3(64), 0, st, (const float*)change, N, 0, criterions + it, p.iters);
    }
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

int tclip_argmax_rows(const float* x, int64_t n_rows, int32_t n_class, int32_t* labels, void* stream) {
    if (!x || !labels || n_rows < 0 || n_class < 1) return fail(TCLIP_ERR_ARG, "bad argument to tclip_argmax_rows");
    if (n_rows == 0) return TCLIP_OK;
    hipLaunchKernelGGL(k_argmax_rows, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n_rows,
                       n_class, labels);
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

int tclip_probability_features(const float* visual, const float* text, int64_t n_rows, int32_t dim, int32_t n_class,
                               float temperature, float* out, void* stream) {
    if (!visual || !text || !out || n_rows < 0 || dim < 1 || n_class < 1) return fail(TCLIP_ERR_ARG, "bad argument to tclip_probability_features");
    if (n_rows == 0) return TCLIP_OK;
    const size_t lds = ((size_t)dim + (size_t)n_class) * sizeof(float);
    if (lds > 60000 || n_rows > 0x7fffffff) return fail(TCLIP_ERR_ARG, "dim + n_class must be <= 15000");
    hipLaunchKernelGGL(k_probability_features, dim3((unsigned)n_rows), dim3(256), lds, (hipStream_t)stream, visual, text,
                       dim, n_class, temperature, out);
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

int tclip_debug_set_rowset_min_rows(int32_t rows) {
    g_rowset_min_rows = rows;
    return TCLIP_OK;
}

int tclip_debug_set_kmeans_tile(int32_t mode) {
    g_kmeans_tile = mode;
    g_mstats_cols = mode;
    g_mstats_tile = mode < 0 ? TCLIP_MSTATS_TILE : mode;
    return TCLIP_OK;
}

int tclip_debug_set_fixed_k_kernels(int32_t on) {
    g_fixed_k_kernels = on;
    return TCLIP_OK;
}

int tclip_debug_set_mm_split(int32_t mode) {
    g_mm_split = mode;
    return TCLIP_OK;
}

int tclip_debug_set_split_keep_placement(int32_t on) {
    g_split_keep_placement = on != 0;
    return TCLIP_OK;
}

int tclip_debug_set_dead_head(int32_t iterations) {
    // the early probe needs its last snapshot (32 iterations in) to lie at or before the first checkpoint (iteration 50)
    if (iterations > 18) return fail(TCLIP_ERR_ARG, "tclip_debug_set_dead_head: at most 18 iterations");
    g_dead_head = iterations < 0 ? TCLIP_DEAD_HEAD : iterations;
    return TCLIP_OK;
}
int tclip_debug_set_probe_chunks(int32_t chunks) {
    g_probe_chunks = chunks < 0 ? TCLIP_PROBE_CHUNKS : chunks;
    return TCLIP_OK;
}

int tclip_profile_enable(int on) {
    if (on && !g_prof.counter) {
        TCLIP_HIP(hipMalloc((void**)&g_prof.counter, 4 * sizeof(unsigned long long)));
        TCLIP_HIP(hipMemset(g_prof.counter, 0, 4 * sizeof(unsigned long long)));
    }
    g_prof.on = on != 0;
    return TCLIP_OK;
}

// length of the union of intervals
static double union_ms(std::vector<std::pair<float, float>>& iv) {
    std::sort(iv.begin(), iv.end());
    double busy = 0.0, cur_lo = 0.0, cur_hi = -1.0;
    for (auto& x : iv) {
        if (cur_hi < cur_lo || x.first > cur_hi) {
            if (cur_hi >= cur_lo) busy += cur_hi - cur_lo;
            cur_lo = x.first;
            cur_hi = x.second;
        } else if (x.second > cur_hi) {
            cur_hi = x.second;
        }
    }
    if (cur_hi >= cur_lo) busy += cur_hi - cur_lo;
    return busy;
}
int tclip_profile_collect(double* mm_busy_ms, double* mm_launch_ms_sum, int64_t* mm_launches,
                          int64_t* element_updates) {
    TCLIP_HIP(hipDeviceSynchronize());
    // launches of different batch groups overlap on their streams: report both the sum of the
    // individual launch durations and the length of the union of the [start, end] intervals
    std::vector<std::pair<float, float>> iv, ivk[2];
    double sum = 0.0;
    for (int k = 0; k < 2; k++) { g_prof.last_sum[k] = 0.0; g_prof.last_launches[k] = 0; }
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float t0 = 0.f, t1 = 0.f;
        TCLIP_HIP(hipEventElapsedTime(&t0, g_prof.ev[0], g_prof.ev[i]));
        TCLIP_HIP(hipEventElapsedTime(&t1, g_prof.ev[0], g_prof.ev[i + 1]));
        iv.emplace_back(t0, t1);
        sum += (double)t1 - (double)t0;
        const int k = i / 2 < g_prof.kind.size() ? g_prof.kind[i / 2] : 0;
        ivk[k].emplace_back(t0, t1);
        g_prof.last_sum[k] += (double)t1 - (double)t0;
        g_prof.last_launches[k]++;
    }
    const double busy = union_ms(iv);
    for (int k = 0; k < 2; k++) g_prof.last_busy[k] = union_ms(ivk[k]);
    if (mm_busy_ms) *mm_busy_ms = busy;
    if (mm_launch_ms_sum) *mm_launch_ms_sum = sum;
    if (mm_launches) *mm_launches = (int64_t)(g_prof.used / 2);
    unsigned long long c[4] = {0, 0, 0, 0};
    if (g_prof.counter) {
        TCLIP_HIP(hipMemcpy(c, g_prof.counter, sizeof c, hipMemcpyDeviceToHost));
        TCLIP_HIP(hipMemset(g_prof.counter, 0, sizeof c));
    }
    g_prof.last_updates[0] = (int64_t)c[0];
    g_prof.last_updates[1] = (int64_t)c[1];
    g_prof.last_split_iterations = (int64_t)c[2];
    g_prof.last_split_sorts = (int64_t)c[3];
    if (element_updates) *element_updates = (int64_t)(c[0] + c[1]);
    g_prof.used = 0;
    return TCLIP_OK;
}
int tclip_profile_last_kernels(double* busy_ms, double* launch_ms_sum, int64_t* launches, int64_t* element_updates) {
    for (int k = 0; k < 2; k++) {
        if (busy_ms) busy_ms[k] = g_prof.last_busy[k];
        if (launch_ms_sum) launch_ms_sum[k] = g_prof.last_sum[k];
        if (launches) launches[k] = g_prof.last_launches[k];
        if (element_updates) element_updates[k] = g_prof.last_updates[k];
    }
    return TCLIP_OK;
}

#ifdef TCLIP_COUNT_SMALL
int tclip_debug_small_count(uint64_t* out) {        // reads and clears g_small_count (only in builds with -DTCLIP_COUNT_SMALL)
    unsigned long long h[4];
    TCLIP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(tclip::g_small_count), sizeof h));
    for (int i = 0; i < 4; i++) { out[i] = h[i]; h[i] = 0; }
    TCLIP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(tclip::g_small_count), h, sizeof h));
    return TCLIP_OK;
}
#endif
#ifdef TCLIP_PHASE_CLOCK
int tclip_debug_phase_clock(uint64_t* out) {        // reads and clears g_phase_clock (only in builds with -DTCLIP_PHASE_CLOCK)
    TCLIP_HIP(hipDeviceSynchronize());
    unsigned long long h[8];
    TCLIP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(tclip::g_phase_clock), sizeof h));
    for (int i = 0; i < 8; i++) out[i] = h[i];
    memset(h, 0, sizeof h);
    TCLIP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(tclip::g_phase_clock), h, sizeof h));
    return TCLIP_OK;
}
#endif
int tclip_profile_last_split_sorts(int64_t* wave_iterations, int64_t* sorts) {
    if (wave_iterations) *wave_iterations = g_prof.last_split_iterations;
    if (sorts) *sorts = g_prof.last_split_sorts;
    return TCLIP_OK;
}

int tclip_selftest_primitives(uint64_t* mismatches) {
    if (!mismatches) return fail(TCLIP_ERR_ARG, "null pointer");
    unsigned long long* d = nullptr;
    TCLIP_HIP(hipMalloc((void**)&d, 19 * sizeof(unsigned long long)));
    TCLIP_HIP(hipMemset(d, 0, 19 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_selftest, dim3(2048), dim3(256), 0, 0, d);
    TCLIP_HIP(hipDeviceSynchronize());
    unsigned long long h[19];
    TCLIP_HIP(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    memcpy(mismatches, h, 14 * sizeof(unsigned long long));
    if (h[4]) snprintf(g_err, sizeof g_err, "MM update mismatch example: a=%08llx y=%08llx psi_s=%08llx fast=%08llx generic=%08llx",
                       h[14], h[15], h[16], h[17], h[18]);
    TCLIP_HIP(hipFree(d));
    return TCLIP_OK;
}

size_t tclip_prototype_workspace_bytes(int32_t T, int32_t Q, int32_t K) {
    if (T < 1 || Q < 1 || K < 2) return 0;
    return align_up((size_t)T * Q * K * 4) + align_up((size_t)T * K * K * 4) + align_up((size_t)T * K * 4) +
           align_up((size_t)T * K);
}

int tclip_cluster_prototypes(int32_t T, int32_t Q, int32_t K, const float* x_q, const int32_t* preds,
                             int32_t* n_clusters, int32_t* cluster_ids, float* prototypes, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (T < 1 || Q < 1 || K < 2 || K > 1024 || !x_q || !preds || !n_clusters || !cluster_ids || !prototypes || !workspace)
        return fail(TCLIP_ERR_ARG, "bad argument to tclip_cluster_prototypes");
    if (workspace_bytes < tclip_prototype_workspace_bytes(T, Q, K)) return fail(TCLIP_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* hot = (float*)ws;
    float* dense = (float*)(ws + align_up((size_t)T * Q * K * 4));
    float* cs = (float*)((char*)dense + align_up((size_t)T * K * K * 4));
    uint8_t* live = (uint8_t*)((char*)cs + align_up((size_t)T * K * 4));
    const int TK = T * K;
    const int Cmax = Q < K ? Q : K;
    hipLaunchKernelGGL(k_one_hot, dim3(ew_grid((size_t)T * Q * K)), dim3(256), 0, st, preds, (size_t)T * Q, K, hot);
    // cluster sizes of the one-hot predictions; "live" = non-empty cluster
    hipLaunchKernelGGL(k_cluster_sizes, dim3((TK + 255) / 256), dim3(256), 0, st, (const float*)hot, T, Q, K, 1, cs, live,
                       (float*)nullptr, (int32_t*)nullptr);
    launch_mstats(st, (const float*)hot, (const float*)x_q, (const float*)cs, (const uint8_t*)live, (const float*)nullptr, (const float*)nullptr, T, Q, K, dense, 0);
    hipLaunchKernelGGL(k_gather_prototypes, dim3(T), dim3(256), 0, st, preds, (const float*)dense, Q, K, Cmax, n_clusters,
                       cluster_ids, prototypes);
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

int tclip_check_task_indices(const int64_t* idx, int64_t n_idx, int64_t n_rows, const int32_t* cols, int64_t n_cols,
                             int32_t n_class, void* stream) {
    if ((n_idx > 0 && !idx) || (n_cols > 0 && !cols) || n_idx < 0 || n_cols < 0 || n_rows < 1 || n_class < 1)
        return fail(TCLIP_ERR_ARG, "bad argument to tclip_check_task_indices");
    static thread_local int32_t* flag = nullptr;           // one word of device memory per calling thread, kept
    static thread_local int flag_device = -1;
    int dev = 0;
    TCLIP_HIP(hipGetDevice(&dev));
    if (flag && flag_device != dev) { (void)hipFree(flag); flag = nullptr; }
    if (!flag) { TCLIP_HIP(hipMalloc((void**)&flag, sizeof(int32_t))); flag_device = dev; }
    hipStream_t st = (hipStream_t)stream;
    TCLIP_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
    if (n_idx > 0)
        hipLaunchKernelGGL(k_check_range<int64_t>, dim3(ew_grid((size_t)n_idx)), dim3(256), 0, st, idx, (size_t)n_idx, (int64_t)n_rows, flag, 1);
    if (n_cols > 0)
        hipLaunchKernelGGL(k_check_range<int32_t>, dim3(ew_grid((size_t)n_cols)), dim3(256), 0, st, cols, (size_t)n_cols, (int64_t)n_class, flag, 2);
    TCLIP_HIP(hipGetLastError());
    int32_t h = 0;
    TCLIP_HIP(hipMemcpyAsync(&h, flag, sizeof h, hipMemcpyDeviceToHost, st));
    TCLIP_HIP(hipStreamSynchronize(st));
    if (h & 1) return fail(TCLIP_ERR_INDEX, "index out of range for the feature table");
    if (h & 2) return fail(TCLIP_ERR_INDEX, "column index out of range");
    return TCLIP_OK;
}

int tclip_gather_rows(const float* table, int64_t n_rows, int32_t K, const int64_t* idx, int64_t n_out, float* out,
                      void* stream) {
    if (!table || !idx || !out || n_rows < 1 || K < 1 || n_out < 0) return fail(TCLIP_ERR_ARG, "bad argument to tclip_gather_rows");
    if (n_out == 0) return TCLIP_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n_out), dim3(128), 0, (hipStream_t)stream, table, n_rows, K, idx, n_out, out);
    TCLIP_HIP(hipGetLastError());
    return TCLIP_OK;
}

}  // extern "C"

#include "tclip_tim.inc"
#include "tclip_lshot.inc"
#endif  // TCLIP_ISA_ONLY
