// Device-side building blocks shared by the kernels in tclip_kernels.hip: reductions that follow
// the association order of torch's CPU kernels (the reference's arithmetic), laid out on 32-lane
// half-wavefront groups.
//
// Why the order matters: the reference runs on torch CPU fp32; its sums are
//   * last-dim sums (alpha.sum(-1), lgamma(alpha).sum(-1), the (alpha-1)*log z contraction):
//     aten SumKernel.cpp vectorized_inner_sum - 8-float vectors (sum_stub has no AVX-512 variant),
//     4 interleaved vector accumulators, 16-element cascade blocks, then a serial pass over the
//     8 lanes;
//   * sums over the query dimension (u.sum(1), the u^T log z statistics): vectorized_outer_sum -
//     per output column a 16-element cascade, except the last (#columns mod 32) columns, which go
//     through the 4-way interleaved row sum (fewer than 8 columns: scalar_outer_sum, the same in
//     groups of 4 columns);
//   * softmax denominators: vec::reduce_all on 16-float vectors + 8/4/2/1 butterfly.
// oracle/tclip_oracle.cpp restates the same orders on the CPU and tests/test_oracle_sums.py pins
// them bit-for-bit against torch.  Element d of a K-vector lives in register e = d / 32 of lane
// d % 32, which makes (lane / 8, lane % 8, e) exactly torch's (accumulator, vector lane, step).
#pragma once
#include <hip/hip_runtime.h>

#include "tclip_math.h"
#include "tclip_rsqrt14_table_dev.h"

namespace tclip {

constexpr int kGroup = 32;          // lanes per row
constexpr float kEpsF = 1e-15f;

// v from the lane N places up inside the 16-lane DPP row (row_shl:N); lanes whose source falls outside the row get 0.
// One VALU operand modifier instead of a ds_bpermute through the LDS crossbar.
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v) {
    static_assert(N >= 1 && N <= 15, "row_shl");
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
// fin + p0[0] + p0[1] + ... + p0[7] in that order, p0[t] being lane t's value; the result is valid in lane 0 of every
// 16-lane row (callers: the first lane of a row's lane group, which starts a DPP row in every layout).
__device__ __forceinline__ float ordered_sum8_lane0(float fin, float p0) {
    fin += p0;
    fin += dpp_row_shl<1>(p0);
    fin += dpp_row_shl<2>(p0);
    fin += dpp_row_shl<3>(p0);
    fin += dpp_row_shl<4>(p0);
    fin += dpp_row_shl<5>(p0);
    fin += dpp_row_shl<6>(p0);
    fin += dpp_row_shl<7>(p0);
    return fin;
}

__device__ __forceinline__ float group_shfl(float v, int src_lane) { return __shfl(v, src_lane, kGroup); }
__device__ __forceinline__ double group_shfl_xor(double v, int m) { return __shfl_xor(v, m, kGroup); }

// x[e] holds element 32e + lane (0 where 32e + lane >= K).  Returns torch's x.sum(-1), the same
// value in all 32 lanes of the group.
// kLane0: the caller needs the sum in lane 0 of the group only (the last eight additions then take their operands
// through DPP instead of eight LDS-crossbar shuffles).
// kFullRegs: the caller guarantees K >= 32 kFullRegs (registers below that index lie inside the row for every K the
// instantiation is used for), which frees their additions from the per-register tests.
template <int E, bool kLane0 = false, int kFullRegs = 0>
__device__ __forceinline__ float group_sum_torch(const float (&x)[E], int K, int lane) {
    if (kFullRegs == 0 && K < 8) {  // scalar_inner_sum: 4 interleaved scalar accumulators
        const int size_ilp = K >> 2;
        float fin = size_ilp ? group_shfl(x[0], 0) : 0.0f;
        for (int i = size_ilp * 4; i < K; i++) fin += group_shfl(x[0], i);
        if (size_ilp) {
            fin += group_shfl(x[0], 1);
            fin += group_shfl(x[0], 2);
            fin += group_shfl(x[0], 3);
        }
        return fin;
    }
    const int vec_size = K >> 3, size_ilp = vec_size >> 2;
    const int j = lane & 7;
    float a0 = 0.0f, a1 = 0.0f, ragged = 0.0f;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (e == 16 && (kFullRegs >= 16 || size_ilp >= 16)) { a1 += a0; a0 = 0.0f; }
        if (e < kFullRegs) {
            a0 += x[e];
        } else {
            if (e < size_ilp) a0 += x[e];
            if (e == size_ilp) ragged = x[e];
        }
    }
    if (E == 32 && size_ilp >= 32) { a1 += a0; a0 = 0.0f; }
    const float pm = a0 + a1;                       // per-(accumulator, lane) partial
    // whole vectors beyond the 4-way interleaved part join accumulator 0 in vector order
    const int nleft = vec_size - 4 * size_ilp;      // 0..3
    float p0 = pm;
    const float l1 = group_shfl(ragged, 8 + j), l2 = group_shfl(ragged, 16 + j);
    if (nleft >= 1) p0 += ragged;
    if (nleft >= 2) p0 += l1;
    if (nleft >= 3) p0 += l2;
    p0 += group_shfl(pm, 8 + j);
    p0 += group_shfl(pm, 16 + j);
    p0 += group_shfl(pm, 24 + j);                   // valid in lanes 0..7
    // scalar tail (K mod 8 elements) first, then the 8 vector lanes in order
    const int ntail = K - 8 * vec_size, tail_base = 8 * (vec_size & 3);
    float fin = 0.0f;
    for (int t = 0; t < ntail; t++) fin += group_shfl(ragged, tail_base + t);    // wave-uniform trip count (0..7)
    if (kLane0) return ordered_sum8_lane0(fin, p0);
#pragma unroll
    for (int t = 0; t < 8; t++) fin += group_shfl(p0, t);
    return fin;
}

// The same sum with a row spread over G = 8 or 16 lanes (short rows: more rows per wavefront, fuller lanes).
// Element d lives in register d / G of lane d % G.  With H = G / 8 lanes-of-eight per group, lane (h, j) =
// 8 h + j holds the 8-float vector v = e H + h in register e; torch's accumulator r = v % 4 and step m = v / 4
// are r = (e % P) H + h, m = e / P with P = 4 / H registers per step, so a lane keeps P partial sums.
// kSure: the caller guarantees that registers below it hold steps of the 4-way interleaved part (e / P < K / 32) for every K
// the instantiation is used for, which frees their additions from the per-register tests (wave-uniform, but K is a run-time
// value: the compiler kept 2 E lane masks in scalar registers, spilled them, and read them back every iteration).
template <int E, int G, bool kLane0 = false, int kSure = 0>
__device__ __forceinline__ float group_sum_torch_g(const float (&x)[E], int K, int lane) {
    static_assert(G == 8 || G == 16, "lanes per row");
    constexpr int H = G / 8, P = 4 / H;
    static_assert((E + P - 1) / P <= 31, "rows this long need the second cascade dump of the 32-lane form");
    auto shfl = [](float v, int src) { return __shfl(v, src, G); };
    if (K < 8) {  // scalar_inner_sum: 4 interleaved scalar accumulators; the whole row sits in register 0
        const int size_ilp = K >> 2;
        float fin = size_ilp ? shfl(x[0], 0) : 0.0f;
        for (int i = size_ilp * 4; i < K; i++) fin += shfl(x[0], i);
        if (size_ilp) {
            fin += shfl(x[0], 1);
            fin += shfl(x[0], 2);
            fin += shfl(x[0], 3);
        }
        return fin;
    }
    const int vec_size = K >> 3, size_ilp = vec_size >> 2;
    const int j = lane & 7;
    float a0[P], a1[P], rag[P];
#pragma unroll
    for (int p = 0; p < P; p++) a0[p] = a1[p] = rag[p] = 0.0f;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int p = e % P, m = e / P;                     // compile-time
        if (p == 0 && m == 16 && size_ilp >= 16) {          // 16-step cascade dump (rows of 512+ elements only)
#pragma unroll
            for (int q = 0; q < P; q++) { a1[q] += a0[q]; a0[q] = 0.0f; }
        }
        if (e < kSure) {
            a0[p] += x[e];
        } else {
            if (m < size_ilp) a0[p] += x[e];
            if (m == size_ilp) rag[p] = x[e];
        }
    }
    float pm[P];
#pragma unroll
    for (int p = 0; p < P; p++) pm[p] = a0[p] + a1[p];
    // accumulator 0 first, then the whole vectors beyond the 4-way part (vector 4 size_ilp + i sits in
    // register-of-step i / H of lane-of-eight i % H), then accumulators 1, 2, 3
    const int nleft = vec_size - 4 * size_ilp;              // 0..3
    float p0 = pm[0];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float v = shfl(rag[i / H], 8 * (i % H) + j);
        if (i < nleft) p0 += v;
    }
#pragma unroll
    for (int r = 1; r < 4; r++) {                                            // valid in lanes 0..7
        if (r % H == 0) p0 += pm[r / H];                                     // own lane's other accumulator
        else if (kLane0 && G == 16) p0 += dpp_row_shl<8>(pm[r / H]);         // lane 8 + j of the same DPP row
        else p0 += shfl(pm[r / H], 8 * (r % H) + j);
    }
    // scalar tail (K mod 8 elements, in the partial vector vec_size) first, then the 8 vector lanes in order
    const int ntail = K - 8 * vec_size, tv = vec_size & 3;   // the partial vector is vector tv of the ragged step
    float fin = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i == tv)                                         // wave-uniform
            for (int t = 0; t < ntail; t++) fin += shfl(rag[i / H], 8 * (i % H) + t);
    }
    if (kLane0) return ordered_sum8_lane0(fin, p0);
#pragma unroll
    for (int t = 0; t < 8; t++) fin += shfl(p0, t);
    return fin;
}

// A row of 512 < K <= 1024 elements on a whole wavefront: lane 32 q + l (q = 0, 1) holds elements
// 512 q + 32 e + l in register e (E = 16).  torch's 4-way interleaved accumulators dump their running sums
// every 16 steps, so accumulator (l / 8, l % 8) is RN(sum of steps 0..15) + RN(sum of steps 16..) - the two
// halves of the wavefront build the two terms independently.  Valid in all 64 lanes.
// kSure: registers below it are steps of the interleaved part in BOTH halves for every K of the instantiation
// (16 + e < K / 32), see group_sum_torch_g.
template <int E, bool kLane0 = false, int kSure = 0>
__device__ __forceinline__ float group_sum_torch_64(const float (&x)[E], int K, int lane) {
    static_assert(E == 16, "two halves of 16 steps");
    const int vec_size = K >> 3, size_ilp = vec_size >> 2;       // 16 <= size_ilp <= 32
    const int q = lane >> 5, l = lane & 31, j = l & 7;
    float acc = 0.0f, ragged = 0.0f;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int m = e + 16 * q;
        if (e < kSure) {
            acc += x[e];
        } else {
            if (m < size_ilp) acc += x[e];
            if (m == size_ilp) ragged = x[e];
        }
    }
    const float pm = acc + __shfl_xor(acc, 32, 64);               // a0 + a1 (either order: one addition)
    const int rq = (size_ilp >> 4) * 32;                           // the half that holds the ragged step
    const int nleft = vec_size - 4 * size_ilp;
    float p0 = pm;
    const float l0 = __shfl(ragged, rq + j, 64), l1 = __shfl(ragged, rq + 8 + j, 64), l2 = __shfl(ragged, rq + 16 + j, 64);
    if (nleft >= 1) p0 += l0;
    if (nleft >= 2) p0 += l1;
    if (nleft >= 3) p0 += l2;
    p0 += __shfl(pm, 8 + j, 64);
    p0 += __shfl(pm, 16 + j, 64);
    p0 += __shfl(pm, 24 + j, 64);                                 // valid in lanes 0..7
    const int ntail = K - 8 * vec_size, tail_base = rq + 8 * (vec_size & 3);
    float fin = 0.0f;
    for (int t = 0; t < ntail; t++) fin += __shfl(ragged, tail_base + t, 64);
    if (kLane0) return ordered_sum8_lane0(fin, p0);
#pragma unroll
    for (int t = 0; t < 8; t++) fin += __shfl(p0, t, 64);
    return fin;
}

template <int G>
__device__ __forceinline__ double group_sum_f64_g(double v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, G);
    return v;
}

// fp64 sum over the 32 lanes of a group (fixed butterfly order, same value in every lane).
__device__ __forceinline__ double group_sum_f64(double v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += group_shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ int dev_ceil_log2(int n) {
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

// torch's multi_row_sum cascade over n values get(0..n-1).
template <typename F>
__device__ __forceinline__ float dsum_cascade(int n, F get) {
    const int cl = dev_ceil_log2(n) / 4;
    const int level_power = cl > 4 ? cl : 4;
    const int step = 1 << level_power, mask = step - 1;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
    int i = 0;
    for (; i + step <= n;) {
        for (int jj = 0; jj < step; ++jj, ++i) acc0 += get(i);
        acc1 += acc0; acc0 = 0.0f;
        if ((i & (mask << level_power)) == 0) {
            acc2 += acc1; acc1 = 0.0f;
            if ((i & (mask << (2 * level_power))) == 0) { acc3 += acc2; acc2 = 0.0f; }
        }
    }
    for (; i < n; ++i) acc0 += get(i);
    acc0 += acc1;
    acc0 += acc2;
    acc0 += acc3;
    return acc0;
}

// torch's row_sum: 4 interleaved cascades, leftovers into partial 0.
template <typename F>
__device__ __forceinline__ float dsum_ilp4(int n, F get) {
    const int size_ilp = n >> 2;
    float p0 = dsum_cascade(size_ilp, [&](int m) { return get(4 * m + 0); });
    const float p1 = dsum_cascade(size_ilp, [&](int m) { return get(4 * m + 1); });
    const float p2 = dsum_cascade(size_ilp, [&](int m) { return get(4 * m + 2); });
    const float p3 = dsum_cascade(size_ilp, [&](int m) { return get(4 * m + 3); });
    for (int i = size_ilp * 4; i < n; i++) p0 += get(i);
    p0 += p1;
    p0 += p2;
    p0 += p3;
    return p0;
}

// torch's sum over a strided dimension for output column `col` of `ncols` contiguous columns:
// vectorized_outer_sum sends the leading multiple of 32 columns through the cascade and the rest
// through the 4-way row sum; with fewer than 8 columns scalar_outer_sum does the same in groups of 4.
__device__ __forceinline__ bool outer_column_is_cascade(long col, long ncols) {
    return ncols >= 8 ? col < (ncols / 32) * 32 : col < (ncols / 4) * 4;
}
template <typename F>
__device__ __forceinline__ float dsum_outer(int n, long col, long ncols, F get) {
    return outer_column_is_cascade(col, ncols) ? dsum_cascade(n, get) : dsum_ilp4(n, get);
}

// torch's contiguous last-dim sum evaluated serially by one thread (small n only).
template <typename F>
__device__ float dsum_inner_serial(int n, F get) {
    if (n < 8) return dsum_ilp4(n, get);
    const int vec_size = n >> 3, size_ilp = vec_size >> 2;
    float p0[8];
    for (int jj = 0; jj < 8; jj++) {
        float p[4];
        for (int r = 0; r < 4; r++) p[r] = dsum_cascade(size_ilp, [&](int m) { return get(8 * (4 * m + r) + jj); });
        for (int vv = size_ilp * 4; vv < vec_size; vv++) p[0] += get(8 * vv + jj);
        p[0] += p[1];
        p[0] += p[2];
        p[0] += p[3];
        p0[jj] = p[0];
    }
    float fin = 0.0f;
    for (int k = vec_size * 8; k < n; k++) fin += get(k);
    for (int jj = 0; jj < 8; jj++) fin += p0[jj];
    return fin;
}

// The estimate torch.sqrt's Heron step starts from (sqrt_torch_inrange_f32: y = VRSQRT14PS(x), s = x y,
// root = fma(fma(-s, s, x), y / 2, s)), from x's bits b (positive normal) and the entry of the derived table
// (tclip_rsqrt14_table_dev.h: indexed by bit 23 and the top 15 mantissa bits of b as they stand, the estimate's mantissa
// already in place under the exponent of 2^63).  x = m 4^k, m in [1, 4); with E the biased exponent k = ((E + 1) >> 1) - 64
// whatever E's parity, so the scaling is one subtraction of bits 24..31 of b + 2^23, moved down one place: two instructions
// for the index and three here, where the 16-bit table took three and seven (integer and select instructions issue at 16
// lanes per clock on gfx950, fp32 arithmetic at 32: the ten were a tenth of an update's issue time).
// NOT VRSQRT14PS at the exact powers of 4: the instruction returns 2^-k there, this returns the table's value for the
// mantissas just above (2^-k (1 - 3 2^-17)).  The square root does not see the difference: with y = 2^-k (1 + e) the product
// s = 2^k (1 + e) is exact, fma(-s, s, x) = -4^k (2 e + e^2) to 2^-24, and the result is RN(2^k (1 - 3 e^2 / 2 + ..)) = 2^k for
// every |e| <= 2^-14 - what the exact estimate gives.  oracle/mathcheck.cpp (mc_sqrt_without_pow4) and k_selftest check
// the whole root, on every power of two of the range and on every float of [1, 4).
static __device__ const uint32_t kRsqrt14DevTab[65536] = {TCLIP_RSQRT14_DEV_TABLE_VALUES};

__device__ __forceinline__ uint32_t rsqrt14_entry(uint32_t b) { return kRsqrt14DevTab[(b >> 8) & 0xffffu]; }
__device__ __forceinline__ float rsqrt14_from_entry(uint32_t b, uint32_t t) {
    return bits_f32(t - (((b + 0x00800000u) >> 1) & 0x7f800000u));
}

__device__ __forceinline__ float sqrt_torch_inrange_dev(float x) {     // sqrt_torch_inrange_f32 on the derived table
    const uint32_t b = f32_bits(x);
    const float y = rsqrt14_from_entry(b, rsqrt14_entry(b));
    const float s = x * y;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), 0.5f * y, s);
}

// One majorize-minimize update of a single Dirichlet parameter (em_dirichlet.py:153-167), given
// psi1 = digamma(a+1) and lg1 = lgamma(a+1): the operation order and the roundings (no
// contraction) of the torch CPU ops.  Branch-free; valid for mm_fast_domain(a), finite y, psi_s.
__device__ __forceinline__ float mm_update_algebra(float a, float y, float psi_s, float psi1, float lg1) {
    const float t = (0.0f - lg1) + psi1 * a;
    const float big = __builtin_fabsf(div_rn_inrange_f32(2.0f * t, a * a));
    const float curv = (a > 1e-11f) ? big : 1.6449340668482264f;   // polygamma(1, 1)
    float b = (psi1 - psi_s) - curv * a;
    b = b - y;
    const float delta = b * b + 4.0f * curv;
    const float nume = -b + sqrt_torch_inrange_dev(delta), deno = 2.0f * curv;
    // curvature exactly 0 (total cancellation in t): IEEE x/0 = +-inf or nan, as the reference gets
    return deno == 0.0f ? nume * __builtin_inff() : div_rn_inrange_f32(nume, deno);
}

__device__ __forceinline__ float mm_update(float a, float y, float psi_s, const LogTabEntry* tab) {
    float psi1, lg1;
    digamma_lgamma_xp1(a, tab, psi1, lg1);
    return mm_update_algebra(a, y, psi_s, psi1, lg1);
}

// The same update with the generic routines and the compiler's IEEE operators: any input.
__device__ __noinline__ float mm_update_generic(float a, float y, float psi_s) {
    const float x1 = a + 1.0f;
    const float psi1 = digamma_f32(x1);
    const float lg1 = lgamma_f32(x1);
    float curv;
    if (a > 1e-11f) {
        const float t = (0.0f - lg1) + psi1 * a;
        curv = __builtin_fabsf((2.0f * t) / (a * a));
    } else {
        curv = 1.6449340668482264f;
    }
    float b = (psi1 - psi_s) - curv * a;
    b = b - y;
    const float delta = b * b + 4.0f * curv;
    return (-b + sqrt_torch_f32(delta)) / (2.0f * curv);
}

// The 16-entry log table in LDS (one copy per workgroup).
__device__ __forceinline__ void load_log_table(LogTabEntry* lds_tab) {
    if (threadIdx.x < 16) lds_tab[threadIdx.x] = kLogTab[threadIdx.x];
    __syncthreads();
}

}  // namespace tclip
