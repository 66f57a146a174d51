// Issue-rate microbenchmark for gfx950 VALU instruction classes used by the MM kernel:
// scalar fp32 fma, packed fp32 fma, v_rcp_f32, fp64 fma, v_cndmask.  Prints lane-ops/s per class.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kChains = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a[kChains]; f2 p[kChains]; double d[kChains];
    for (int i = 0; i < kChains; i++) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1}; d[i] = a[i]; }
    const float m = 0.999f, c = 0.001f;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < kChains; i++) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], f2{m, m}, f2{c, c});
            if (MODE == 2) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (MODE == 3) d[i] = __builtin_fma(d[i], (double)m, (double)c);
            if (MODE == 4) a[i] = a[i] > 0.5f ? a[(i + 1) % kChains] : c;
            if (MODE == 5) p[i] = p[i] + f2{c, c};
            if (MODE == 6) p[i] = p[i] * f2{m, m};
            if (MODE == 7) {     // one rcp and four INDEPENDENT fmas (other chains' registers): serialised = 9.0 + 4 x 2.7, overlapped = max
                a[i] = __builtin_amdgcn_rcpf(a[i]);
                d[i] = d[i];
                p[i].x = __builtin_fmaf(p[i].x, m, c); p[i].y = __builtin_fmaf(p[i].y, m, c);
                p[(i + 1) % kChains].x = __builtin_fmaf(p[(i + 1) % kChains].x, m, c); p[(i + 1) % kChains].y = __builtin_fmaf(p[(i + 1) % kChains].y, m, c);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < kChains; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * kIters * kChains;
    printf("%-10s %8.3f ms  %.3e wave-instr/s  -> %.2f cycles per wave-instr per SIMD (2.4 GHz, 1024 SIMDs)\n", name, ms,
           wave_instr / (ms * 1e-3), 2.4e9 * 1024 / (wave_instr / (ms * 1e-3)));
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("fma_f32", out); run<1>("pk_fma_f32", out); run<5>("pk_add_f32", out); run<6>("pk_mul_f32", out);
    run<2>("rcp_f32", out); run<3>("fma_f64", out); run<4>("cndmask", out);
    run<7>("rcp+4fma", out);      // counted as ONE instruction group per count: compare with 9.0 + 4 x 2.7 = 19.7 (no overlap)
    return 0;
}
