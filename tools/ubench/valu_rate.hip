// Microbenchmark: VALU issue cost of a wave64 f32 instruction on gfx950 as a function of waves per SIMD.
// Each lane runs ITER x 8 independent v_fma_f32 chains; prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ void k(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0001f, c = 0.5f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            }
        } else if (KIND == 1) {   // f64 fma
#pragma unroll
            for (int u = 0; u < 16; ++u) { d0 = __builtin_fma(d0, 1.0001, 0.5); d1 = __builtin_fma(d1, 1.0001, 0.5); d2 = __builtin_fma(d2, 1.0001, 0.5); d3 = __builtin_fma(d3, 1.0001, 0.5); }
        } else if (KIND == 2) {   // dependent chain of f32 fma (latency)
#pragma unroll
            for (int u = 0; u < 64; ++u) a0 = __builtin_fmaf(a0, m, c);
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + float(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int nblk : {256, 1})
    for (int kind = 0; kind < 2; ++kind)
        for (int threads : {256, 1024}) {     // 1 block per CU: waves/SIMD = threads/256 (min 1 wave on 1 SIMD)
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                if (rep == 1) hipEventRecord(e0, 0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(threads), 0, 0, out, iters, cyc);
                if (rep == 1) hipEventRecord(e1, 0);
                hipDeviceSynchronize();
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double instr_per_wave = iters * 64.0;
            const double waves_per_simd = threads >= 256 ? threads / 256.0 : 1.0;
            printf("blocks %3d kind %d threads %4d: %.2f ticks per wave-instr (one wave), %.2f ticks/instr/SIMD; kernel %.3f ms -> %.3f ns per wave-instr per SIMD, tick = %.3f ns\n", nblk, kind, threads,
                   c / instr_per_wave, c / (instr_per_wave * waves_per_simd), ms, ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / c);
        }
    return 0;
}
