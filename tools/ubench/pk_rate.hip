// Microbenchmark: issue cost of v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 vs v_fma_f32 on gfx950 (4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, int iters) {
    v2 a0 = {float(threadIdx.x), 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const v2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {   // packed fma
                a0 = __builtin_elementwise_fma(a0, m, c); a1 = __builtin_elementwise_fma(a1, m, c); a2 = __builtin_elementwise_fma(a2, m, c); a3 = __builtin_elementwise_fma(a3, m, c);
                a4 = __builtin_elementwise_fma(a4, m, c); a5 = __builtin_elementwise_fma(a5, m, c); a6 = __builtin_elementwise_fma(a6, m, c); a7 = __builtin_elementwise_fma(a7, m, c);
            } else if (KIND == 1) {   // packed add
                a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
            } else if (KIND == 2) {   // packed mul
                a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
            } else if (KIND == 4) {   // packed add, BOTH operands VGPR pairs
                a0 = a0 + a4; a1 = a1 + a5; a2 = a2 + a6; a3 = a3 + a7; a4 = a4 + a1; a5 = a5 + a2; a6 = a6 + a3; a7 = a7 + a0;
            } else if (KIND == 5) {   // packed fma, THREE VGPR-pair operands
                a0 = __builtin_elementwise_fma(a0, a4, a1); a1 = __builtin_elementwise_fma(a1, a5, a2); a2 = __builtin_elementwise_fma(a2, a6, a3); a3 = __builtin_elementwise_fma(a3, a7, a0);
                a4 = __builtin_elementwise_fma(a4, a0, a5); a5 = __builtin_elementwise_fma(a5, a1, a6); a6 = __builtin_elementwise_fma(a6, a2, a7); a7 = __builtin_elementwise_fma(a7, a3, a4);
            } else if (KIND == 6) {   // scalar fma, three VGPR operands
                a0.x = __builtin_fmaf(a0.x, a4.x, a1.x); a1.x = __builtin_fmaf(a1.x, a5.x, a2.x); a2.x = __builtin_fmaf(a2.x, a6.x, a3.x); a3.x = __builtin_fmaf(a3.x, a7.x, a0.x);
                a4.x = __builtin_fmaf(a4.x, a0.x, a5.x); a5.x = __builtin_fmaf(a5.x, a1.x, a6.x); a6.x = __builtin_fmaf(a6.x, a2.x, a7.x); a7.x = __builtin_fmaf(a7.x, a3.x, a4.x);
            } else if (KIND == 7) {   // scalar add, two VGPR operands
                a0.x = a0.x + a4.x; a1.x = a1.x + a5.x; a2.x = a2.x + a6.x; a3.x = a3.x + a7.x; a4.x = a4.x + a1.x; a5.x = a5.x + a2.x; a6.x = a6.x + a3.x; a7.x = a7.x + a0.x;
            } else if (KIND == 3) {   // scalar fma on .x only (reference)
                a0.x = __builtin_fmaf(a0.x, m.x, c.x); a1.x = __builtin_fmaf(a1.x, m.x, c.x); a2.x = __builtin_fmaf(a2.x, m.x, c.x); a3.x = __builtin_fmaf(a3.x, m.x, c.x);
                a4.x = __builtin_fmaf(a4.x, m.x, c.x); a5.x = __builtin_fmaf(a5.x, m.x, c.x); a6.x = __builtin_fmaf(a6.x, m.x, c.x); a7.x = __builtin_fmaf(a7.x, m.x, c.x);
            }
        }
    }
    v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
    float* out; hipMalloc(&out, 1 << 22);
    const int iters = 2000;
    const char* names[] = {"v_pk_fma_f32 (1 vgpr src)", "v_pk_add_f32 (1 vgpr src)", "v_pk_mul_f32 (1 vgpr src)", "v_fma_f32 (1 vgpr src)", "v_pk_add_f32 (2 vgpr src)", "v_pk_fma_f32 (3 vgpr src)", "v_fma_f32 (3 vgpr src)", "v_add_f32 (2 vgpr src)"};
    for (int kind = 0; kind < 8; ++kind) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) hipEventRecord(e0, 0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (kind == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(1024), 0, 0, out, iters);
            if (rep == 1) hipEventRecord(e1, 0);
            hipDeviceSynchronize();
        }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s 4 waves/SIMD: %.3f ms -> %.3f ns per wave-instruction per SIMD\n", names[kind], ms, ms * 1e6 / (iters * 64.0 * 4));
    }
    return 0;
}
