// Microbenchmark (round 5): what does v_mfma_f32_16x16x4_f32 cost the VECTOR issue port of its SIMD on gfx950?
// A workgroup of up to 1024 lanes, one per CU; role group g = wave / 4 (waves w, w + 4, w + 8, w + 12 share a SIMD — the HW_ID
// each wave prints confirms it).  Every role runs `iters` rounds of one of:
//   kind 0  nothing (the wave leaves at once)
//   kind 1  16 independent v_fma_f32 per round
//   kind 2  16 v_mfma_f32_16x16x4_f32 per round on 4 accumulators (dependence distance 4 instructions = 128 cycles > 40)
//   kind 3  16 matrix instructions with `fill` independent v_fma_f32 behind each (one wave feeding both pipes)
//   kind 4  16 v_mfma_f32_32x32x2_f32 per round (the other f32-input shape)
//   kind 5  16 v_mfma_f32_32x32x16_bf16 per round (for contrast: a low-precision matrix instruction beside vector code)
//   kind 6  16 v_mfma_f32_16x16x4_f32 per round with FILL x `s_nop 7` (8 wait states each) behind every one: does a matrix wave that
//           does NOT present its next matrix instruction at once leave the vector issue port to the other waves?
//   kind 7  the same with v_mfma_f32_32x32x16_bf16
// Prints shader-clock ticks per instruction for one wave of every role.  The question behind it: a radix-16
// butterfly is ~340 vector instructions per 64 butterflies = 10.7 issue cycles per butterfly at 2 cycles per instruction; as four
// real 16 x 16 products it is 16 matrix instructions per 16 butterflies — if each of them holds the vector issue port for 8 cycles
// that is 8 issue cycles per butterfly, and the matrix form frees a quarter of the pass, not all of it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Roles { int kind[4]; int fill; };

template <int FILL>
__global__ __launch_bounds__(1024) void k(float* out, int iters, Roles r, long long* cyc, unsigned* hwid) {
    const int wave = threadIdx.x >> 6, role = wave >> 2;
    const int kind = r.kind[role];
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = float(threadIdx.x + i);
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    const float m = 1.0001f, c = 0.5f;
    float am = float(threadIdx.x & 15) * 0.01f, bm = float(threadIdx.x >> 4) * 0.02f;
    if (kind == 0) return;
    __syncthreads();      // (all remaining waves start together; idle roles left before — a barrier counts only live waves)
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (kind == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    } else if (kind == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d3, 0, 0, 0);
            }
        }
    } else if (kind == 6) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d0, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("s_nop 7");
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d1, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("s_nop 7");
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d2, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("s_nop 7");
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d3, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("s_nop 7");
            }
        }
    } else if (kind == 7) {
        f32x16 e0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, e1 = e0;
        bf16x8 ab, bb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(am + float(i)); bb[i] = (__bf16)(bm - float(i)); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, e0, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("s_nop 7");
                e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, e1, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("s_nop 7");
            }
        }
        d0[0] += e0[0] + e1[5];
    } else if (kind == 4) {
        f32x16 e0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, e1 = e0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bm, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x2f32(am, bm, e1, 0, 0, 0);
            }
        }
        d0[0] += e0[0] + e1[5];
    } else if (kind == 5) {
        f32x16 e0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, e1 = e0;
        bf16x8 ab, bb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(am + float(i)); bb[i] = (__bf16)(bm - float(i)); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, e1, 0, 0, 0);
            }
        }
        d0[0] += e0[0] + e1[5];
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d0, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[f]) : "v"(m), "v"(c));
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d1, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[f]) : "v"(m), "v"(c));
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d2, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[f]) : "v"(m), "v"(c));
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(am, bm, d3, 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[f]) : "v"(m), "v"(c));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    s += d0[0] + d1[1] + d2[2] + d3[3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        cyc[wave] = (long long)(t1 - t0);
        hwid[wave] = id;
    }
}

int main() {
    float* out; long long* cyc; unsigned* hwid;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16 * 8); hipMalloc(&hwid, 16 * 4);
    const int iters = 4000;
    struct Case { const char* name; int threads; Roles r; };
    // fill is a run-time trip count of a loop of single instructions: its branch overhead rides along (scalar), fine for a ratio
    std::vector<Case> cases = {
        {"1 wave/SIMD: vector", 256, {{1, 0, 0, 0}, 0}},
        {"1 wave/SIMD: matrix", 256, {{2, 0, 0, 0}, 0}},
        {"2 waves/SIMD: vector + vector", 512, {{1, 1, 0, 0}, 0}},
        {"2 waves/SIMD: matrix + matrix", 512, {{2, 2, 0, 0}, 0}},
        {"2 waves/SIMD: matrix + vector", 512, {{2, 1, 0, 0}, 0}},
        {"3 waves/SIMD: matrix + vector + vector", 768, {{2, 1, 1, 0}, 0}},
        {"4 waves/SIMD: matrix + 3 x vector", 1024, {{2, 1, 1, 1}, 0}},
        {"4 waves/SIMD: 2 x matrix + 2 x vector", 1024, {{2, 2, 1, 1}, 0}},
        {"3 waves/SIMD: vector x 3", 768, {{1, 1, 1, 0}, 0}},
        {"1 wave/SIMD: matrix with 2 vector behind each", 256, {{3, 0, 0, 0}, 2}},
        {"1 wave/SIMD: matrix with 4 vector behind each", 256, {{3, 0, 0, 0}, 4}},
        {"1 wave/SIMD: matrix with 6 vector behind each", 256, {{3, 0, 0, 0}, 6}},
        {"1 wave/SIMD: matrix with 8 vector behind each", 256, {{3, 0, 0, 0}, 8}},
        {"2 waves/SIMD: (matrix with 4 vector) + vector", 512, {{3, 1, 0, 0}, 4}},
        {"matrix + 2 x s_nop 7 each, alone", 256, {{6, 0, 0, 0}, 2}},
        {"(matrix + 2 x s_nop 7) + vector", 512, {{6, 1, 0, 0}, 2}},
        {"(matrix + 2 x s_nop 7) + 3 x vector", 1024, {{6, 1, 1, 1}, 2}},
        {"matrix + 4 x s_nop 7 each, alone", 256, {{6, 0, 0, 0}, 4}},
        {"(matrix + 4 x s_nop 7) + vector", 512, {{6, 1, 0, 0}, 4}},
        {"(matrix + 4 x s_nop 7) + 2 x vector", 768, {{6, 1, 1, 0}, 4}},
        {"(matrix + 4 x s_nop 7) + 3 x vector", 1024, {{6, 1, 1, 1}, 4}},
        {"(matrix + 6 x s_nop 7) + 3 x vector", 1024, {{6, 1, 1, 1}, 6}},
        {"2 x (matrix + 4 x s_nop 7) + 2 x vector", 1024, {{6, 6, 1, 1}, 4}},
        {"bf16 matrix + 4 x s_nop 7 each, alone", 256, {{7, 0, 0, 0}, 4}},
        {"(bf16 matrix + 4 x s_nop 7) + 3 x vector", 1024, {{7, 1, 1, 1}, 4}},
        {"1 wave/SIMD: f32 32x32x2", 256, {{4, 0, 0, 0}, 0}},
        {"2 waves/SIMD: f32 32x32x2 + vector", 512, {{4, 1, 0, 0}, 0}},
        {"3 waves/SIMD: f32 32x32x2 + 2 x vector", 768, {{4, 1, 1, 0}, 0}},
        {"1 wave/SIMD: bf16 32x32x16", 256, {{5, 0, 0, 0}, 0}},
        {"2 waves/SIMD: bf16 32x32x16 + vector", 512, {{5, 1, 0, 0}, 0}},
        {"3 waves/SIMD: bf16 32x32x16 + 2 x vector", 768, {{5, 1, 1, 0}, 0}},
        {"4 waves/SIMD: bf16 32x32x16 + 3 x vector", 1024, {{5, 1, 1, 1}, 0}},
    };
    for (int nblk : {1, 256}) {
        printf("---- %d workgroup(s)\n", nblk);
        for (auto& cs : cases) {
            hipMemset(cyc, 0, 16 * 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                switch (cs.r.fill) {
                    case 2: hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(cs.threads), 0, 0, out, iters, cs.r, cyc, hwid); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(nblk), dim3(cs.threads), 0, 0, out, iters, cs.r, cyc, hwid); break;
                    case 6: hipLaunchKernelGGL(k<6>, dim3(nblk), dim3(cs.threads), 0, 0, out, iters, cs.r, cyc, hwid); break;
                    case 8: hipLaunchKernelGGL(k<8>, dim3(nblk), dim3(cs.threads), 0, 0, out, iters, cs.r, cyc, hwid); break;
                    default: hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(cs.threads), 0, 0, out, iters, cs.r, cyc, hwid); break;
                }
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            long long c[16]; unsigned id[16];
            hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
            hipMemcpy(id, hwid, sizeof id, hipMemcpyDeviceToHost);
            printf("%-48s:", cs.name);
            for (int role = 0; role < cs.threads / 256; ++role) {
                const int kind = cs.r.kind[role];
                if (!kind) continue;
                const int w = role * 4;      // the role's wave on the SIMD of wave 0
                const double per_round = double(c[w]) / iters;
                if (kind == 1) printf("  vector %.2f ticks/instr", per_round / 16);
                else if (kind == 2 || kind == 4 || kind == 5 || kind == 6 || kind == 7) printf("  matrix %.2f ticks/instr", per_round / 16);
                else printf("  mixed %.1f ticks per (matrix + %d vector)", per_round / 16, cs.r.fill);
                printf(" [simd %u]", (id[w] >> 4) & 3u);
            }
            long long cmax = 0; for (int w = 0; w < 16; ++w) cmax = c[w] > cmax ? c[w] : cmax;
            printf("   | kernel %.1f us, longest wave %lld ticks -> %.2f ticks/ns\n", ms * 1e3, cmax, cmax / (ms * 1e6));
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
