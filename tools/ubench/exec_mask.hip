// Does a VALU instruction of a lone wave issue faster when few lanes are active?  (gfx950)
// Per variant: 512 dependent v_fma_f32, then 512 independent ones (4 chains), timed with s_memtime; one wave per workgroup,
// one workgroup on the device.  Build: hipcc --offload-arch=gfx950 -O3 exec_mask.hip -o exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); return (long long)t; }
template <int CH>
__device__ __forceinline__ float chain(float x, float a, float b) {
    float y0 = x, y1 = x + 1.f, y2 = x + 2.f, y3 = x + 3.f;
#pragma unroll
    for (int i = 0; i < 512 / CH; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y0) : "v"(a), "v"(b));
        if (CH > 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y1) : "v"(a), "v"(b));
        if (CH > 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y2) : "v"(a), "v"(b));
        if (CH > 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y3) : "v"(a), "v"(b));
    }
    return y0 + y1 + y2 + y3;
}
__global__ void k(float* out, long long* t, int active, float a, float b) {
    const int lane = threadIdx.x;
    float r = 0.f;
    long long t0 = 0, t1 = 0, t2 = 0;
    if (lane < active) {
        t0 = now();
        r = chain<1>(float(lane), a, b);
        t1 = now();
        r += chain<4>(r, a, b);
        t2 = now();
    }
    out[lane] = r;
    if (lane == 0) { t[0] = t1 - t0; t[1] = t2 - t1; }
}
int main() {
    float* o; long long* t;
    hipMalloc(&o, 256); hipMalloc(&t, 16);
    for (int active : {64, 48, 32, 17, 16, 8, 1}) {
        long long h[2];
        for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, active, 0.999f, 0.001f); hipDeviceSynchronize(); }
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("active lanes %2d: dependent %.2f ticks/instr, 4 chains %.2f ticks/instr (s_memtime ticks, 100 MHz: x24 = 2.4 GHz cycles)\n", active, h[0] / 512.0, h[1] / 512.0);
    }
    return 0;
}
