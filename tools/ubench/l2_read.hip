// l2_read.hip — how fast a CU reads cache-resident data with the correlation kernels' access shape: 16-byte buffer loads, a wave
// reading 1 KB per instruction, every workgroup sweeping a SPAN of bytes over and over; workgroups of one XCD share spans in groups
// (`share` workgroups read the same span: the spectra of a bin are read by the workgroups of cb workers, a code table by ~5 bins).
//   hipcc --offload-arch=gfx950 -O3 l2_read.hip -o l2_read && ./l2_read
// Prints, per (lanes, workgroups per CU, span, share): chip-wide TB/s and GB/s per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int T, int UNROLL>
__global__ __launch_bounds__(T) void sweep(const u32x4* __restrict__ base, unsigned* __restrict__ sink, int span_vec, int share,
                                           int n_spans, int reps) {
    // blockIdx & 7 = XCD (round-robin dispatch); inside an XCD consecutive workgroups share a span in groups of `share`
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int span_id = (xcd * 4096 + slot / share) % n_spans;
    const u32x4* p = base + size_t(span_id) * span_vec;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (int i = threadIdx.x; i + (UNROLL - 1) * T < span_vec; i += UNROLL * T) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * T];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
        asm volatile("" ::: "memory");
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc.x;
}

template <int T>
static void run(const u32x4* d, unsigned* sink, int wg_per_cu, size_t span_bytes, int share, size_t pool_bytes) {
    const int span_vec = int(span_bytes / 16), n_spans = int(pool_bytes / span_bytes);
    const int grid = 256 * wg_per_cu;
    const int reps = int((size_t(64) << 20) / span_bytes) > 0 ? int((size_t(64) << 20) / span_bytes) : 1;     // 64 MB read per workgroup
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((sweep<T, 8>), dim3(grid), dim3(T), 0, 0, d, sink, span_vec, share, n_spans, 2);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((sweep<T, 8>), dim3(grid), dim3(T), 0, 0, d, sink, span_vec, share, n_spans, reps);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = double(grid) * reps * double(span_vec) * 16.0;
    const int distinct = (grid / 8 + share - 1) / share * 8;
    printf("lanes %4d x %d WG/CU  span %4zu KB  share %2d  (distinct spans %4d = %6.1f MB)  %6.2f TB/s  %6.1f GB/s per CU\n", T, wg_per_cu,
           span_bytes >> 10, share, distinct, distinct * double(span_bytes) / 1e6, bytes / ms / 1e9, bytes / ms / 1e6 / 256.0);
}

int main() {
    const size_t pool = size_t(1) << 30;
    u32x4* d; unsigned* sink;
    CK(hipMalloc(&d, pool)); CK(hipMemset(d, 1, pool));
    CK(hipMalloc(&sink, 1 << 20));
    for (size_t span : {size_t(128) << 10, size_t(512) << 10}) {
        for (int share : {32, 8, 4, 1}) {
            run<1024>(d, sink, 1, span, share, pool);
            run<512>(d, sink, 2, span, share * 2, pool);
        }
    }
    run<256>(d, sink, 4, size_t(128) << 10, 16, pool);
    run<256>(d, sink, 8, size_t(128) << 10, 32, pool);
    return 0;
}
