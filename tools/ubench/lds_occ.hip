// What the runtime believes about LDS per CU, and the occupancy it reports for a 512-thread kernel at several LDS sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 4) void k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = 1.f; __syncthreads(); o[threadIdx.x] = s[(threadIdx.x + 1) & 511]; }
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu  maxSharedMemoryPerMultiProcessor %zu  CUs %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount);
    for (size_t kb : {8, 16, 24, 32, 40, 48, 64, 80}) {
        int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 512, kb * 1024);
        printf("dynamic LDS %zu KB: %d blocks/CU (%s)\n", kb, n, hipGetErrorString(e));
    }
    return 0;
}
