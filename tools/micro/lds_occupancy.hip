// How many 64-thread blocks with N bytes of dynamic LDS does a gfx950 CU hold?  (LDS limit of the persistent grid)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(float *out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    out[blockIdx.x * 64 + threadIdx.x] = lds[63 - threadIdx.x];
}
int main() {
    for (int bytes = 7168; bytes <= 11264; bytes += 256) {
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, (size_t)bytes);
        printf("dyn LDS %5d B: %d blocks/CU (%s)\n", bytes, n, hipGetErrorString(e));
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) == hipSuccess)
        printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, CUs %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount);
    return 0;
}
