// Does a wave64 VALU instruction cost less when only the lower 32 lanes are active?  (gfx950: SIMD-32, 2 passes)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int mode, int iters) {
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (mode == 1) on = lane < 32;          // lower half
    if (mode == 2) on = (lane & 1) == 0;    // every other lane (32 active, both halves)
    if (mode == 3) on = lane < 16;          // lower quarter
    if (mode == 4) on = lane >= 32;         // upper half
    if (mode == 5) on = lane < 8;
    float a = (float)lane, b = 1.0001f, c = 0.5f;
    unsigned x = lane * 2654435761u;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 16; j++) { a = a * b + c; c = c * b + a; }
#pragma unroll
            for (int j = 0; j < 4; j++) { unsigned long long p = (unsigned long long)x * 0xD2511F53u; x = (unsigned)(p >> 32) ^ (unsigned)p; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + (float)x;
}
int main() {
    float *d;
    hipMalloc(&d, 1024 * 256 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 4; waves *= 2)
    for (int mode = 0; mode <= 5; mode++) {
        hipLaunchKernelGGL(k, dim3(256 * 4), dim3(64 * waves), 0, 0, d, mode, 2000); // warm
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256 * 4), dim3(64 * waves), 0, 0, d, mode, 20000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("waves/block %d (= per SIMD) mode %d: %.3f ms\n", waves, mode, ms);
    }
    return 0;
}
