// What issue resources do the wavefronts of a CU share on gfx950?  (r03 question: the cooperative traversal round costs
// ~3 800 cycles whatever its lane utilisation, for ~220 VALU + ~190 scalar/branch instructions per wavefront at 16
// wavefronts per CU: is the scalar unit shared by the four SIMDs?)
// Streams of INDEPENDENT instructions of one kind (or an alternating mix), timed with s_memtime inside the kernel, for
// 1 / 4 / 8 / 16 wavefronts per CU (blocks of 256 threads = one wavefront per SIMD, k blocks per CU).
// Prints cycles per instruction per wavefront and the aggregate instructions per cycle per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// mode 0: SALU only (s_add_u32 on 8 registers), 1: VALU only (v_add_f32 on 8 registers), 2: alternating SALU/VALU,
// 3: s_and_b64 on SGPR pairs (the exec-mask bookkeeping kind), 4: VALU : SALU = 1 : 1 with a branch every 16
// 5: v_cmp + s_and_b64 + v_cndmask triplets (compare results combined on the scalar unit)
template <int MODE>
__global__ void k(unsigned long long *out, int iters) {
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
    unsigned long long m0 = 1, m1 = 2, m2 = 3, m3 = 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            REP16(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                               "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc", "vcc");)
        } else if (MODE == 1) {
            REP16(asm volatile("v_add_f32 %0, %0, 1.0\n v_add_f32 %1, %1, 1.0\n v_add_f32 %2, %2, 1.0\n v_add_f32 %3, %3, 1.0\n"
                               "v_add_f32 %4, %4, 1.0\n v_add_f32 %5, %5, 1.0\n v_add_f32 %6, %6, 1.0\n v_add_f32 %7, %7, 1.0\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : : "scc", "vcc");)
        } else if (MODE == 2) {
            REP16(asm volatile("s_add_u32 %0, %0, 1\n v_add_f32 %4, %4, 1.0\n s_add_u32 %1, %1, 1\n v_add_f32 %5, %5, 1.0\n"
                               "s_add_u32 %2, %2, 1\n v_add_f32 %6, %6, 1.0\n s_add_u32 %3, %3, 1\n v_add_f32 %7, %7, 1.0\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "scc", "vcc");)
        } else if (MODE == 3) {
            REP16(asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_and_b64 %2, %2, %3\n s_or_b64 %3, %3, %0\n"
                               "s_and_b64 %0, %0, %2\n s_or_b64 %1, %1, %3\n s_and_b64 %2, %2, %0\n s_or_b64 %3, %3, %1\n"
                               : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc", "vcc");)
        } else if (MODE == 4) {
            REP16(asm volatile("s_add_u32 %0, %0, 1\n v_add_f32 %4, %4, 1.0\n s_add_u32 %1, %1, 1\n v_add_f32 %5, %5, 1.0\n"
                               "s_add_u32 %2, %2, 1\n v_add_f32 %6, %6, 1.0\n s_cmp_eq_u32 %3, 0\n v_add_f32 %7, %7, 1.0\n"
                               "s_cbranch_scc1 1f\n s_nop 0\n1:\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "scc", "vcc");)
        } else {
            REP16(asm volatile("v_cmp_gt_f32_e64 %4, %0, %1\n v_cmp_lt_f32_e64 %5, %2, %3\n s_and_b64 %4, %4, %5\n v_cndmask_b32_e64 %0, %0, %1, %4\n"
                               "v_cmp_gt_f32_e64 %6, %2, %3\n v_cmp_lt_f32_e64 %7, %0, %1\n s_and_b64 %6, %6, %7\n v_cndmask_b32_e64 %2, %2, %3, %6\n"
                               : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc", "vcc");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 == 0xdeadbeefu || v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 1.2345f || (m0 ^ m1 ^ m2 ^ m3) == 0x1234567ull)
        out[0] = 0;
}

template <int MODE>
static void run(const char *name, int per_iter, unsigned long long *d, int cus) {
    const int iters = 2000;
    for (int wpc : {1, 4, 8, 16}) { // wavefronts per CU
        const int threads = wpc == 1 ? 64 : 256, blocks_per_cu = wpc == 1 ? 1 : wpc / 4;
        const int blocks = cus * blocks_per_cu;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 200);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * threads / 64;
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), d, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[nw / 2];
        const double n = (double)iters * per_iter;
        printf("%-28s %2d waves/CU: %6.2f cycles/instr/wave  -> %5.2f instr/cycle/CU   (wall %.3f ms)\n", name, wpc, med / n,
               n * wpc / med, ms);
    }
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned long long *d;
    hipMalloc(&d, (size_t)cus * 16 * sizeof(unsigned long long));
    printf("CUs %d; s_memtime cycles (100 MHz constant clock on some parts: compare rows, not absolute values)\n", cus);
    run<0>("SALU s_add_u32", 128, d, cus);
    run<3>("SALU s_and/or_b64", 128, d, cus);
    run<1>("VALU v_add_f32", 128, d, cus);
    run<2>("SALU,VALU alternating", 128, d, cus);
    run<4>("SALU,VALU + branch / 8", 144, d, cus);
    run<5>("v_cmp x2, s_and, v_cndmask", 128, d, cus);
    return 0;
}
