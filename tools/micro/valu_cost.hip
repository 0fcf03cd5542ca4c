// Issue cost of the VALU instructions the render kernel leans on (gfx950), 16 wavefronts per CU (4 per SIMD, the
// kernel's occupancy): independent streams, s_memtime inside the kernel; cycles per instruction per SIMD.
// Question (r03): what does Philox4x32 cost (20 x v_mad_u64_u32 per block) relative to plain fp32 work, and what do
// v_sqrt / v_rcp / v_div_* sequences, v_cndmask, v_cmp and v_max3 cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define BODY(INS)                                                                                                      \
    REP16(asm volatile(INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7)                                 \
                       : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)::"scc", "vcc", "s20", "s22", "s23");)
#define PKBODY(S) REP16(asm volatile(S S : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)::"scc", "vcc");)
#define I_ADD(r) "v_add_f32 " #r ", " #r ", 1.0\n"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", " #r ", 1.0\n"
#define I_MAX3(r) "v_max3_f32 " #r ", " #r ", " #r ", 1.0\n"
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", " #r "\n"
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", " #r "\n"
#define I_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", " #r "\n"
#define I_SQRT(r) "v_sqrt_f32 " #r ", " #r "\n"
#define I_RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define I_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", " #r ", vcc\n"
#define I_CMP(r) "v_cmp_gt_f32 vcc, " #r ", " #r "\n"
#define I_XOR(r) "v_xor_b32 " #r ", " #r ", " #r "\n"
#define I_ADDU(r) "v_add_u32 " #r ", " #r ", " #r "\n"
#define I_MBCNT(r) "v_mbcnt_lo_u32_b32 " #r ", -1, " #r "\n"
#define I_READLANE(r) "v_readlane_b32 s20, " #r ", 3\n"
#define I_CNDMASK64(r) "v_cndmask_b32_e64 " #r ", " #r ", " #r ", s[22:23]\n"
#define I_CNDMASKK(r) "v_cndmask_b32 " #r ", 0, " #r ", vcc\n"
#define I_CNDADD(r) "v_cndmask_b32 " #r ", " #r ", " #r ", vcc\n v_add_f32 " #r ", " #r ", 1.0\n"
#define I_BFI(r) "v_bfi_b32 " #r ", " #r ", " #r ", " #r "\n"
#define I_MIN(r) "v_min_f32 " #r ", " #r ", " #r "\n"
#define I_CMPS(r) "v_cmp_gt_f32_e64 s[22:23], " #r ", " #r "\n"
#define I_MOV(r) "v_mov_b32 " #r ", " #r "\n"
#define I_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define I_MED3(r) "v_med3_f32 " #r ", " #r ", " #r ", 1.0\n"
#define I_CND64VCC(r) "v_cndmask_b32_e64 " #r ", " #r ", " #r ", vcc\n"
#define I_ADDC(r) "v_addc_co_u32_e32 " #r ", vcc, " #r ", " #r ", vcc\n"
#define I_SNOP(r) "s_nop 0\n"
#define I_WRITELANE(r) "v_writelane_b32 " #r ", s22, 3\n"
#define I_CMPCND64(r) "v_cmp_gt_f32 vcc, " #r ", " #r "\n v_cndmask_b32_e64 " #r ", " #r ", " #r ", vcc\n"
#define I_CMPSCND64(r) "v_cmp_gt_f32_e64 s[22:23], " #r ", " #r "\n v_cndmask_b32_e64 " #r ", " #r ", " #r ", s[22:23]\n"
#define I_C3_32(r) "v_cmp_gt_f32 vcc, " #r ", " #r "\n v_cndmask_b32 " #r ", " #r ", " #r ", vcc\n v_cndmask_b32 " #r ", 1.0, " #r ", vcc\n v_cndmask_b32 " #r ", 2.0, " #r ", vcc\n"
#define I_C3_64(r) "v_cmp_gt_f32 vcc, " #r ", " #r "\n v_cndmask_b32_e64 " #r ", " #r ", " #r ", vcc\n v_cndmask_b32_e64 " #r ", 1.0, " #r ", vcc\n v_cndmask_b32_e64 " #r ", 2.0, " #r ", vcc\n"
#define I_C3_S(r) "v_cmp_gt_f32_e64 s[22:23], " #r ", " #r "\n v_cndmask_b32_e64 " #r ", " #r ", " #r ", s[22:23]\n v_cndmask_b32_e64 " #r ", 1.0, " #r ", s[22:23]\n v_cndmask_b32_e64 " #r ", 2.0, " #r ", s[22:23]\n"
#define I_CFF_32(r) "v_cmp_gt_f32 vcc, " #r ", " #r "\n v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_cndmask_b32 " #r ", 1.0, " #r ", vcc\n"
#define I_CFF_64(r) "v_cmp_gt_f32 vcc, " #r ", " #r "\n v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_cndmask_b32_e64 " #r ", 1.0, " #r ", vcc\n"
#define I_F32(r) "v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_cndmask_b32 " #r ", 1.0, " #r ", vcc\n v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_cndmask_b32 " #r ", 2.0, " #r ", vcc\n"
#define I_F64(r) "v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_cndmask_b32_e64 " #r ", 1.0, " #r ", vcc\n v_fma_f32 " #r ", " #r ", " #r ", 1.0\n v_cndmask_b32_e64 " #r ", 2.0, " #r ", vcc\n"
#define I_CMPCND(r) "v_cmp_gt_f32 vcc, " #r ", " #r "\n v_cndmask_b32 " #r ", " #r ", " #r ", vcc\n"

template <int MODE>
__global__ void k(unsigned long long *out, int iters) {
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
    unsigned long long w0 = threadIdx.x, w1 = 1, w2 = 2, w3 = 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { BODY(I_ADD) }
        else if (MODE == 1) { BODY(I_FMA) }
        else if (MODE == 2) { BODY(I_MAX3) }
        else if (MODE == 3) { BODY(I_MULLO) }
        else if (MODE == 4) { BODY(I_MULHI) }
        else if (MODE == 5) { BODY(I_MUL24) }
        else if (MODE == 6) { BODY(I_SQRT) }
        else if (MODE == 7) { BODY(I_RCP) }
        else if (MODE == 8) { BODY(I_CNDMASK) }
        else if (MODE == 9) { BODY(I_CMP) }
        else if (MODE == 10) { BODY(I_XOR) }
        else if (MODE == 11) { BODY(I_ADDU) }
        else if (MODE == 12) {
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n"
                               "v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3\n"
                               "v_mad_u64_u32 %0, vcc, %4, %6, %0\n v_mad_u64_u32 %1, vcc, %5, %7, %1\n"
                               "v_mad_u64_u32 %2, vcc, %6, %4, %2\n v_mad_u64_u32 %3, vcc, %7, %5, %3\n"
                               : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "scc", "vcc");)
        }
        else if (MODE == 13) { BODY(I_MBCNT) }
        else if (MODE == 15) { BODY(I_CNDMASK64) }
        else if (MODE == 16) { BODY(I_CNDMASKK) }
        else if (MODE == 17) { BODY(I_CNDADD) }
        else if (MODE == 18) { BODY(I_BFI) }
        else if (MODE == 19) { BODY(I_MIN) }
        else if (MODE == 20) { BODY(I_CMPS) }
        else if (MODE == 21) { BODY(I_MOV) }
        else if (MODE == 22) { BODY(I_LSHL) }
        else if (MODE == 23) { BODY(I_MED3) }
        else if (MODE == 24) { BODY(I_CMPCND) }
        else if (MODE == 14) { BODY(I_READLANE) }
        else if (MODE == 25) { BODY(I_CND64VCC) }
        else if (MODE == 34) { BODY(I_C3_32) }
        else if (MODE == 35) { BODY(I_C3_64) }
        else if (MODE == 36) { BODY(I_C3_S) }
        else if (MODE == 37) { BODY(I_CFF_32) }
        else if (MODE == 38) { BODY(I_CFF_64) }
        else if (MODE == 39) { BODY(I_F32) }
        else if (MODE == 40) { BODY(I_F64) }
        else if (MODE == 26) { BODY(I_ADDC) }
        else if (MODE == 27) { BODY(I_SNOP) }
        else if (MODE == 28) { BODY(I_WRITELANE) }
        else if (MODE == 29) { BODY(I_CMPCND64) }
        else if (MODE == 30) { BODY(I_CMPSCND64) }
        else if (MODE == 31) { PKBODY("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n") }
        else if (MODE == 32) { PKBODY("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n") }
        else if (MODE == 33) { PKBODY("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n") }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 1.2345f || (w0 ^ w1 ^ w2 ^ w3) == 0x1234567ull) out[0] = 0;
}
static int g_threads = 256; // 256: four wavefronts per SIMD (the kernel's occupancy); 64: one
template <int MODE>
static double run(const char *name, unsigned long long *d, int cus, double base) {
    const int iters = 1000, blocks = cus * 4, threads = g_threads;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    const int nw = blocks * (threads / 64);
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), d, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[nw / 2] / (iters * 128.0) / (threads / 64); // cycles per instruction per SIMD (its waves share it)
    printf("%-18s %6.2f cycles/instr/SIMD  (x%.2f of v_add_f32)\n", name, per, base > 0 ? per / base : 1.0);
    return per;
}
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    unsigned long long *d;
    hipMalloc(&d, (size_t)p.multiProcessorCount * 16 * sizeof(unsigned long long));
    const double b = run<0>("v_add_f32", d, p.multiProcessorCount, 0);
    run<1>("v_fma_f32", d, p.multiProcessorCount, b);
    run<2>("v_max3_f32", d, p.multiProcessorCount, b);
    run<10>("v_xor_b32", d, p.multiProcessorCount, b);
    run<11>("v_add_u32", d, p.multiProcessorCount, b);
    run<8>("v_cndmask_b32", d, p.multiProcessorCount, b);
    run<9>("v_cmp_gt_f32", d, p.multiProcessorCount, b);
    run<5>("v_mul_u32_u24", d, p.multiProcessorCount, b);
    run<3>("v_mul_lo_u32", d, p.multiProcessorCount, b);
    run<4>("v_mul_hi_u32", d, p.multiProcessorCount, b);
    run<12>("v_mad_u64_u32", d, p.multiProcessorCount, b);
    run<6>("v_sqrt_f32", d, p.multiProcessorCount, b);
    run<7>("v_rcp_f32", d, p.multiProcessorCount, b);
    run<13>("v_mbcnt_lo", d, p.multiProcessorCount, b);
    run<14>("v_readlane_b32", d, p.multiProcessorCount, b);
    run<15>("v_cndmask_e64 sgpr", d, p.multiProcessorCount, b);
    run<16>("v_cndmask 0,v,vcc", d, p.multiProcessorCount, b);
    run<17>("cndmask+add (x2)", d, p.multiProcessorCount, b);
    run<18>("v_bfi_b32", d, p.multiProcessorCount, b);
    run<19>("v_min_f32", d, p.multiProcessorCount, b);
    run<20>("v_cmp_e64 sgpr", d, p.multiProcessorCount, b);
    run<21>("v_mov_b32", d, p.multiProcessorCount, b);
    run<22>("v_lshlrev_b32", d, p.multiProcessorCount, b);
    run<23>("v_med3_f32", d, p.multiProcessorCount, b);
    run<24>("cmp+cndmask (x2)", d, p.multiProcessorCount, b);
    for (int pass = 0; pass < 2; pass++) { // r03, second look at the v_cndmask forms (VCC vs SGPR pair), packed fp32, hazard fillers
        g_threads = pass ? 64 : 256;
        printf("-- %d wavefront(s) per SIMD\n", g_threads / 64);
        const double b2 = run<0>("v_add_f32", d, p.multiProcessorCount, 0);
        run<8>("v_cndmask_e32 vcc", d, p.multiProcessorCount, b2);
        run<25>("v_cndmask_e64 vcc", d, p.multiProcessorCount, b2);
        run<15>("v_cndmask_e64 sgpr", d, p.multiProcessorCount, b2);
        run<26>("v_addc_co_u32 vcc", d, p.multiProcessorCount, b2);
        run<24>("cmp vcc+cnd e32 (x2)", d, p.multiProcessorCount, b2);
        run<29>("cmp vcc+cnd e64 (x2)", d, p.multiProcessorCount, b2);
        run<30>("cmp s+cnd e64 s (x2)", d, p.multiProcessorCount, b2);
        printf("   (groups of 2 or 4 instructions, "(xN)": the figure is cycles per GROUP)\n");
        run<34>("cmp;3 x cnd e32 (x4)", d, p.multiProcessorCount, b2);
        run<35>("cmp;3 x cnd e64 (x4)", d, p.multiProcessorCount, b2);
        run<36>("cmp s;3 x cnd s (x4)", d, p.multiProcessorCount, b2);
        run<37>("cmp;fma;fma;cnd32 (x4)", d, p.multiProcessorCount, b2);
        run<38>("cmp;fma;fma;cnd64 (x4)", d, p.multiProcessorCount, b2);
        run<39>("fma;cnd32;fma;cnd32 (x4)", d, p.multiProcessorCount, b2);
        run<40>("fma;cnd64;fma;cnd64 (x4)", d, p.multiProcessorCount, b2);
        run<27>("s_nop 0", d, p.multiProcessorCount, b2);
        run<28>("v_writelane_b32", d, p.multiProcessorCount, b2);
        run<31>("v_pk_fma_f32", d, p.multiProcessorCount, b2);
        run<32>("v_pk_mul_f32", d, p.multiProcessorCount, b2);
        run<33>("v_pk_add_f32", d, p.multiProcessorCount, b2);
    }
    return 0;
}
