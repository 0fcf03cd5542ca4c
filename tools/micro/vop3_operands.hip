// Does a three-operand VALU instruction cost more when its sources are three DIFFERENT VGPRs (r04)?
// valu_cost.hip measured v_fma_f32 r, r, r, 1.0 at the price of v_add_f32 — one register, one constant.  The render
// kernel's one-fma slab distances (plane * inv_d + nood: three distinct VGPRs) made the 4-wide visit 24 instructions
// shorter and the frame 3.5 % SLOWER (profiles/r04_experiments/ab_fma_slab.log).  This measures, at 4 wavefronts per
// SIMD like the kernel: VOP3 with 1 / 2 / 3 distinct VGPR sources in the same and in different register banks (bank =
// register number mod 4), the VOP2 pair it would replace (v_sub + v_mul), v_fmac (VOP2 encoding of fma), v_max3.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
// eight independent destinations v40..v47; sources v48.. (set up once)
#define CL "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc"
#define G8(a) a(40) a(41) a(42) a(43) a(44) a(45) a(46) a(47)
#define STR(x) #x
// 3 distinct sources, all in different banks (48 %4=0, 49=1, 50=2) ; same bank (48, 52, 56)
#define FMA_3DIFF(d) "v_fma_f32 v" STR(d) ", v48, v49, v50\n"
#define FMA_3SAME(d) "v_fma_f32 v" STR(d) ", v48, v52, v56\n"
#define FMA_2DIFF(d) "v_fma_f32 v" STR(d) ", v48, v49, 1.0\n"
#define FMA_2SAME(d) "v_fma_f32 v" STR(d) ", v48, v52, 1.0\n"
#define FMA_1(d) "v_fma_f32 v" STR(d) ", v48, v48, 1.0\n"
#define FMA_ACC(d) "v_fma_f32 v" STR(d) ", v48, v49, v" STR(d) "\n"
#define FMAC(d) "v_fmac_f32 v" STR(d) ", v48, v49\n"
#define FMA_SGPR(d) "v_fma_f32 v" STR(d) ", v48, s20, v50\n"
#define SUBMUL(d) "v_sub_f32 v" STR(d) ", v48, v49\n v_mul_f32 v" STR(d) ", v" STR(d) ", v50\n"
#define MUL_2(d) "v_mul_f32 v" STR(d) ", v48, v49\n"
#define ADD_2(d) "v_add_f32 v" STR(d) ", v48, v49\n"
#define ADD_1(d) "v_add_f32 v" STR(d) ", v48, 1.0\n"
#define MAX3_3(d) "v_max3_f32 v" STR(d) ", v48, v49, v50\n"
#define MAX_2(d) "v_max_f32 v" STR(d) ", v48, v49\n"
#define MAXMAX(d) "v_max_f32 v" STR(d) ", v48, v49\n v_max_f32 v" STR(d) ", v" STR(d) ", v50\n"
#define CND_E64(d) "v_cndmask_b32_e64 v" STR(d) ", v48, v49, s[22:23]\n"
#define ADD_S(d) "v_add_f32 v" STR(d) ", s20, v48\n"
#define SUB_S(d) "v_sub_f32 v" STR(d) ", s20, v48\n"
#define SUBREV_S(d) "v_subrev_f32 v" STR(d) ", s20, v48\n"
#define MUL_S(d) "v_mul_f32 v" STR(d) ", s20, v48\n"
#define MAX_S(d) "v_max_f32 v" STR(d) ", s20, v48\n"
#define FMA_LIT(d) "v_fma_f32 v" STR(d) ", v48, v49, 0x40490fdb\n"
#define ADD_LIT(d) "v_add_f32 v" STR(d) ", 0x40490fdb, v48\n"
#define MOV_S(d) "v_mov_b32 v" STR(d) ", s20\n"
#define FMAC_S(d) "v_fmac_f32 v" STR(d) ", s20, v48\n"
#define MADAK(d) "v_fmaak_f32 v" STR(d) ", v48, v49, 0x40490fdb\n"
#define CMP_S(d) "v_cmp_gt_f32 vcc, s20, v48\n"
#define SUBSUB(d) "v_sub_f32 v" STR(d) ", v48, v49\n v_sub_f32 v" STR(d) ", v" STR(d) ", v50\n"
#define PKFMA(d) "v_pk_fma_f32 v[" STR(d) ":" STR(d) "+1], v[48:49], v[50:51], v[52:53]\n"

template <int MODE>
__global__ void k(unsigned long long *out, int iters) {
    asm volatile("v_mov_b32 v48, 1.0\n v_mov_b32 v49, 2.0\n v_mov_b32 v50, 0.5\n v_mov_b32 v51, 0.5\n v_mov_b32 v52, 1.0\n v_mov_b32 v53, 1.0\n v_mov_b32 v56, 0.5\n"
                 "s_mov_b32 s20, 1.0\n s_mov_b64 s[22:23], -1\n" ::: "v48", "v49", "v50", "v51", "v52", "v53", "v56", "s20", "s22", "s23");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP16(asm volatile(G8(ADD_1) ::: CL);) }
        else if (MODE == 1) { REP16(asm volatile(G8(FMA_1) ::: CL);) }
        else if (MODE == 2) { REP16(asm volatile(G8(FMA_2DIFF) ::: CL);) }
        else if (MODE == 3) { REP16(asm volatile(G8(FMA_2SAME) ::: CL);) }
        else if (MODE == 4) { REP16(asm volatile(G8(FMA_3DIFF) ::: CL);) }
        else if (MODE == 5) { REP16(asm volatile(G8(FMA_3SAME) ::: CL);) }
        else if (MODE == 6) { REP16(asm volatile(G8(FMA_ACC) ::: CL);) }
        else if (MODE == 7) { REP16(asm volatile(G8(FMAC) ::: CL);) }
        else if (MODE == 8) { REP16(asm volatile(G8(FMA_SGPR) ::: CL);) }
        else if (MODE == 9) { REP16(asm volatile(G8(SUBMUL) ::: CL);) }
        else if (MODE == 10) { REP16(asm volatile(G8(MUL_2) ::: CL);) }
        else if (MODE == 11) { REP16(asm volatile(G8(ADD_2) ::: CL);) }
        else if (MODE == 12) { REP16(asm volatile(G8(MAX3_3) ::: CL);) }
        else if (MODE == 13) { REP16(asm volatile(G8(MAX_2) ::: CL);) }
        else if (MODE == 14) { REP16(asm volatile(G8(MAXMAX) ::: CL);) }
        else if (MODE == 15) { REP16(asm volatile(G8(CND_E64) ::: CL);) }
        else if (MODE == 16) { REP16(asm volatile(G8(ADD_S) ::: CL);) }
        else if (MODE == 17) { REP16(asm volatile(G8(SUB_S) ::: CL);) }
        else if (MODE == 18) { REP16(asm volatile(G8(MUL_S) ::: CL);) }
        else if (MODE == 19) { REP16(asm volatile(G8(MAX_S) ::: CL);) }
        else if (MODE == 20) { REP16(asm volatile(G8(MADAK) ::: CL);) }
        else if (MODE == 21) { REP16(asm volatile(G8(ADD_LIT) ::: CL);) }
        else if (MODE == 22) { REP16(asm volatile(G8(MOV_S) ::: CL);) }
        else if (MODE == 23) { REP16(asm volatile(G8(FMAC_S) ::: CL);) }
        else if (MODE == 24) { REP16(asm volatile(G8(SUBREV_S) ::: CL);) }
        else if (MODE == 25) { REP16(asm volatile(G8(CMP_S) ::: CL);) }
        else if (MODE == 26) { REP16(asm volatile(G8(SUBSUB) ::: CL);) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
static double run(const char *name, unsigned long long *d, int cus, double base, int per_group) {
    const int iters = 1000, blocks = cus * 4, threads = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    const int nw = blocks * (threads / 64);
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), d, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[nw / 2] / (iters * 128.0) / (threads / 64); // per GROUP (one G8 slot) per SIMD
    printf("%-44s %6.2f /group/SIMD (%d instr)  x%.2f of v_add_f32\n", name, per, per_group, base > 0 ? per / base : 1.0);
    return per;
}
int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * p.multiProcessorCount * 4 * 4);
    const double b = run<0>("v_add_f32 d, v, 1.0", d, p.multiProcessorCount, 0, 1);
    run<11>("v_add_f32 d, v, v (2 VGPR, banks 0 1)", d, p.multiProcessorCount, b, 1);
    run<10>("v_mul_f32 d, v, v", d, p.multiProcessorCount, b, 1);
    run<1>("v_fma_f32 d, v, v(same), 1.0", d, p.multiProcessorCount, b, 1);
    run<2>("v_fma_f32 d, v, v, 1.0 (banks 0 1)", d, p.multiProcessorCount, b, 1);
    run<3>("v_fma_f32 d, v, v, 1.0 (banks 0 0)", d, p.multiProcessorCount, b, 1);
    run<4>("v_fma_f32 d, v, v, v (banks 0 1 2)", d, p.multiProcessorCount, b, 1);
    run<5>("v_fma_f32 d, v, v, v (banks 0 0 0)", d, p.multiProcessorCount, b, 1);
    run<6>("v_fma_f32 d, v, v, d", d, p.multiProcessorCount, b, 1);
    run<7>("v_fmac_f32 d, v, v", d, p.multiProcessorCount, b, 1);
    run<8>("v_fma_f32 d, v, s, v", d, p.multiProcessorCount, b, 1);
    run<9>("v_sub_f32 + v_mul_f32 (the pair an fma replaces)", d, p.multiProcessorCount, b, 2);
    run<12>("v_max3_f32 d, v, v, v", d, p.multiProcessorCount, b, 1);
    run<13>("v_max_f32 d, v, v", d, p.multiProcessorCount, b, 1);
    run<14>("v_max_f32 x2 (what a max3 replaces)", d, p.multiProcessorCount, b, 2);
    run<15>("v_cndmask_b32_e64 d, v, v, s[]", d, p.multiProcessorCount, b, 1);
    run<16>("v_add_f32 d, s, v   (VOP2, SGPR source)", d, p.multiProcessorCount, b, 1);
    run<17>("v_sub_f32 d, s, v", d, p.multiProcessorCount, b, 1);
    run<24>("v_subrev_f32 d, s, v", d, p.multiProcessorCount, b, 1);
    run<18>("v_mul_f32 d, s, v", d, p.multiProcessorCount, b, 1);
    run<23>("v_fmac_f32 d, s, v", d, p.multiProcessorCount, b, 1);
    run<19>("v_max_f32 d, s, v", d, p.multiProcessorCount, b, 1);
    run<25>("v_cmp_gt_f32 vcc, s, v", d, p.multiProcessorCount, b, 1);
    run<22>("v_mov_b32 d, s", d, p.multiProcessorCount, b, 1);
    run<21>("v_add_f32 d, literal, v", d, p.multiProcessorCount, b, 1);
    run<20>("v_fmaak_f32 d, v, v, literal", d, p.multiProcessorCount, b, 1);
    run<26>("v_sub_f32 x2 dependent (v, v)", d, p.multiProcessorCount, b, 2);
    return 0;
}
