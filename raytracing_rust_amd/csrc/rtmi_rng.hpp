// rtmi_rng.hpp — Philox4x32-10 counter streams and the rejection samplers.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include <type_traits>

#include "rtmi_types.hpp"

// ----------------------------------------------------------------------------------
// Philox4x32-10; stream = (block, sample, pixel, 0) under key = seed
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                       uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of separate high and low multiplies:
        // 32-bit integer multiplies are the slow VALU operations of this path
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // three-input XOR as ONE v_bitop3_b32 (truth table 0x96; gfx950 has no v_xor3): 2 instead of 4 per round
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
#else
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
#endif
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
// ---- RngRing ----------------------------------------------------------------------------------------------------
// Stream state of one path: the next word to read (`rd`) and to generate (`wr`, a multiple of 4), both counted from
// the start of the stream, and a ring of 8 words in LDS ([slot][lane] of the wavefront, conflict-free): up to two
// Philox blocks buffered.  A lane that is short of words triggers an evaluation; EVERY lane of the active set that
// has room for a block (<= 4 words buffered) generates its next one in that same evaluation.  With the 4-word
// register buffer a single draw (fog, shutter time, Schlick) found "some lane at a block boundary" nearly every
// time and evaluated Philox for the quarter of the lanes that were; now the lanes stay in phase and most single
// draws find their word buffered.  Same stream, same words, same order.
struct RngRing {
    uint32_t sample, pixel;
    uint32_t rd, wr;
    uint32_t *ring; // LDS: this lane's column, stride 64 words between slots
};
#define RTMI_RNG_RING_WORDS 512 /* 8 slots x 64 lanes per wavefront */
__device__ __forceinline__ void rng_attach(RngRing &g, uint32_t *wave_ring) { g.ring = wave_ring + (threadIdx.x & 63); }
__device__ __forceinline__ void rng_init(RngRing &g, uint32_t sample, uint32_t pixel) {
    g.sample = sample; g.pixel = pixel; g.rd = 0u; g.wr = 0u;
}
// all ACTIVE lanes call; afterwards every active lane holds at least n (<= 3) unread words
__device__ __forceinline__ void rng_need(RngRing &g, uint32_t n, uint32_t k0, uint32_t k1) {
    const uint32_t have = g.wr - g.rd;
    if (__ballot(have < n) != 0ull) { // uniform over the active lanes
        if (have <= 4u) {
            uint32_t o0, o1, o2, o3;
            philox(g.wr >> 2, g.sample, g.pixel, 0u, k0, k1, o0, o1, o2, o3);
            uint32_t *w = g.ring + ((g.wr & 4u) << 6); // slots 0..3 or 4..7
            w[0] = o0; w[64] = o1; w[128] = o2; w[192] = o3;
            g.wr += 4u;
        }
    }
}
__device__ __forceinline__ uint32_t rng_word(RngRing &g) {
    const uint32_t w = g.ring[(g.rd & 7u) << 6];
    g.rd++;
    return w;
}
// rng.gen::<f64>() of the reference (24-bit uniform, rtmi_u01)
__device__ __forceinline__ float rng_uniform(RngRing &g, uint32_t k0, uint32_t k1) {
    rng_need(g, 1u, k0, k1);
    return rtmi_u01(rng_word(g));
}
__device__ __forceinline__ void rng_take3(RngRing &g, uint32_t k0, uint32_t k1, uint32_t &w0, uint32_t &w1, uint32_t &w2) {
    rng_need(g, 3u, k0, k1);
    w0 = rng_word(g); w1 = rng_word(g); w2 = rng_word(g);
}
__device__ __forceinline__ void rng_take2(RngRing &g, uint32_t k0, uint32_t k1, uint32_t &w0, uint32_t &w1) {
    rng_need(g, 2u, k0, k1);
    w0 = rng_word(g); w1 = rng_word(g);
}
// ---- RngReg: one 4-word block in registers ---------------------------------------------------------------------
// Measured (r02): the ring wins wherever the kernel has no BVH to traverse (cornell_box +16 %, cornell_smoke +12 %,
// two_spheres +23 %) and loses 2 % where it has (final_scene, random_spheres: a draw then waits for LDS inside the
// long item loop — and a hybrid that keeps the register block and parks one prefetched block per lane in LDS, so that
// draws never wait for LDS, loses 3 % there as well.  Counters of the ring on final_scene: VALU instructions +0.6 % —
// the evaluations it saves on single draws are spent on new paths joining with empty buffers — LDS instructions
// +52 %, issue-wait share +17 %, same clock);
// the cooperative kernel therefore takes RngRing in its lean instantiation and RngReg in the other.
struct RngReg {
    uint32_t block, sample, pixel;
    uint32_t b0, b1, b2, b3;
    uint32_t pos;
};
__device__ __forceinline__ void rng_init(RngReg &g, uint32_t sample, uint32_t pixel) {
    g.block = 0; g.sample = sample; g.pixel = pixel; g.pos = 4;
}
__device__ __forceinline__ void rng_attach(RngReg &, uint32_t *) {}
// Word selection (r03).  The n-th unread word of a lane sits at position pos + n of the sequence b0 b1 b2 b3 | n0 n1 n2
// (current block, then the next one).  A chain "pos == 0 ? b0 : pos == 1 ? b1 : ..." is turned into a switch by the
// compiler and lowered to nested exec-mask regions — dozens of scalar instructions per draw, in the innermost loop of the
// rejection samplers.  Written as a two-stage barrel shifter on the bits of pos it stays a handful of v_cndmask:
//   stage 1 (bit 0 of pos): y[i] = x[i + 1] or x[i];  stage 2 (bit 1): z[i] = y[i + 2] or y[i];  pos == 4: the new block.
#define RTMI_SEL(c, a, b) ((c) ? (a) : (b))

// rng.gen::<f64>() of the reference (24-bit uniform, rtmi_u01)
__device__ __forceinline__ float rng_uniform(RngReg &g, uint32_t k0, uint32_t k1) {
    if (g.pos == 4) {
        philox(g.block, g.sample, g.pixel, 0u, k0, k1, g.b0, g.b1, g.b2, g.b3);
        g.block++;
        g.pos = 0;
    }
    const bool bit0 = (g.pos & 1u) != 0u, bit1 = (g.pos & 2u) != 0u;
    const uint32_t y0 = RTMI_SEL(bit0, g.b1, g.b0), y1 = RTMI_SEL(bit0, g.b3, g.b2);
    const uint32_t w = RTMI_SEL(bit1, y1, y0);
    g.pos++;
    return rtmi_u01(w);
}

// The next THREE (resp. TWO) consecutive words of the stream with at most ONE Philox evaluation
// for the whole wavefront.  Calling rng_uniform three times evaluates Philox up to three times per
// wavefront (lanes sit at different positions of their 4-word blocks, so at every call some lane
// needs a refill and the others wait).  Same stream, same words, same order: bit-identical.
__device__ __forceinline__ void rng_take3(RngReg &g, uint32_t k0, uint32_t k1, uint32_t &w0, uint32_t &w1, uint32_t &w2) {
    uint32_t n0 = 0u, n1 = 0u, n2 = 0u, n3 = 0u;
    const uint32_t pos = g.pos;
    const bool need = pos >= 2u; // fewer than three words left in the current block
    if (need) {
        philox(g.block, g.sample, g.pixel, 0u, k0, k1, n0, n1, n2, n3);
        g.block++;
    }
    const bool bit0 = (pos & 1u) != 0u, bit1 = (pos & 2u) != 0u, full = pos == 4u;
    const uint32_t y0 = RTMI_SEL(bit0, g.b1, g.b0), y1 = RTMI_SEL(bit0, g.b2, g.b1), y2 = RTMI_SEL(bit0, g.b3, g.b2),
                   y3 = RTMI_SEL(bit0, n0, g.b3), y4 = RTMI_SEL(bit0, n1, n0);
    const uint32_t z0 = RTMI_SEL(bit1, y2, y0), z1 = RTMI_SEL(bit1, y3, y1), z2 = RTMI_SEL(bit1, y4, y2);
    w0 = RTMI_SEL(full, n0, z0); w1 = RTMI_SEL(full, n1, z1); w2 = RTMI_SEL(full, n2, z2);
    g.b0 = RTMI_SEL(need, n0, g.b0); g.b1 = RTMI_SEL(need, n1, g.b1); g.b2 = RTMI_SEL(need, n2, g.b2); g.b3 = RTMI_SEL(need, n3, g.b3);
    g.pos = need ? pos - 1u : pos + 3u; // 2->1, 3->2, 4->3 in the new block
}
__device__ __forceinline__ void rng_take2(RngReg &g, uint32_t k0, uint32_t k1, uint32_t &w0, uint32_t &w1) {
    uint32_t n0 = 0u, n1 = 0u, n2 = 0u, n3 = 0u;
    const uint32_t pos = g.pos;
    const bool need = pos >= 3u;
    if (need) {
        philox(g.block, g.sample, g.pixel, 0u, k0, k1, n0, n1, n2, n3);
        g.block++;
    }
    const bool bit0 = (pos & 1u) != 0u, bit1 = (pos & 2u) != 0u, full = pos == 4u;
    const uint32_t y0 = RTMI_SEL(bit0, g.b1, g.b0), y1 = RTMI_SEL(bit0, g.b2, g.b1), y2 = RTMI_SEL(bit0, g.b3, g.b2), y3 = RTMI_SEL(bit0, n0, g.b3);
    const uint32_t z0 = RTMI_SEL(bit1, y2, y0), z1 = RTMI_SEL(bit1, y3, y1);
    w0 = RTMI_SEL(full, n0, z0); w1 = RTMI_SEL(full, n1, z1);
    g.b0 = RTMI_SEL(need, n0, g.b0); g.b1 = RTMI_SEL(need, n1, g.b1); g.b2 = RTMI_SEL(need, n2, g.b2); g.b3 = RTMI_SEL(need, n3, g.b3);
    g.pos = need ? pos - 2u : pos + 2u; // 3->1, 4->2
}


// src/util.rs:4-13 (draws x, y, z per trial)
template <typename RngT>
__device__ __forceinline__ F3 random_in_unit_sphere(RngT &g, uint32_t k0, uint32_t k1) {
    for (;;) {
        uint32_t w0, w1, w2;
        rng_take3(g, k0, k1, w0, w1, w2);
        // 2u - 1 with u = k * 2^-24 (rtmi_u01) as ONE fma on the integer: k * 2^-23 and (2u) - 1 are both exact (a
        // multiple of 2^-23 in [-1, 1)), so fma(k, 2^-23, -1) is the same value without the separate scaling
        F3 p = f3(__builtin_fmaf((float)(w0 >> 8), 0x1.0p-23f, -1.0f), __builtin_fmaf((float)(w1 >> 8), 0x1.0p-23f, -1.0f),
                  __builtin_fmaf((float)(w2 >> 8), 0x1.0p-23f, -1.0f));
        if (dot(p, p) < 1.0f) return p;
    }
}
// src/util.rs:15-24 (draws x, y per trial)
template <typename RngT>
__device__ __forceinline__ F3 random_in_unit_disk(RngT &g, uint32_t k0, uint32_t k1) {
    for (;;) {
        uint32_t w0, w1;
        rng_take2(g, k0, k1, w0, w1);
        F3 p = f3(__builtin_fmaf((float)(w0 >> 8), 0x1.0p-23f, -1.0f), __builtin_fmaf((float)(w1 >> 8), 0x1.0p-23f, -1.0f), 0.0f);
        if (dot(p, p) < 1.0f) return p;
    }
}
