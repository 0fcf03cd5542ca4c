/* rtmi_math.h — the fp32 transcendental contract of the rtmi device path.
 *
 * The reference computes in f64 and calls Rust std `sin`, `ln`, `atan2`, `asin`
 * (src/texture.rs:41,68; src/medium.rs:40; src/sphere.rs:10-11).  The MI355X path
 * computes in fp32.  A hit/miss decision that flips on a 1-ulp difference changes a
 * pixel by O(emission/spp), so "GPU == fixed-seed CPU result within 1e-4" is only
 * reachable when both sides evaluate these four functions to the SAME bits.  ocml's
 * sinf/logf and glibc's do not agree bit-for-bit, therefore the contract defines them
 * here, built only from operations that are correctly rounded on both x86-64 and
 * gfx950 (+ - * / sqrt fma rint, integer ops) — given `-ffp-contract=off` and no
 * fast-math on either side.  The polynomials are the classic single-precision
 * Cephes minimax fits (public domain, S. Moshier); accuracy is ~1-2 ulp on the
 * ranges the path uses, which is all the contract needs.
 *
 * This header is part of the C-ABI specification (see include/rtmi.h): a host that
 * wants to predict device output bit-for-bit includes it.  It compiles as C99,
 * C++17 and HIP.  It contains no algorithm of the path itself.
 */
#ifndef RTMI_MATH_H
#define RTMI_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RTMI_HD __host__ __device__ __forceinline__
#else
#define RTMI_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

RTMI_HD uint32_t rtmi_f2u(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
RTMI_HD float rtmi_u2f(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

#define RTMI_PI_F 3.1415927410125732f      /* (float)pi   */
#define RTMI_PIO2_F 1.5707963705062866f    /* (float)pi/2 */
#define RTMI_PIO4_F 0.7853981852531433f    /* (float)pi/4 */
#define RTMI_2_OVER_PI_F 0.6366197466850281f /* (float)(2/pi) — also Rust's FRAC_2_PI (sphere.rs:13) */

/* 24-bit uniform in [0,1) from one Philox word: the path's `rng.gen::<f64>()`
 * replacement (reference: 53-bit; tests/test.rs:66, util.rs:8).  Exactly
 * representable in fp32 and f64, so the f32 and f64 oracles see identical draws. */
RTMI_HD float rtmi_u01(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; }

/* sin(x): Cody-Waite reduction by pi/2 (3 constants, fma) + Cephes sinf/cosf kernels.
 * |x| > 2^20 or non-finite returns 0 (the only consumer compares the product of
 * three sines against 0 or feeds 0.5*(1+sin)); identical on both sides by construction. */
RTMI_HD float rtmi_sinf(float x) {
    if (!(__builtin_fabsf(x) <= 1048576.0f)) return 0.0f;
    float k = __builtin_rintf(x * RTMI_2_OVER_PI_F);
    float r = __builtin_fmaf(-k, 0x1.921fb6p+0f, x);
    r = __builtin_fmaf(-k, -0x1.777a5cp-25f, r);
    r = __builtin_fmaf(-k, -0x1.ee59dap-50f, r);
    int q = (int)k;
    float z = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                              4.166664568298827e-2f);
    float c = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}

/* ln(x) for x >= 0 (Cephes logf).  x == 0 -> -inf, which the medium sampler turns into
 * "no scattering event" exactly like the reference's ln(0.0) (medium.rs:40-41). */
RTMI_HD float rtmi_logf(float x) {
    if (x == 0.0f) return rtmi_u2f(0xff800000u);
    if (!(x > 0.0f)) return rtmi_u2f(0x7fc00000u);
    if (x > 3.4028234663852886e38f) return x;
    int eadj = 0;
    if (x < 1.17549435e-38f) {
        x = x * 8388608.0f;
        eadj = -23;
    }
    uint32_t b = rtmi_f2u(x);
    int e = (int)((b >> 23) & 0xffu) - 126 + eadj;
    float m = rtmi_u2f((b & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = p * m * z;
    float fe = (float)e;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

/* atan(x) (Cephes atanf) */
RTMI_HD float rtmi_atanf(float x) {
    int neg = x < 0.0f;
    float a = __builtin_fabsf(x);
    float y0;
    if (a > 2.414213562373095f) {
        y0 = RTMI_PIO2_F;
        a = -(1.0f / a);
    } else if (a > 0.4142135623730950f) {
        y0 = RTMI_PIO4_F;
        a = (a - 1.0f) / (a + 1.0f);
    } else {
        y0 = 0.0f;
    }
    float z = a * a;
    float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    float r = y0 + __builtin_fmaf(p * z, a, a);
    return neg ? -r : r;
}

/* atan2(y,x) with the quadrant rules the sphere UV needs (sphere.rs:10).  -0 is
 * treated as +0 (differs from IEEE atan2 only on the measure-zero negative x axis). */
RTMI_HD float rtmi_atan2f(float y, float x) {
    if (x > 0.0f) return rtmi_atanf(y / x);
    if (x < 0.0f) {
        float t = rtmi_atanf(y / x);
        return (y < 0.0f) ? t - RTMI_PI_F : t + RTMI_PI_F;
    }
    if (y > 0.0f) return RTMI_PIO2_F;
    if (y < 0.0f) return -RTMI_PIO2_F;
    if (x == 0.0f && y == 0.0f) return 0.0f;
    return rtmi_u2f(0x7fc00000u); /* NaN operand */
}

/* asin(x) (Cephes asinf); |x| > 1 -> NaN like Rust's f64::asin (sphere.rs:11). */
RTMI_HD float rtmi_asinf(float x) {
    int neg = x < 0.0f;
    float a = __builtin_fabsf(x);
    if (!(a <= 1.0f)) return rtmi_u2f(0x7fc00000u);
    if (a < 1.0e-4f) return x;
    float z, w;
    int flag = a > 0.5f;
    if (flag) {
        z = 0.5f * (1.0f - a);
        w = __builtin_sqrtf(z);
    } else {
        w = a;
        z = a * a;
    }
    float p = __builtin_fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    float r = __builtin_fmaf(p * z, w, w);
    if (flag) {
        r = r + r;
        r = RTMI_PIO2_F - r;
    }
    return neg ? -r : r;
}

#endif /* RTMI_MATH_H */
