// rtmi_device.hip — gfx950 (MI355X, CDNA4) device path of the per-pixel render loop.
//
// One wavefront (64 lanes) renders one 8x8 pixel tile for one chunk of the sample range:
// lane = pixel.  Every lane runs its own `for s in 0..ns` loop (tests/test.rs:65-70) with
// path regeneration: a lane whose path ended immediately starts its next sample, so all
// 64 lanes stay busy in the bounce loop although path lengths differ (1..51 hit queries,
// src/color.rs:6-23).  The recursion of `color` is unrolled into the throughput form
// L += T*emitted; T *= attenuation.  Random numbers are Philox4x32-10 counter streams
// keyed per (pixel, sample), so the result is independent of tiling, chunking and the
// number of GPUs.  BVH traversal keeps a per-lane stack in LDS (runtime-indexed per-lane
// arrays would spill to scratch).  No MFMA: there is no dense contraction on this path.
//
// Arithmetic follows the fp32 contract of DESIGN.md: op order as written here, no FMA
// contraction (-ffp-contract=off), IEEE / and sqrt, transcendental functions from
// include/rtmi_math.h.  Every device function cites the reference lines it implements.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rtmi.h"
#include "rtmi_math.h"

#define RTMI_FLT_MAX 3.40282346638528859811704183484516925e+38f
#define WAVES_PER_BLOCK 1

// ----------------------------------------------------------------------------------
// small vector type with explicit operation order (nalgebra Vector3 semantics)
// ----------------------------------------------------------------------------------
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 vdiv(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float norm(F3 a) { return __builtin_sqrtf(dot(a, a)); }
__device__ __forceinline__ F3 normalize(F3 a) { return vdiv(a, norm(a)); }
template <int I>
__device__ __forceinline__ float comp(F3 a) {
    return I == 0 ? a.x : (I == 1 ? a.y : a.z);
}

struct DevScene {
    const rtmi_item *items;
    const float4 *prim_a;
    const float4 *prim_b;
    const rtmi_prim_meta *meta;
    const float4 *nodes; // 4 x float4 per rtmi_bvh_node
    const rtmi_xform *xforms;
    const rtmi_material *mats;
    const rtmi_texture *texs;
    const rtmi_perlin *perlin;
    const rtmi_image *images;
    const uint8_t *image_data;
    uint32_t n_items;
};

struct DevCamera {
    F3 origin, llc, horizontal, vertical, u, v;
    float time0, time1, lens_radius;
};

struct DevParams {
    uint32_t nx, ny, ns, max_depth;
    float t_min;
    uint32_t key0, key1;
    uint32_t tile_rank, tile_world, tiles_x, ntiles_local, nchunks;
    unsigned long long *path_sig;
    unsigned long long *prof;
    uint32_t stack_depth;
    uint32_t shade_threshold;
    uint32_t coop_cap;
    unsigned int *status;
};

// ----------------------------------------------------------------------------------
// Philox4x32-10; stream = (block, sample, pixel, 0) under key = seed
// ----------------------------------------------------------------------------------
struct Rng {
    uint32_t block, sample, pixel;
    uint32_t b0, b1, b2, b3;
    uint32_t pos;
};
__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                       uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}
__device__ __forceinline__ void rng_init(Rng &g, uint32_t sample, uint32_t pixel) {
    g.block = 0; g.sample = sample; g.pixel = pixel; g.pos = 4;
}
// rng.gen::<f64>() of the reference (24-bit uniform, rtmi_u01)
__device__ __forceinline__ float rng_uniform(Rng &g, uint32_t k0, uint32_t k1) {
    if (g.pos == 4) {
        philox(g.block, g.sample, g.pixel, 0u, k0, k1, g.b0, g.b1, g.b2, g.b3);
        g.block++;
        g.pos = 0;
    }
    uint32_t w = g.pos == 0 ? g.b0 : (g.pos == 1 ? g.b1 : (g.pos == 2 ? g.b2 : g.b3));
    g.pos++;
    return rtmi_u01(w);
}

// The next THREE (resp. TWO) consecutive words of the stream with at most ONE Philox evaluation
// for the whole wavefront.  Calling rng_uniform three times evaluates Philox up to three times per
// wavefront (lanes sit at different positions of their 4-word blocks, so at every call some lane
// needs a refill and the others wait).  Same stream, same words, same order: bit-identical.
__device__ __forceinline__ void rng_take3(Rng &g, uint32_t k0, uint32_t k1, uint32_t &w0, uint32_t &w1, uint32_t &w2) {
    uint32_t n0 = 0u, n1 = 0u, n2 = 0u, n3 = 0u;
    const uint32_t pos = g.pos;
    if (pos >= 2u) { // fewer than three words left in the current block
        philox(g.block, g.sample, g.pixel, 0u, k0, k1, n0, n1, n2, n3);
        g.block++;
    }
    w0 = pos == 0u ? g.b0 : (pos == 1u ? g.b1 : (pos == 2u ? g.b2 : (pos == 3u ? g.b3 : n0)));
    w1 = pos == 0u ? g.b1 : (pos == 1u ? g.b2 : (pos == 2u ? g.b3 : (pos == 3u ? n0 : n1)));
    w2 = pos == 0u ? g.b2 : (pos == 1u ? g.b3 : (pos == 2u ? n0 : (pos == 3u ? n1 : n2)));
    if (pos >= 2u) { g.b0 = n0; g.b1 = n1; g.b2 = n2; g.b3 = n3; g.pos = pos - 1u; } // 2->1, 3->2, 4->3
    else g.pos = pos + 3u;
}
__device__ __forceinline__ void rng_take2(Rng &g, uint32_t k0, uint32_t k1, uint32_t &w0, uint32_t &w1) {
    uint32_t n0 = 0u, n1 = 0u, n2 = 0u, n3 = 0u;
    const uint32_t pos = g.pos;
    if (pos >= 3u) {
        philox(g.block, g.sample, g.pixel, 0u, k0, k1, n0, n1, n2, n3);
        g.block++;
    }
    w0 = pos == 0u ? g.b0 : (pos == 1u ? g.b1 : (pos == 2u ? g.b2 : (pos == 3u ? g.b3 : n0)));
    w1 = pos == 0u ? g.b1 : (pos == 1u ? g.b2 : (pos == 2u ? g.b3 : (pos == 3u ? n0 : n1)));
    if (pos >= 3u) { g.b0 = n0; g.b1 = n1; g.b2 = n2; g.b3 = n3; g.pos = pos - 2u; } // 3->1, 4->2
    else g.pos = pos + 2u;
}

// src/util.rs:4-13 (draws x, y, z per trial)
__device__ __forceinline__ F3 random_in_unit_sphere(Rng &g, uint32_t k0, uint32_t k1) {
    for (;;) {
        uint32_t w0, w1, w2;
        rng_take3(g, k0, k1, w0, w1, w2);
        const float x = rtmi_u01(w0), y = rtmi_u01(w1), z = rtmi_u01(w2);
        F3 p = f3(2.0f * x - 1.0f, 2.0f * y - 1.0f, 2.0f * z - 1.0f);
        if (dot(p, p) < 1.0f) return p;
    }
}
// src/util.rs:15-24 (draws x, y per trial)
__device__ __forceinline__ F3 random_in_unit_disk(Rng &g, uint32_t k0, uint32_t k1) {
    for (;;) {
        uint32_t w0, w1;
        rng_take2(g, k0, k1, w0, w1);
        const float x = rtmi_u01(w0), y = rtmi_u01(w1);
        F3 p = f3(2.0f * x - 1.0f, 2.0f * y - 1.0f, 0.0f);
        if (dot(p, p) < 1.0f) return p;
    }
}

// ----------------------------------------------------------------------------------
// instance transforms — src/traslate.rs:18-24, src/rotate.rs:85-113
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void rot_fwd(float s, float c, float &a, float &b) {
    float na = c * a + s * b;
    float nb = -s * a + c * b;
    a = na; b = nb;
}
__device__ __forceinline__ void rot_inv(float s, float c, float &a, float &b) {
    float na = c * a - s * b;
    float nb = s * a + c * b;
    a = na; b = nb;
}
// world -> object; returns true when the direction changed (a rotation was applied)
__device__ __forceinline__ bool xform_ray(const rtmi_xform *xf, int first, int count, F3 &o, F3 &d) {
    bool rotated = false;
    for (int k = 0; k < count; k++) {
        const rtmi_xform X = xf[first + k];
        switch (X.kind) {
        case RTMI_XF_TRANSLATE: o = o - f3(X.x, X.y, X.z); break;
        case RTMI_XF_ROTATE_X: rot_fwd(X.x, X.y, o.y, o.z); rot_fwd(X.x, X.y, d.y, d.z); rotated = true; break;
        case RTMI_XF_ROTATE_Y: rot_fwd(X.x, X.y, o.z, o.x); rot_fwd(X.x, X.y, d.z, d.x); rotated = true; break;
        default: rot_fwd(X.x, X.y, o.x, o.y); rot_fwd(X.x, X.y, d.x, d.y); rotated = true; break;
        }
    }
    return rotated;
}
// object -> world for the hit point and normal (innermost wrapper first)
__device__ __forceinline__ void xform_hit(const rtmi_xform *xf, int first, int count, F3 &p, F3 &n) {
    for (int k = count - 1; k >= 0; k--) {
        const rtmi_xform X = xf[first + k];
        switch (X.kind) {
        case RTMI_XF_TRANSLATE: p = p + f3(X.x, X.y, X.z); break;
        case RTMI_XF_ROTATE_X: rot_inv(X.x, X.y, p.y, p.z); rot_inv(X.x, X.y, n.y, n.z); break;
        case RTMI_XF_ROTATE_Y: rot_inv(X.x, X.y, p.z, p.x); rot_inv(X.x, X.y, n.z, n.x); break;
        default: rot_inv(X.x, X.y, p.x, p.y); rot_inv(X.x, X.y, n.x, n.y); break;
        }
    }
}

// ----------------------------------------------------------------------------------
// intersectors
// ----------------------------------------------------------------------------------
struct RayF { // a ray in one frame, with the per-frame derived values
    F3 o, d, inv_d;
    float a, inv_a; // d.d and 1/(d.d)  (sphere.rs:40, contract: t = (-b -+ sqrt)*inv_a)
};
__device__ __forceinline__ void ray_derive(RayF &r) {
    r.inv_d = f3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z); // aabb.rs:33
    r.a = dot(r.d, r.d);
    r.inv_a = 1.0f / r.a;
}

// AABB::hit — src/aabb.rs:31-44.  The sequential early-out is an OR of the three tests.
__device__ __forceinline__ bool aabb_hit(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                         const RayF &r, float t_min, float t_max) {
    float t0 = (mnx - r.o.x) * r.inv_d.x, t1 = (mxx - r.o.x) * r.inv_d.x;
    bool neg = r.inv_d.x < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    bool fail = t_max <= t_min;
    t0 = (mny - r.o.y) * r.inv_d.y; t1 = (mxy - r.o.y) * r.inv_d.y;
    neg = r.inv_d.y < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    fail |= t_max <= t_min;
    t0 = (mnz - r.o.z) * r.inv_d.z; t1 = (mxz - r.o.z) * r.inv_d.z;
    neg = r.inv_d.z < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    fail |= t_max <= t_min;
    return !fail;
}

// Sphere::hit / MovingSphere::hit — src/sphere.rs:37-77, 122-164 (t only; the record is
// built once for the closest hit in finalize_hit)
__device__ __forceinline__ bool sphere_test(const RayF &r, F3 c, float radius, float t_min, float t_max, float &t_out) {
    F3 oc = r.o - c;
    float b = dot(oc, r.d);
    float cc = dot(oc, oc) - radius * radius;
    float disc = b * b - r.a * cc;
    if (disc > 0.0f) {
        float sq = __builtin_sqrtf(disc);
        float t = (-b - sq) * r.inv_a;
        if (t < t_max && t > t_min) { t_out = t; return true; }
        t = (-b + sq) * r.inv_a;
        if (t < t_max && t > t_min) { t_out = t; return true; }
    }
    return false;
}
// MovingSphere::center — src/sphere.rs:115-118 (contract: (time - t0) * inv_dt)
__device__ __forceinline__ F3 moving_center(float4 A, float4 B, float inv_dt, float time) {
    float f = (time - B.w) * inv_dt;
    return f3(A.x, A.y, A.z) + f3(B.x, B.y, B.z) * f;
}

// Rect::hit — src/rect.rs:39-69 with (k,a,b) = YZ:(0,1,2) ZX:(1,2,0) XY:(2,0,1)
template <int P>
__device__ __forceinline__ bool rect_test(float x0, float y0, float x1, float y1, float k, const RayF &r, float t_min,
                                          float t_max, float &t_out) {
    constexpr int K = P == 0 ? 0 : (P == 1 ? 1 : 2);
    constexpr int A = P == 0 ? 1 : (P == 1 ? 2 : 0);
    constexpr int B = P == 0 ? 2 : (P == 1 ? 0 : 1);
    float t = (k - comp<K>(r.o)) * comp<K>(r.inv_d);
    if (t < t_min || t > t_max) return false;
    float x = comp<A>(r.o) + t * comp<A>(r.d);
    float y = comp<B>(r.o) + t * comp<B>(r.d);
    if (x < x0 || x > x1 || y < y0 || y > y1) return false;
    t_out = t;
    return true;
}
__device__ __forceinline__ bool rect_test_rt(int plane, float4 A, float k, const RayF &r, float t_min, float t_max,
                                             float &t_out) {
    if (plane == 0) return rect_test<0>(A.x, A.y, A.z, A.w, k, r, t_min, t_max, t_out);
    if (plane == 1) return rect_test<1>(A.x, A.y, A.z, A.w, k, r, t_min, t_max, t_out);
    return rect_test<2>(A.x, A.y, A.z, A.w, k, r, t_min, t_max, t_out);
}
// Cube::hit — src/cube.rs:84-86: HittableList scan (hittable.rs:37-47) of the six rects in
// construction order (cube.rs:21-74); a later face wins a tie because Rect accepts t == t_max.
__device__ __forceinline__ bool cube_test(float4 A, float4 B, const RayF &r, float t_min, float t_max, float &t_out,
                                          int &face) {
    const float ax = A.x, ay = A.y, az = A.z, bx = A.w, by = B.x, bz = B.y;
    float cl = t_max, t;
    bool any = false;
    if (rect_test<2>(ax, ay, bx, by, bz, r, t_min, cl, t)) { cl = t; any = true; face = 0; }
    if (rect_test<2>(ax, ay, bx, by, az, r, t_min, cl, t)) { cl = t; any = true; face = 1; }
    if (rect_test<1>(az, ax, bz, bx, by, r, t_min, cl, t)) { cl = t; any = true; face = 2; }
    if (rect_test<1>(az, ax, bz, bx, ay, r, t_min, cl, t)) { cl = t; any = true; face = 3; }
    if (rect_test<0>(ay, az, by, bz, bx, r, t_min, cl, t)) { cl = t; any = true; face = 4; }
    if (rect_test<0>(ay, az, by, bz, ax, r, t_min, cl, t)) { cl = t; any = true; face = 5; }
    t_out = cl;
    return any;
}

// one primitive against (t_min, t_max); pf = prim << 3 | face.  The three planes are loaded up
// front (independent addresses): one memory latency instead of up to three dependent ones.
__device__ __forceinline__ bool prim_test(const DevScene &sc, int type, int idx, const RayF &r, float time,
                                          float t_min, float t_max, float &t_out, int &pf) {
    const float4 A = sc.prim_a[idx];
    const float4 B = sc.prim_b[idx];
    const rtmi_prim_meta M = sc.meta[idx];
    bool h = false;
    int face = 0;
    if (type == RTMI_PRIM_SPHERE) {
        h = sphere_test(r, f3(A.x, A.y, A.z), A.w, t_min, t_max, t_out);
    } else if (type == RTMI_PRIM_MSPHERE) {
        h = sphere_test(r, moving_center(A, B, M.inv_dt, time), A.w, t_min, t_max, t_out);
    } else if (type == RTMI_PRIM_RECT) {
        const int plane = (int)((M.flags >> RTMI_PRIMFLAG_PLANE_SHIFT) & 3u);
        h = rect_test_rt(plane, A, B.x, r, t_min, t_max, t_out);
    } else {
        h = cube_test(A, B, r, t_min, t_max, t_out, face);
    }
    pf = (idx << 3) | face;
    return h;
}

// ---- lane-activity profiling (PROF instantiation only; diagnostics, never on the timed path) -----
// slot s: prof[2s] += active lanes, prof[2s+1] += 64 (one wave-iteration).  Accumulated in LDS,
// flushed to global once per block.
#define RTMI_PROF_SLOTS 32
template <bool PROF>
__device__ __forceinline__ void prof_tick(unsigned long long *prof_lds, int slot, bool active) {
    if (PROF) {
        const unsigned long long m = __ballot(active);
        if (m != 0ull && (int)(__ffsll((long long)m) - 1) == (int)(threadIdx.x & 63)) {
            atomicAdd(&prof_lds[2 * slot], (unsigned long long)__popcll(m));
            atomicAdd(&prof_lds[2 * slot + 1], 64ull);
        }
    }
}

// section timing (PROF only): elapsed shader cycles of this wave since the previous stamp are added
// to slot `slot` (word 0 = cycles, word 1 = number of stamps)
template <bool PROF>
__device__ __forceinline__ void prof_time(unsigned long long *prof_lds, int slot, unsigned long long &t_prev) {
    if (PROF) {
        const unsigned long long t = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&prof_lds[2 * slot], t - t_prev);
            atomicAdd(&prof_lds[2 * slot + 1], 1ull);
        }
        t_prev = __builtin_readcyclecounter();
    }
}

// AABB::hit as above, additionally returning the entry distance max(t_min, near slabs).
__device__ __forceinline__ bool aabb_hit_t(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                           const RayF &r, float t_min, float t_max, float &t_enter) {
    float t0 = (mnx - r.o.x) * r.inv_d.x, t1 = (mxx - r.o.x) * r.inv_d.x;
    bool neg = r.inv_d.x < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    bool fail = t_max <= t_min;
    t0 = (mny - r.o.y) * r.inv_d.y; t1 = (mxy - r.o.y) * r.inv_d.y;
    neg = r.inv_d.y < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    fail |= t_max <= t_min;
    t0 = (mnz - r.o.z) * r.inv_d.z; t1 = (mxz - r.o.z) * r.inv_d.z;
    neg = r.inv_d.z < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    fail |= t_max <= t_min;
    t_enter = t_min;
    return !fail;
}

// BVHNode::hit — src/bvh.rs:70-89, iteratively.
//
// EXACT (FAST = false): children visited left before right, so folding leaf hits with
// "replace unless best.t < t" reproduces the pairwise `l.t < r.t ? l : r` (tie -> right).
// Every box and every leaf is tested against the query's own (t_min, t_max), as in the
// reference; a leaf child has no box test of its own (bvh.rs:72-73).
//
// FAST: the same result with fewer visits.  (1) the nearer child first; (2) a subtree is
// skipped when its box is entered later than the best hit so far plus a generous margin
// (no primitive inside can then beat or tie it); (3) a leaf child is skipped when the ray
// misses its PADDED box (stored by the lowering).  The winner among equal t is the
// primitive that is rightmost in the tree = the largest primitive index (leaves are stored
// left to right), which is what the fold above yields.  Internal boxes are still tested
// against (t_min, t_max) with the reference's own arithmetic, so they prune identically.
// stack: this lane's LDS column (node refs); stack + 64*RTMI_MAX_BVH_DEPTH: entry distances.
template <bool FAST, bool PROF>
__device__ __forceinline__ bool bvh_query(const DevScene &sc, int root, float scale, const RayF &r, float time,
                                          float t_min, float t_max, uint32_t *stack, float &t_out, int &pf_out,
                                          unsigned long long *prof, int slot) {
    bool have = false;
    float bt = FAST ? RTMI_FLT_MAX : 0.0f;
    int bpf = 0;
    int sp = 0;
    int cur = root;
    float *stack_t = reinterpret_cast<float *>(stack + 64 * RTMI_MAX_BVH_DEPTH);
    // prune when t_enter > bt + |bt|/128 + scale/8192/|d|
    const float m_abs = FAST ? scale * (1.0f / 8192.0f) * __builtin_sqrtf(r.inv_a) : 0.0f;
    float limit = RTMI_FLT_MAX;
    for (;;) {
        prof_tick<PROF>(prof, slot, true);          // lanes alive in this traversal iteration
        prof_tick<PROF>(prof, 14, cur >= 0);        // ... of which at an internal node
        prof_tick<PROF>(prof, 15, cur < 0);         // ... of which at a leaf
        if (cur >= 0) {
            const float4 *n = sc.nodes + (size_t)cur * 4;
            const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            const int left = __float_as_int(n3.x), right = __float_as_int(n3.y);
            if (!FAST) {
                bool vl = left < 0 || aabb_hit(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, r, t_min, t_max);
                bool vr = right < 0 || aabb_hit(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, r, t_min, t_max);
                if (right == left) vr = false; // BVHNode over one object: the same leaf twice, same result
                if (vl) {
                    if (vr) { stack[sp * 64] = (uint32_t)right; sp++; }
                    cur = left;
                    continue;
                }
                if (vr) { cur = right; continue; }
            } else {
                float tl, tr;
                bool vl = aabb_hit_t(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, r, t_min, t_max, tl);
                bool vr = aabb_hit_t(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, r, t_min, t_max, tr);
                vl = vl && !(tl > limit);
                vr = vr && !(tr > limit) && right != left;
                if (vl && vr) {
                    const bool lfirst = !(tr < tl);
                    stack[sp * 64] = (uint32_t)(lfirst ? right : left);
                    stack_t[sp * 64] = lfirst ? tr : tl;
                    sp++;
                    cur = lfirst ? left : right;
                    continue;
                }
                if (vl) { cur = left; continue; }
                if (vr) { cur = right; continue; }
            }
        } else {
            const int type = (int)(((uint32_t)cur >> 28) & 7u);
            const int idx = (int)((uint32_t)cur & 0x0fffffffu);
            float t;
            int pf;
            if (prim_test(sc, type, idx, r, time, t_min, t_max, t, pf)) {
                if (!FAST) {
                    if (!have || !(bt < t)) { bt = t; bpf = pf; have = true; }
                } else {
                    if (!have || t < bt || (t == bt && pf > bpf)) {
                        bt = t; bpf = pf; have = true;
                        limit = bt + (__builtin_fabsf(bt) * (1.0f / 128.0f) + m_abs);
                    }
                }
            }
        }
        if (!FAST) {
            if (sp == 0) break;
            sp--;
            cur = (int)stack[sp * 64];
        } else {
            bool got = false;
            while (sp > 0) {
                sp--;
                if (!(stack_t[sp * 64] > limit)) { cur = (int)stack[sp * 64]; got = true; break; }
            }
            if (!got) break;
        }
    }
    t_out = bt;
    pf_out = bpf;
    return have;
}

// geometry of one item against (q_min, q_max): HittableList scan or BVH
template <bool FAST, bool PROF>
__device__ __forceinline__ bool geom_query(const DevScene &sc, const rtmi_item &I, const RayF &r, float time,
                                           float q_min, float q_max, uint32_t *stack, float &t_out,
                                           int &pf_out, unsigned long long *prof, int slot) {
    if (I.kind == RTMI_ITEM_BVH) {
        // BVHNode::hit of the root: its own bbox first (bvh.rs:71)
        if (!aabb_hit(I.root_min[0], I.root_min[1], I.root_min[2], I.root_max[0], I.root_max[1], I.root_max[2], r,
                      q_min, q_max))
            return false;
        return bvh_query<FAST, PROF>(sc, I.first, I.scale, r, time, q_min, q_max, stack, t_out, pf_out, prof, slot);
    }
    // HittableList::hit — hittable.rs:37-47
    float cl = q_max;
    bool any = false;
    for (int k = 0; k < I.count; k++) {
        const int idx = I.first + k;
        const int type = sc.meta[idx].type;
        float t;
        int pf;
        prof_tick<PROF>(prof, 13, true);            // list primitive tests
        if (prim_test(sc, type, idx, r, time, q_min, cl, t, pf)) { cl = t; any = true; pf_out = pf; }
    }
    t_out = cl;
    return any;
}

// ----------------------------------------------------------------------------------
// wave-cooperative BVH traversal (FAST semantics; RTMI_FLAG_COOP)
//
// Measured with per-lane traversal: only 5-13 % of the lanes are active per traversal iteration
// — a few rays walk long while the others have left the tree or never entered it.  Here the 64
// lanes of the wavefront are WORKERS on a wave-shared LIFO of (ray, node) entries in LDS:
//   * every lane that owns a ray entering the tree publishes its ray context in LDS and pushes
//     the root; then all 64 lanes, owners or not, pop entries and process them;
//   * a worker that processed a node keeps the nearer surviving child itself (depth-first, so
//     its ray context stays in registers) and pushes the farther one for anybody to take;
//   * a leaf hit is folded into the ray's best hit with ONE LDS atomicMin on the 64-bit key
//     (order-preserving bits of t, inverted primitive id): minimum t, ties -> the larger
//     primitive index = the rightmost leaf, exactly the fold of BVHNode::hit (bvh.rs:75-81).
//     The fold is order-independent, so the result does not depend on who processes what.
// Pruning is the fast-cull rule (subtree entered later than the ray's best hit + margin);
// internal boxes are tested against the query's own (t_min, t_max) with the reference's
// arithmetic.  LIFO order makes workers take the deepest pending entries first, which bounds
// the pool by 64 * (tree depth + 1) entries; the caller reports an overflow loudly.
// ----------------------------------------------------------------------------------
#define COOP_NONE 0xffffffffu
#define COOP_SENTINEL 0xffffffffffffffffull
__device__ __forceinline__ uint32_t f2sort(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sort2f(uint32_t s) {
    return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s);
}
// node child reference (rtmi_bvh_node.left/right) -> 26-bit pool encoding
__device__ __forceinline__ uint32_t coop_enc(int ref) {
    if (ref >= 0) return (uint32_t)ref;
    const uint32_t u = (uint32_t)ref;
    return (1u << 25) | (((u >> 28) & 7u) << 22) | (u & 0x003fffffu);
}

// All 64 lanes must call this together.  LDS layout for this wave (uint32 words):
//   pool [cap][2] | ctx [64][12] floats | best [64] uint64
template <bool PROF>
__device__ __forceinline__ void coop_bvh_query(const DevScene &sc, int root, float scale, bool active, const RayF &R,
                                               float time, float q_min, float q_max, uint32_t *wlds, int cap,
                                               bool &have, float &t_out, int &pf_out, bool &overflow,
                                               unsigned long long *prof, int slot) {
    const int lane = threadIdx.x & 63;
    volatile uint32_t *pool = wlds;
    float4 *ctx = reinterpret_cast<float4 *>(wlds + 2 * cap);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(wlds + 2 * cap + 64 * 12);

    const unsigned long long m_act = __ballot(active);
    have = false;
    if (m_act == 0ull) return; // wave-uniform: nobody enters this tree
    // ---- publish ray contexts, push roots
    best[lane] = COOP_SENTINEL;
    if (active) {
        // 48 B per ray; a, inv_a and the pruning margin are recomputed by the worker with the same
        // operations on the same values (bit-identical), which keeps the wave's LDS under 10 KB
        ctx[lane * 3 + 0] = make_float4(R.o.x, R.o.y, R.o.z, time);
        ctx[lane * 3 + 1] = make_float4(R.d.x, R.d.y, R.d.z, q_min);
        ctx[lane * 3 + 2] = make_float4(R.inv_d.x, R.inv_d.y, R.inv_d.z, q_max);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_act, 0u));
        pool[2 * pos] = ((uint32_t)lane << 26) | (uint32_t)root;
        pool[2 * pos + 1] = __float_as_uint(q_min); // entry distance of the root: conservative
    }
    int top = __popcll(m_act);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    uint32_t cur = COOP_NONE; // 26-bit node/leaf encoding of the entry this worker holds
    int ray = 0, cray = -1;
    float tent = 0.0f;
    RayF W;                   // context of ray `cray`
    W.o = f3(0, 0, 0); W.d = f3(0, 0, 1); W.inv_d = f3(0, 0, 0); W.a = 1.0f; W.inv_a = 1.0f;
    float wtime = 0.0f, wqmin = 0.0f, wqmax = 0.0f, wmabs = 0.0f;

    for (;;) {
        // ---- idle workers take the deepest pending entries
        const bool needw = cur == COOP_NONE;
        const unsigned long long m_need = __ballot(needw);
        const int n_need = __popcll(m_need);
        if (top == 0 && n_need == 64) break;
        if (top > cap - 64) { overflow = true; break; }
        const int take = n_need < top ? n_need : top;
        if (needw) {
            const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_need, 0u));
            if (r < take) {
                const int idx = top - 1 - r;
                const uint32_t e0 = pool[2 * idx];
                tent = __uint_as_float(pool[2 * idx + 1]);
                cur = e0 & 0x03ffffffu;
                ray = (int)(e0 >> 26);
            }
        }
        top -= take;
        prof_tick<PROF>(prof, slot, cur != COOP_NONE);
        bool push = false;
        uint32_t push_ref = 0u;
        float push_t = 0.0f;
        if (cur != COOP_NONE) {
            if (ray != cray) { // switch ray context
                const float4 c0 = ctx[ray * 3 + 0], c1 = ctx[ray * 3 + 1], c2 = ctx[ray * 3 + 2];
                W.o = f3(c0.x, c0.y, c0.z); wtime = c0.w;
                W.d = f3(c1.x, c1.y, c1.z); wqmin = c1.w;
                W.inv_d = f3(c2.x, c2.y, c2.z); wqmax = c2.w;
                W.a = dot(W.d, W.d);
                W.inv_a = 1.0f / W.a;
                wmabs = scale * (1.0f / 8192.0f) * __builtin_sqrtf(W.inv_a);
                cray = ray;
            }
            // the ray's best hit so far -> pruning limit
            const unsigned long long key = *reinterpret_cast<volatile unsigned long long *>(&best[ray]);
            float limit = RTMI_FLT_MAX;
            if (key != COOP_SENTINEL) {
                const float bt = sort2f((uint32_t)(key >> 32));
                limit = bt + (__builtin_fabsf(bt) * (1.0f / 128.0f) + wmabs);
            }
            if (tent > limit) {
                cur = COOP_NONE;
            } else if (!(cur & (1u << 25))) { // internal node
                const float4 *n = sc.nodes + (size_t)cur * 4;
                const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
                const int left = __float_as_int(n3.x), right = __float_as_int(n3.y);
                float tl, tr;
                bool vl = aabb_hit_t(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, W, wqmin, wqmax, tl);
                bool vr = aabb_hit_t(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, W, wqmin, wqmax, tr);
                vl = vl && !(tl > limit);
                vr = vr && !(tr > limit) && right != left;
                if (vl && vr) {
                    const bool lfirst = !(tr < tl);
                    push = true;
                    push_ref = coop_enc(lfirst ? right : left);
                    push_t = lfirst ? tr : tl;
                    cur = coop_enc(lfirst ? left : right);
                    tent = lfirst ? tl : tr;
                } else if (vl) { cur = coop_enc(left); tent = tl; }
                else if (vr) { cur = coop_enc(right); tent = tr; }
                else cur = COOP_NONE;
            } else { // leaf
                const int type = (int)((cur >> 22) & 7u);
                const int idx = (int)(cur & 0x003fffffu);
                float t;
                int pf;
                if (prim_test(sc, type, idx, W, wtime, wqmin, wqmax, t, pf)) {
                    const unsigned long long k = ((unsigned long long)f2sort(t) << 32) | (unsigned long long)(0x7fffffffu - (uint32_t)pf);
                    atomicMin(&best[ray], k);
                }
                cur = COOP_NONE;
            }
        }
        // ---- publish the far children
        const unsigned long long m_push = __ballot(push);
        if (push) {
            const int pos = top + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_push >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_push, 0u));
            pool[2 * pos] = ((uint32_t)ray << 26) | push_ref;
            pool[2 * pos + 1] = __float_as_uint(push_t);
        }
        top += __popcll(m_push);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (active) {
        const unsigned long long key = *reinterpret_cast<volatile unsigned long long *>(&best[lane]);
        if (key != COOP_SENTINEL) {
            have = true;
            t_out = sort2f((uint32_t)(key >> 32));
            pf_out = (int)(0x7fffffffu - (uint32_t)key);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// geometry of one item for the whole wavefront: every lane calls it; `active` lanes own a query
template <bool PROF>
__device__ __forceinline__ bool geom_query_coop(const DevScene &sc, const rtmi_item &I, bool active, const RayF &r,
                                                float time, float q_min, float q_max, uint32_t *wlds, int cap,
                                                float &t_out, int &pf_out, bool &overflow, unsigned long long *prof,
                                                int slot) {
    if (I.kind == RTMI_ITEM_BVH) { // wave-uniform branch
        // BVHNode::hit of the root: its own bbox first (bvh.rs:71)
        const bool enter = active && aabb_hit(I.root_min[0], I.root_min[1], I.root_min[2], I.root_max[0], I.root_max[1],
                                              I.root_max[2], r, q_min, q_max);
        bool have = false;
        coop_bvh_query<PROF>(sc, I.first, I.scale, enter, r, time, q_min, q_max, wlds, cap, have, t_out, pf_out, overflow,
                             prof, slot);
        return have;
    }
    // HittableList::hit — hittable.rs:37-47
    float cl = q_max;
    bool any = false;
    if (active) {
        for (int k = 0; k < I.count; k++) {
            const int idx = I.first + k;
            const int type = sc.meta[idx].type;
            float t;
            int pf;
            prof_tick<PROF>(prof, 13, true);
            if (prim_test(sc, type, idx, r, time, q_min, cl, t, pf)) { cl = t; any = true; pf_out = pf; }
        }
    }
    t_out = cl;
    return any;
}

// ----------------------------------------------------------------------------------
// textures — src/texture.rs, src/perlin.rs
// ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t as_usize_u32(float x) { // Rust `as usize`, see DESIGN.md
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0u; // fp32 values >= 2^32 are multiples of 512: low 8 bits are 0
    return (uint32_t)x;
}
// Perlin::noise + perlin_interpolation — perlin.rs:76-97, 38-56
__device__ __forceinline__ float perlin_noise(const rtmi_perlin *pn, F3 p) {
    const float fx = __builtin_floorf(p.x), fy = __builtin_floorf(p.y), fz = __builtin_floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const uint32_t i = as_usize_u32(fx), j = as_usize_u32(fy), k = as_usize_u32(fz);
    const float uu = u * u * (3.0f - 2.0f * u);
    const float vv = v * v * (3.0f - 2.0f * v);
    const float ww = w * w * (3.0f - 2.0f * w);
    const float4 *rv = reinterpret_cast<const float4 *>(pn->ranvec);
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
        for (int dj = 0; dj < 2; dj++)
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                const int h = pn->perm[(i + di) & 255u] ^ pn->perm[256 + ((j + dj) & 255u)] ^
                              pn->perm[512 + ((k + dk) & 255u)];
                const float4 c = rv[h];
                const float wx = u - (float)di, wy = v - (float)dj, wz = w - (float)dk;
                const float fi = di ? uu : (1.0f - uu); // i*uu + (1-i)*(1-uu) with i in {0,1}
                const float fj = dj ? vv : (1.0f - vv);
                const float fk = dk ? ww : (1.0f - ww);
                accum += fi * fj * fk * (c.x * wx + c.y * wy + c.z * wz);
            }
    return accum;
}
// Perlin::turb — perlin.rs:99-109
__device__ __forceinline__ float perlin_turb(const rtmi_perlin *pn, F3 p, int depth) {
    float accum = 0.0f, weight = 1.0f;
    F3 tp = p;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, tp);
        weight *= 0.5f;
        tp = tp * 2.0f;
    }
    return __builtin_fabsf(accum);
}
// Texture::value — texture.rs:21-25 (Solid), :39-48 (Checker), :65-71 (Noise), :86-108 (Image)
__device__ __forceinline__ F3 tex_value(const DevScene &sc, int tex, float u, float v, F3 p) {
    rtmi_texture t = sc.texs[tex];
    for (int guard = 0; guard < 16 && t.kind == RTMI_TEX_CHECKER; guard++) {
        const float s = rtmi_sinf(10.0f * p.x) * rtmi_sinf(10.0f * p.y) * rtmi_sinf(10.0f * p.z);
        t = sc.texs[s < 0.0f ? t.i0 : t.i1];
    }
    if (t.kind == RTMI_TEX_NOISE) {
        const float g = 0.5f * (1.0f + rtmi_sinf(t.f0 * p.x + 5.0f * perlin_turb(sc.perlin + t.i0, p, 7)));
        return f3(g, g, g);
    }
    if (t.kind == RTMI_TEX_IMAGE) {
        const rtmi_image im = sc.images[t.i0];
        uint32_t i = as_usize_u32(u * (float)im.nx);
        uint32_t j = as_usize_u32((1.0f - v) * (float)im.ny);
        if (i > im.nx - 1) i = im.nx - 1;
        if (j > im.ny - 1) j = im.ny - 1;
        const uint8_t *px = sc.image_data + im.offset + 3ull * i + 3ull * im.nx * j;
        return f3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    }
    return f3(t.f0, t.f1, t.f2);
}

// get_sphere_uv — sphere.rs:9-15 (FRAC_2_PI, sic)
__device__ __forceinline__ void sphere_uv(F3 n, float &u, float &v) {
    const float phi = rtmi_atan2f(n.z, n.x);
    const float theta = rtmi_asinf(n.y);
    u = 1.0f - (phi + RTMI_PI_F) / (2.0f * RTMI_PI_F);
    v = (theta + RTMI_2_OVER_PI_F) / RTMI_PI_F;
}

// ----------------------------------------------------------------------------------
// materials — src/material.rs
// ----------------------------------------------------------------------------------
__device__ __forceinline__ F3 reflect(F3 v, F3 n) { // material.rs:9-11
    const float s = 2.0f * dot(v, n);
    return v - n * s;
}
__device__ __forceinline__ bool refract(F3 v, F3 n, float ni_over_nt, F3 &out) { // material.rs:13-23
    const F3 uv = normalize(v);
    const float dt = dot(uv, n);
    const float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (disc > 0.0f) {
        out = (uv - n * dt) * ni_over_nt - n * __builtin_sqrtf(disc);
        return true;
    }
    return false;
}
__device__ __forceinline__ float schlick(float cosine, float ref_idx) { // material.rs:25-28
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    const float x = 1.0f - cosine;
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);
}

// ----------------------------------------------------------------------------------
// path pieces shared by the render kernels
// ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sig_mix(uint32_t x, uint32_t k) {
    x ^= (k + 1u) * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

struct Path { // one camera path in flight (per lane)
    F3 ro, rd;
    float rtime;
    F3 T, L;
    uint32_t depth;
};

// next sample of this pixel: tests/test.rs:66-68 + Camera::get_ray (camera.rs:53-67)
__device__ __forceinline__ void camera_sample(const DevCamera &cam, const DevParams &P, Rng &g, uint32_t k0, uint32_t k1,
                                              uint32_t s, uint32_t pixel, uint32_t px, uint32_t j, Path &pa) {
    rng_init(g, s, pixel);
    uint32_t wu, wv;
    rng_take2(g, k0, k1, wu, wv); // u then v — tests/test.rs:66-67
    const float u = ((float)px + rtmi_u01(wu)) / (float)P.nx;
    const float v = ((float)j + rtmi_u01(wv)) / (float)P.ny;
    F3 origin = cam.origin;
    if (cam.lens_radius != 0.0f) {
        const F3 rdk = random_in_unit_disk(g, k0, k1) * cam.lens_radius;
        const F3 offset = cam.u * rdk.x + cam.v * rdk.y;
        origin = cam.origin + offset;
    }
    pa.rtime = cam.time0 + rng_uniform(g, k0, k1) * (cam.time1 - cam.time0);
    pa.ro = origin;
    pa.rd = cam.llc + cam.horizontal * u + cam.vertical * v - origin;
    pa.T = f3(1, 1, 1);
    pa.L = f3(0, 0, 0);
    pa.depth = 0;
}

// ConstantMedium::hit after both boundary queries — medium.rs:33-53.  Returns true when the
// medium scatters before the boundary exit / the closest hit so far; the draw happens only when
// the clamped interval is non-empty, as in the reference.
__device__ __forceinline__ bool medium_sample(float t1, float t2, float t_min, float closest, F3 world_d,
                                              float neg_inv_density, Rng &g, uint32_t k0, uint32_t k1, float &t_out) {
    if (t1 < t_min) t1 = t_min;
    if (t2 > closest) t2 = closest;
    if (t1 < t2) {
        const float dn = norm(world_d);
        const float dist_inside = (t2 - t1) * dn;
        const float hit_distance = neg_inv_density * rtmi_logf(rng_uniform(g, k0, k1));
        if (hit_distance < dist_inside) {
            t_out = t1 + hit_distance / dn;
            return true;
        }
    }
    return false;
}

// HitRecord of the closest hit (hittable.rs:9-16), built once, then
// color(): emitted + attenuation * color(scattered) — color.rs:8-15, in throughput form.
// Returns true when the path continues (pa holds the scattered ray), false when it ended.
__device__ __forceinline__ bool shade_hit(const DevScene &sc, uint32_t max_depth, Rng &g, uint32_t k0, uint32_t k1,
                                          float closest, int best_item, int best_pf, bool best_medium, Path &pa) {
    const rtmi_item I = sc.items[best_item];
    F3 hp, hn;
    float hu = 0.0f, hv = 0.0f;
    int mat_idx;
    if (best_medium) {
        hp = pa.ro + pa.rd * closest;      // ray.pointing_at(t) — medium.rs:47
        hn = f3(1.0f, 0.0f, 0.0f);         // medium.rs:48
        mat_idx = I.medium_material;
    } else {
        F3 lo = pa.ro, ld = pa.rd;
        if (I.xform_count > 0) xform_ray(sc.xforms, I.xform_first, I.xform_count, lo, ld);
        const int idx = best_pf >> 3, face = best_pf & 7;
        const rtmi_prim_meta M = sc.meta[idx];
        const float4 A = sc.prim_a[idx];
        mat_idx = M.material;
        const bool needs_uv = (sc.mats[mat_idx].flags & RTMI_MATFLAG_NEEDS_UV) != 0u;
        hp = lo + ld * closest; // ray.pointing_at(t)
        if (M.type == RTMI_PRIM_SPHERE || M.type == RTMI_PRIM_MSPHERE) {
            F3 c = f3(A.x, A.y, A.z);
            if (M.type == RTMI_PRIM_MSPHERE) c = moving_center(A, sc.prim_b[idx], M.inv_dt, pa.rtime);
            hn = vdiv(hp - c, A.w); // sphere.rs:50 — outward, never face-forwarded
            if (needs_uv) sphere_uv(hn, hu, hv);
        } else {
            int plane;
            float x0, y0, x1, y1;
            if (M.type == RTMI_PRIM_RECT) {
                plane = (int)((M.flags >> RTMI_PRIMFLAG_PLANE_SHIFT) & 3u);
                x0 = A.x; y0 = A.y; x1 = A.z; y1 = A.w;
            } else { // cube face -> its rect (cube.rs:21-74)
                const float4 B = sc.prim_b[idx];
                const float ax = A.x, ay = A.y, az = A.z, bx = A.w, by = B.x, bz = B.y;
                if (face < 2) { plane = 2; x0 = ax; y0 = ay; x1 = bx; y1 = by; }
                else if (face < 4) { plane = 1; x0 = az; y0 = ax; x1 = bz; y1 = bx; }
                else { plane = 0; x0 = ay; y0 = az; x1 = by; y1 = bz; }
            }
            hn = f3(plane == 0 ? 1.0f : 0.0f, plane == 1 ? 1.0f : 0.0f, plane == 2 ? 1.0f : 0.0f); // rect.rs:58-59
            if (needs_uv) { // rect.rs:52-56
                const float x = plane == 0 ? lo.y + closest * ld.y : (plane == 1 ? lo.z + closest * ld.z : lo.x + closest * ld.x);
                const float y = plane == 0 ? lo.z + closest * ld.z : (plane == 1 ? lo.x + closest * ld.x : lo.y + closest * ld.y);
                hu = (x - x0) / (x1 - x0);
                hv = (y - y0) / (y1 - y0);
            }
        }
        if (I.xform_count > 0) xform_hit(sc.xforms, I.xform_first, I.xform_count, hp, hn);
        if (((M.flags ^ I.flags) & 1u) != 0u) hn = -hn; // FlipNormals — hittable.rs:78-83
    }

    // Material::emitted / Material::scatter (material.rs).  The rejection sampler and the texture
    // lookup are needed by several materials; they are evaluated ONCE here for all lanes that need
    // them (a per-material copy would run the same long code serially for each lane subset).  The
    // draw order per lane is unchanged: Lambertian/Isotropic/fuzzy Metal draw only inside the sampler,
    // Dielectric draws its single uniform, DiffuseLight draws nothing.
    const rtmi_material M = sc.mats[mat_idx];
    const int kind = M.kind;
    const bool can_scatter = pa.depth < max_depth; // color.rs:9
    const bool textured = kind == RTMI_MAT_LAMBERTIAN || kind == RTMI_MAT_METAL || kind == RTMI_MAT_ISOTROPIC;
    const bool want_sample = can_scatter && (kind == RTMI_MAT_LAMBERTIAN || kind == RTMI_MAT_ISOTROPIC ||
                                             (kind == RTMI_MAT_METAL && M.param > 0.0f));
    F3 rs = f3(0, 0, 0);
    if (want_sample) rs = random_in_unit_sphere(g, k0, k1);
    F3 tv = f3(1, 1, 1);
    if (kind == RTMI_MAT_DIFFUSE_LIGHT || (can_scatter && textured)) tv = tex_value(sc, M.tex, hu, hv, hp);
    if (kind == RTMI_MAT_DIFFUSE_LIGHT) pa.L = pa.L + pa.T * tv; // material.rs:148-150
    bool scattered = false;
    const F3 rd = pa.rd;
    F3 nd = rd, att = f3(1, 1, 1);
    if (can_scatter) {
        if (kind == RTMI_MAT_LAMBERTIAN) { // material.rs:49-53 (contract: dir = normal + rand)
            nd = hn + rs;
            att = tv;
            scattered = true;
        } else if (kind == RTMI_MAT_METAL) { // material.rs:75-87
            F3 refl = reflect(normalize(rd), hn);
            if (M.param > 0.0f) refl = refl + rs * M.param;
            if (dot(refl, hn) > 0.0f) {
                nd = refl;
                att = tv;
                scattered = true;
            }
        } else if (kind == RTMI_MAT_DIELECTRIC) { // material.rs:106-126
            F3 outward;
            float ni_over_nt, cosine;
            const float ddn = dot(rd, hn);
            if (ddn > 0.0f) {
                cosine = M.param * ddn / norm(rd);
                outward = -hn;
                ni_over_nt = M.param;
            } else {
                cosine = -ddn / norm(rd);
                outward = hn;
                ni_over_nt = 1.0f / M.param;
            }
            F3 refr;
            bool took_refraction = false;
            if (refract(rd, outward, ni_over_nt, refr)) {
                const float reflect_prob = schlick(cosine, M.param);
                if (rng_uniform(g, k0, k1) >= reflect_prob) { nd = refr; took_refraction = true; }
            }
            if (!took_refraction) nd = reflect(rd, hn);
            scattered = true;
        } else if (kind == RTMI_MAT_ISOTROPIC) { // material.rs:165-168
            nd = rs;
            att = tv;
            scattered = true;
        }
    }
    if (scattered) {
        pa.T = pa.T * att;
        pa.ro = hp;
        pa.rd = nd;
        pa.depth++;
    }
    return scattered;
}

// work item of a wavefront: (sample chunk, local tile) -> pixel of this lane
struct LaneJob {
    uint32_t item, ltile, px, j, pixel, s_begin, s_end;
    bool in_image, wave_has_work;
};
__device__ __forceinline__ LaneJob lane_job(const DevParams &P, int wave, int lane) {
    LaneJob J;
    J.item = blockIdx.x * WAVES_PER_BLOCK + wave; // (chunk, local tile)
    const uint32_t nitems = P.ntiles_local * P.nchunks;
    J.wave_has_work = J.item < nitems;
    const uint32_t chunk = J.item / P.ntiles_local;
    J.ltile = J.item - chunk * P.ntiles_local;
    const uint32_t tile = J.ltile * P.tile_world + P.tile_rank;
    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    J.px = tx * RTMI_TILE + (lane & 7);
    const uint32_t row = ty * RTMI_TILE + (lane >> 3);
    J.in_image = J.px < P.nx && row < P.ny;
    J.j = P.ny - 1u - row;          // `for j in (0..ny).rev()` — tests/test.rs:62
    J.pixel = J.j * P.nx + J.px;    // stream id of this pixel
    J.s_begin = (uint32_t)(((uint64_t)P.ns * chunk) / P.nchunks);
    J.s_end = (uint32_t)(((uint64_t)P.ns * (chunk + 1)) / P.nchunks);
    return J;
}

// ----------------------------------------------------------------------------------
// render kernel, two-phase form (RTMI_FLAG_SYNC): the wavefront alternates between
//   phase A  every lane that holds no unshaded hit traces: (next camera sample if its path ended)
//            + world.hit(); a lane whose ray misses immediately starts its next sample and traces
//            again, while lanes that already found a hit wait.  The phase ends when at least
//            `P.shade_threshold` lanes hold a hit (or no lane can produce one any more).
//   phase B  all lanes holding a hit build the hit record and run the material.
// Shading (Perlin turbulence, rejection samplers, Philox refills, ...) is expensive and very
// divergent; batching it until most lanes need it runs it at high lane utilisation, at the
// price of a few partially filled tracing rounds.  Per-lane program order is unchanged, so
// results do not depend on the threshold.
// ----------------------------------------------------------------------------------
template <bool FAST, bool SIG, bool PROF>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void rtmi_render_kernel(DevScene sc, DevCamera cam, DevParams P,
                                                                           double *__restrict__ partial) {
    __shared__ unsigned long long prof_lds[PROF ? 2 * RTMI_PROF_SLOTS : 1];
    unsigned long long *prof = prof_lds;
    if (PROF) {
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS) prof_lds[threadIdx.x] = 0ull;
        __syncthreads();
    }
    // per wave: [0] node refs, [1] entry distances (FAST only); entry-major so lanes never bank-conflict
    __shared__ uint32_t lds_stack[WAVES_PER_BLOCK][FAST ? 2 : 1][RTMI_MAX_BVH_DEPTH][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t *stack = &lds_stack[wave][0][0][lane];
    unsigned long long sig = 0ull;
    const LaneJob J = lane_job(P, wave, lane);
    if (!PROF && !J.wave_has_work) return;
    const uint32_t k0 = P.key0, k1 = P.key1;
    const int threshold = (int)P.shade_threshold;

    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0; // `col += color(..)` — tests/test.rs:69 (f64 like the reference)
    uint32_t s = (J.in_image && J.wave_has_work) ? J.s_begin : J.s_end;
    bool alive = false, done = s >= J.s_end, have_hit = false;
    Rng g;
    rng_init(g, 0, 0);
    Path pa;
    pa.ro = f3(0, 0, 0); pa.rd = f3(0, 0, 1); pa.rtime = 0.0f; pa.T = f3(1, 1, 1); pa.L = f3(0, 0, 0); pa.depth = 0;
    float closest = RTMI_FLT_MAX;
    int best_item = -1, best_pf = 0;
    bool best_medium = false;

    for (;;) {
        // ================= phase A: trace until enough lanes hold a hit =================
        for (;;) {
            const bool need = !have_hit && !done;
            if (__ballot(need) == 0ull) break;
            prof_tick<PROF>(prof, 0, need);
            if (need) {
                if (!alive) {
                    camera_sample(cam, P, g, k0, k1, s, J.pixel, J.px, J.j, pa);
                    alive = true;
                }
                // ---- world.hit(ray, 0.001, f64::MAX): scan of the top-level list (hittable.rs:37-47)
                RayF W;
                W.o = pa.ro; W.d = pa.rd;
                ray_derive(W);
                closest = RTMI_FLT_MAX;
                best_item = -1; best_pf = 0; best_medium = false;
                for (uint32_t it = 0; it < sc.n_items; it++) {
                    const rtmi_item I = sc.items[it];
                    RayF R = W;
                    if (I.xform_count > 0) {
                        if (xform_ray(sc.xforms, I.xform_first, I.xform_count, R.o, R.d)) ray_derive(R);
                    }
                    const int slot = 1 + (it < 11u ? (int)it : 11);
                    if (!(I.flags & RTMI_ITEMFLAG_MEDIUM)) {
                        float t;
                        int pf;
                        if (geom_query<FAST, PROF>(sc, I, R, pa.rtime, P.t_min, closest, stack, t, pf, prof, slot)) {
                            closest = t; best_item = (int)it; best_pf = pf; best_medium = false;
                        }
                    } else {
                        // ConstantMedium::hit — medium.rs:28-56
                        float t1, t2, tm;
                        int pf;
                        if (geom_query<FAST, PROF>(sc, I, R, pa.rtime, -RTMI_FLT_MAX, RTMI_FLT_MAX, stack, t1, pf, prof, slot)) {
                            if (geom_query<FAST, PROF>(sc, I, R, pa.rtime, t1 + 0.0001f, RTMI_FLT_MAX, stack, t2, pf, prof, slot)) {
                                if (medium_sample(t1, t2, P.t_min, closest, W.d, I.neg_inv_density, g, k0, k1, tm)) {
                                    closest = tm; best_item = (int)it; best_medium = true;
                                }
                            }
                        }
                    }
                }
                if (best_item >= 0) {
                    have_hit = true;
                } else { // miss: black background (color.rs:21); the path ends, next sample
                    acc0 += (double)pa.L.x; acc1 += (double)pa.L.y; acc2 += (double)pa.L.z;
                    s++; alive = false;
                    done = s >= J.s_end;
                }
            }
            if (__popcll(__ballot(have_hit)) >= threshold) break;
        }
        // ================= phase B: shade every lane that holds a hit =================
        if (__ballot(have_hit) == 0ull) break; // nobody holds a hit and nobody can trace: all done
        prof_tick<PROF>(prof, 16, have_hit);
        if (have_hit) {
            have_hit = false;
            if (SIG) sig += (unsigned long long)sig_mix(__float_as_uint(closest), pa.depth);
            if (!shade_hit(sc, P.max_depth, g, k0, k1, closest, best_item, best_pf, best_medium, pa)) {
                // absorbed, emitter or depth limit: the path ends
                acc0 += (double)pa.L.x; acc1 += (double)pa.L.y; acc2 += (double)pa.L.z;
                s++; alive = false;
                done = s >= J.s_end;
            }
        }
    }

    if (PROF) {
        __syncthreads();
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS && prof_lds[threadIdx.x] != 0ull) atomicAdd(P.prof + threadIdx.x, prof_lds[threadIdx.x]);
        if (!J.wave_has_work) return;
    }
    // partial[chunk][ltile][channel][lane]
    double *out = partial + ((size_t)J.item * 3) * 64 + lane;
    out[0] = acc0; out[64] = acc1; out[128] = acc2;
    if (SIG && J.in_image) atomicAdd(P.path_sig + (size_t)J.ltile * 64 + lane, sig); // integer add: order-independent
}

// ----------------------------------------------------------------------------------
// render kernel, two-phase form with wave-cooperative BVH traversal (default).
// Same schedule as rtmi_render_kernel; the item scan of phase A is executed by ALL lanes (lanes
// without a pending query are workers for the others' BVH traversals).
// ----------------------------------------------------------------------------------
template <bool SIG, bool PROF, int WPS>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, WPS) void rtmi_render_coop(DevScene sc, DevCamera cam, DevParams P,
                                                                              double *__restrict__ partial) {
    __shared__ unsigned long long prof_lds[PROF ? 2 * RTMI_PROF_SLOTS : 1];
    unsigned long long *prof = prof_lds;
    if (PROF) {
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS) prof_lds[threadIdx.x] = 0ull;
        __syncthreads();
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[]; // per wave: pool | ctx | best
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int cap = (int)P.coop_cap;
    uint32_t *wlds = lds_dyn + (size_t)wave * (2u * cap + 64u * 12u + 128u);
    unsigned long long sig = 0ull;
    const LaneJob J = lane_job(P, wave, lane);
    if (!PROF && !J.wave_has_work) return;
    const uint32_t k0 = P.key0, k1 = P.key1;
    const int threshold = (int)P.shade_threshold;

    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    uint32_t s = (J.in_image && J.wave_has_work) ? J.s_begin : J.s_end;
    bool alive = false, done = s >= J.s_end, have_hit = false, overflow = false;
    Rng g;
    rng_init(g, 0, 0);
    Path pa;
    pa.ro = f3(0, 0, 0); pa.rd = f3(0, 0, 1); pa.rtime = 0.0f; pa.T = f3(1, 1, 1); pa.L = f3(0, 0, 0); pa.depth = 0;
    float closest = RTMI_FLT_MAX;
    int best_item = -1, best_pf = 0;
    bool best_medium = false;
    unsigned long long tstamp = PROF ? __builtin_readcyclecounter() : 0ull;

    for (;;) {
        // ================= phase A =================
        for (;;) {
            const bool need = !have_hit && !done;
            if (__ballot(need) == 0ull) break;
            prof_tick<PROF>(prof, 0, need);
            prof_time<PROF>(prof, 31, tstamp); // loop overhead / phase switching
            if (need && !alive) {
                camera_sample(cam, P, g, k0, k1, s, J.pixel, J.px, J.j, pa);
                alive = true;
            }
            prof_time<PROF>(prof, 25, tstamp); // camera samples
            RayF W;
            W.o = pa.ro; W.d = pa.rd;
            ray_derive(W);
            if (need) { closest = RTMI_FLT_MAX; best_item = -1; best_pf = 0; best_medium = false; }
            for (uint32_t it = 0; it < sc.n_items; it++) { // executed by all 64 lanes
                const rtmi_item I = sc.items[it];
                RayF R = W;
                if (I.xform_count > 0) {
                    if (xform_ray(sc.xforms, I.xform_first, I.xform_count, R.o, R.d)) ray_derive(R);
                }
                const int slot = 1 + (it < 11u ? (int)it : 11);
                if (!(I.flags & RTMI_ITEMFLAG_MEDIUM)) {
                    float t;
                    int pf;
                    if (geom_query_coop<PROF>(sc, I, need, R, pa.rtime, P.t_min, closest, wlds, cap, t, pf, overflow, prof, slot)) {
                        closest = t; best_item = (int)it; best_pf = pf; best_medium = false;
                    }
                    prof_time<PROF>(prof, I.kind == RTMI_ITEM_BVH ? (it == 0 ? 27 : 28) : 26, tstamp);
                } else {
                    // ConstantMedium::hit — medium.rs:28-56
                    float t1 = 0.0f, t2 = 0.0f, tm;
                    int pf;
                    const bool h1 = geom_query_coop<PROF>(sc, I, need, R, pa.rtime, -RTMI_FLT_MAX, RTMI_FLT_MAX, wlds, cap, t1, pf, overflow, prof, slot);
                    const bool h2 = geom_query_coop<PROF>(sc, I, need && h1, R, pa.rtime, t1 + 0.0001f, RTMI_FLT_MAX, wlds, cap, t2, pf, overflow, prof, slot);
                    if (need && h1 && h2) {
                        if (medium_sample(t1, t2, P.t_min, closest, W.d, I.neg_inv_density, g, k0, k1, tm)) {
                            closest = tm; best_item = (int)it; best_medium = true;
                        }
                    }
                    prof_time<PROF>(prof, 29, tstamp); // media
                }
            }
            if (need) {
                if (best_item >= 0) {
                    have_hit = true;
                } else { // miss: black background (color.rs:21)
                    acc0 += (double)pa.L.x; acc1 += (double)pa.L.y; acc2 += (double)pa.L.z;
                    s++; alive = false;
                    done = s >= J.s_end;
                }
            }
            if (__popcll(__ballot(have_hit)) >= threshold) break;
        }
        // ================= phase B =================
        if (__ballot(have_hit) == 0ull) break;
        prof_tick<PROF>(prof, 16, have_hit);
        if (have_hit) {
            have_hit = false;
            if (SIG) sig += (unsigned long long)sig_mix(__float_as_uint(closest), pa.depth);
            if (!shade_hit(sc, P.max_depth, g, k0, k1, closest, best_item, best_pf, best_medium, pa)) {
                acc0 += (double)pa.L.x; acc1 += (double)pa.L.y; acc2 += (double)pa.L.z;
                s++; alive = false;
                done = s >= J.s_end;
            }
        }
        prof_time<PROF>(prof, 30, tstamp); // shading
    }
    if (__ballot(overflow) != 0ull && lane == 0) atomicAdd(P.status, 1u); // reported loudly by the host

    if (PROF) {
        __syncthreads();
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS && prof_lds[threadIdx.x] != 0ull) atomicAdd(P.prof + threadIdx.x, prof_lds[threadIdx.x]);
        if (!J.wave_has_work) return;
    }
    double *out = partial + ((size_t)J.item * 3) * 64 + lane;
    out[0] = acc0; out[64] = acc1; out[128] = acc2;
    if (SIG && J.in_image) atomicAdd(P.path_sig + (size_t)J.ltile * 64 + lane, sig);
}

// ----------------------------------------------------------------------------------
// render kernel, asynchronous form (default).
//
// Measured on the synchronous kernel: inside BVH traversal only 4-10 % of the lanes are active
// per iteration (a few lanes walk long while the rest have already left the tree), because the
// whole wavefront waits at every item and at every bounce.  Here every lane is its own state
// machine over the SAME per-lane program order (items in list order, media draws in order, so
// results are bit-identical): a lane that finished its hit query goes on to shade, to its next
// bounce and to its next sample while others still traverse.  Each loop iteration the wavefront
// VOTES (ballots) for the state most lanes are in and executes only that body, which lets lanes
// that drifted apart re-converge; lanes in other states wait one round.
//   ST_ITEM : commit the finished item into `closest` (incl. ConstantMedium logic), then enter
//             following items; single-primitive items are tested right here
//   ST_NODE : one BVH node step        ST_PRIM : one primitive (BVH leaf or nested-list member)
//   ST_SHADE: hit record + material    ST_NEW  : next camera sample      ST_DONE
// ----------------------------------------------------------------------------------
enum { ST_ITEM = 0, ST_NODE = 1, ST_PRIM = 2, ST_SHADE = 3, ST_NEW = 4, ST_DONE = 5 };

template <bool FAST, bool SIG, bool PROF>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void rtmi_render_async(DevScene sc, DevCamera cam, DevParams P,
                                                                          double *__restrict__ partial) {
    __shared__ unsigned long long prof_lds[PROF ? 2 * RTMI_PROF_SLOTS : 1];
    unsigned long long *prof = prof_lds;
    if (PROF) {
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS) prof_lds[threadIdx.x] = 0ull;
        __syncthreads();
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[]; // [wave][2][stack_depth][64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t SD = P.stack_depth;
    uint32_t *stack = lds_dyn + (size_t)wave * 2u * SD * 64u + lane;
    float *stack_t = reinterpret_cast<float *>(stack + SD * 64u);
    unsigned long long sig = 0ull;
    const LaneJob J = lane_job(P, wave, lane);
    if (!PROF && !J.wave_has_work) return;
    const uint32_t k0 = P.key0, k1 = P.key1;
    const int n_items = (int)sc.n_items;

    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    uint32_t s = (J.in_image && J.wave_has_work) ? J.s_begin : J.s_end;
    Rng g;
    rng_init(g, 0, 0);
    Path pa;
    pa.ro = f3(0, 0, 0); pa.rd = f3(0, 0, 1); pa.rtime = 0.0f; pa.T = f3(1, 1, 1); pa.L = f3(0, 0, 0); pa.depth = 0;

    // hit-query state
    RayF W;             // world-frame ray of the current query
    W.o = pa.ro; W.d = pa.rd; W.inv_d = f3(0, 0, 0); W.a = 1.0f; W.inv_a = 1.0f;
    RayF R = W;         // ray in the frame of the current item
    int it = 0, ph = 0; // item index, ConstantMedium phase (0: first boundary query, 1: second)
    bool pending = false;           // item `it` has finished with (have, bt, bpf) and must be committed
    float closest = RTMI_FLT_MAX, t1 = 0.0f;
    int best_item = -1, best_pf = 0;
    bool best_medium = false;
    uint32_t iflags = 0u;
    float q_min = 0.0f, q_max = 0.0f;
    // traversal / list state of the current item
    int cur = 0, sp = 0, pend = 0;  // node-or-leaf ref | list cursor, stack pointer, list end
    bool have = false, is_list = false;
    float bt = 0.0f, limit = RTMI_FLT_MAX, m_abs = 0.0f;
    int bpf = 0;
    int st = ST_NEW;

    for (;;) {
        // ---- vote
        const unsigned long long mI = __ballot(st == ST_ITEM), mN = __ballot(st == ST_NODE), mP = __ballot(st == ST_PRIM),
                                 mS = __ballot(st == ST_SHADE), mC = __ballot(st == ST_NEW);
        const int nI = __popcll(mI), nN = __popcll(mN), nP = __popcll(mP), nS = __popcll(mS), nC = __popcll(mC);
        if ((nI | nN | nP | nS | nC) == 0) break; // every lane is ST_DONE
        int run = ST_ITEM, best_n = nI;
        if (nN > best_n) { run = ST_NODE; best_n = nN; }
        if (nP > best_n) { run = ST_PRIM; best_n = nP; }
        if (nS > best_n) { run = ST_SHADE; best_n = nS; }
        if (nC > best_n) { run = ST_NEW; best_n = nC; }
        prof_tick<PROF>(prof, 20 + run, st == run);

        if (run == ST_NODE) {
            if (st == ST_NODE) {
                const float4 *n = sc.nodes + (size_t)cur * 4;
                const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
                const int left = __float_as_int(n3.x), right = __float_as_int(n3.y);
                int next = 0;
                bool got = false;
                if (!FAST) {
                    bool vl = left < 0 || aabb_hit(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, R, q_min, q_max);
                    bool vr = right < 0 || aabb_hit(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, R, q_min, q_max);
                    if (right == left) vr = false;
                    if (vl) {
                        if (vr) { stack[sp * 64] = (uint32_t)right; sp++; }
                        next = left; got = true;
                    } else if (vr) { next = right; got = true; }
                    if (!got && sp > 0) { sp--; next = (int)stack[sp * 64]; got = true; }
                } else {
                    float tl, tr;
                    bool vl = aabb_hit_t(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, R, q_min, q_max, tl);
                    bool vr = aabb_hit_t(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, R, q_min, q_max, tr);
                    vl = vl && !(tl > limit);
                    vr = vr && !(tr > limit) && right != left;
                    if (vl && vr) {
                        const bool lfirst = !(tr < tl);
                        stack[sp * 64] = (uint32_t)(lfirst ? right : left);
                        stack_t[sp * 64] = lfirst ? tr : tl;
                        sp++;
                        next = lfirst ? left : right; got = true;
                    } else if (vl) { next = left; got = true; }
                    else if (vr) { next = right; got = true; }
                    while (!got && sp > 0) {
                        sp--;
                        if (!(stack_t[sp * 64] > limit)) { next = (int)stack[sp * 64]; got = true; }
                    }
                }
                if (got) { cur = next; st = next >= 0 ? ST_NODE : ST_PRIM; }
                else { pending = true; st = ST_ITEM; }
            }
        } else if (run == ST_PRIM) {
            if (st == ST_PRIM) {
                float t;
                int pf;
                if (is_list) { // HittableList::hit — hittable.rs:37-47 (nested list of primitives)
                    const int type = sc.meta[cur].type;
                    if (prim_test(sc, type, cur, R, pa.rtime, q_min, bt, t, pf)) { bt = t; bpf = pf; have = true; }
                    cur++;
                    if (cur >= pend) { pending = true; st = ST_ITEM; }
                } else {       // BVH leaf
                    const int type = (int)(((uint32_t)cur >> 28) & 7u);
                    const int idx = (int)((uint32_t)cur & 0x0fffffffu);
                    if (prim_test(sc, type, idx, R, pa.rtime, q_min, q_max, t, pf)) {
                        if (!FAST) {
                            if (!have || !(bt < t)) { bt = t; bpf = pf; have = true; }
                        } else if (!have || t < bt || (t == bt && pf > bpf)) {
                            bt = t; bpf = pf; have = true;
                            limit = bt + (__builtin_fabsf(bt) * (1.0f / 128.0f) + m_abs);
                        }
                    }
                    bool got = false;
                    int next = 0;
                    if (!FAST) {
                        if (sp > 0) { sp--; next = (int)stack[sp * 64]; got = true; }
                    } else {
                        while (!got && sp > 0) {
                            sp--;
                            if (!(stack_t[sp * 64] > limit)) { next = (int)stack[sp * 64]; got = true; }
                        }
                    }
                    if (got) { cur = next; st = next >= 0 ? ST_NODE : ST_PRIM; }
                    else { pending = true; st = ST_ITEM; }
                }
            }
        } else if (run == ST_ITEM) {
            if (st == ST_ITEM) {
                for (;;) {
                    // ---- commit the finished item (hittable.rs:40-45; medium.rs:30-53)
                    if (pending) {
                        pending = false;
                        if (!(iflags & RTMI_ITEMFLAG_MEDIUM)) {
                            if (have) { closest = bt; best_item = it; best_pf = bpf; best_medium = false; }
                            it++;
                        } else if (ph == 0) {
                            if (have) { t1 = bt; ph = 1; } else { it++; }
                        } else {
                            if (have) {
                                float tm;
                                if (medium_sample(t1, bt, P.t_min, closest, W.d, sc.items[it].neg_inv_density, g, k0, k1, tm)) {
                                    closest = tm; best_item = it; best_medium = true;
                                }
                            }
                            ph = 0;
                            it++;
                        }
                    }
                    // ---- end of the list: world.hit() is complete (color.rs:7)
                    if (it >= n_items) {
                        if (best_item >= 0) { st = ST_SHADE; }
                        else { // miss: black background (color.rs:21)
                            acc0 += (double)pa.L.x; acc1 += (double)pa.L.y; acc2 += (double)pa.L.z;
                            s++; st = ST_NEW;
                        }
                        break;
                    }
                    // ---- enter item `it`
                    const rtmi_item I = sc.items[it];
                    iflags = I.flags;
                    R = W;
                    if (I.xform_count > 0) {
                        if (xform_ray(sc.xforms, I.xform_first, I.xform_count, R.o, R.d)) ray_derive(R);
                    }
                    if (iflags & RTMI_ITEMFLAG_MEDIUM) {
                        q_min = ph == 0 ? -RTMI_FLT_MAX : t1 + 0.0001f;
                        q_max = RTMI_FLT_MAX;
                    } else {
                        q_min = P.t_min;
                        q_max = closest;
                    }
                    have = false;
                    if (I.kind == RTMI_ITEM_BVH) {
                        // BVHNode::hit of the root: its own bbox first (bvh.rs:71)
                        if (!aabb_hit(I.root_min[0], I.root_min[1], I.root_min[2], I.root_max[0], I.root_max[1],
                                      I.root_max[2], R, q_min, q_max)) {
                            pending = true;
                            continue;
                        }
                        cur = I.first; sp = 0; is_list = false;
                        bt = FAST ? RTMI_FLT_MAX : 0.0f; bpf = 0; limit = RTMI_FLT_MAX;
                        m_abs = FAST ? I.scale * (1.0f / 8192.0f) * __builtin_sqrtf(R.inv_a) : 0.0f;
                        st = ST_NODE;
                        break;
                    }
                    if (I.count == 1) { // a single primitive: test it here
                        float t;
                        int pf;
                        const int type = sc.meta[I.first].type;
                        if (prim_test(sc, type, I.first, R, pa.rtime, q_min, q_max, t, pf)) { bt = t; bpf = pf; have = true; }
                        pending = true;
                        continue;
                    }
                    if (I.count <= 0) { pending = true; continue; }
                    cur = I.first; pend = I.first + I.count; is_list = true; bt = q_max; bpf = 0;
                    st = ST_PRIM;
                    break;
                }
            }
        } else if (run == ST_SHADE) {
            if (st == ST_SHADE) {
                if (SIG) sig += (unsigned long long)sig_mix(__float_as_uint(closest), pa.depth);
                if (shade_hit(sc, P.max_depth, g, k0, k1, closest, best_item, best_pf, best_medium, pa)) {
                    W.o = pa.ro; W.d = pa.rd;
                    ray_derive(W);
                    it = 0; ph = 0; pending = false; closest = RTMI_FLT_MAX; best_item = -1; best_medium = false;
                    st = ST_ITEM;
                } else {
                    acc0 += (double)pa.L.x; acc1 += (double)pa.L.y; acc2 += (double)pa.L.z;
                    s++; st = ST_NEW;
                }
            }
        } else { // ST_NEW
            if (st == ST_NEW) {
                if (s >= J.s_end) { st = ST_DONE; }
                else {
                    camera_sample(cam, P, g, k0, k1, s, J.pixel, J.px, J.j, pa);
                    W.o = pa.ro; W.d = pa.rd;
                    ray_derive(W);
                    it = 0; ph = 0; pending = false; closest = RTMI_FLT_MAX; best_item = -1; best_medium = false;
                    st = ST_ITEM;
                }
            }
        }
    }

    if (PROF) {
        __syncthreads();
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS && prof_lds[threadIdx.x] != 0ull) atomicAdd(P.prof + threadIdx.x, prof_lds[threadIdx.x]);
        if (!J.wave_has_work) return;
    }
    double *out = partial + ((size_t)J.item * 3) * 64 + lane;
    out[0] = acc0; out[64] = acc1; out[128] = acc2;
    if (SIG && J.in_image) atomicAdd(P.path_sig + (size_t)J.ltile * 64 + lane, sig);
}

// `col /= ns; sqrt; clamp; (255.99*c) as i32` — tests/test.rs:71-78, per local texel.
// Chunk partial sums are added in chunk order (deterministic).
__global__ void rtmi_resolve_kernel(const double *__restrict__ partial, rtmi_texel *__restrict__ out, DevParams P) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P.ntiles_local * 64u) return;
    const uint32_t ltile = tid >> 6, lane = tid & 63u;
    double sum[3] = {0.0, 0.0, 0.0};
    for (uint32_t c = 0; c < P.nchunks; c++) {
        const double *src = partial + ((size_t)(c * P.ntiles_local + ltile) * 3) * 64 + lane;
        sum[0] += src[0]; sum[1] += src[64]; sum[2] += src[128];
    }
    rtmi_texel tx;
    uint32_t q[3];
    float lin[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const double m = sum[ch] / (double)P.ns;
        lin[ch] = (float)m;
        double g = sqrt(m);
        g = (g > 0.0) ? ((g < 1.0) ? g : 1.0) : 0.0; // nalgebra::clamp(val, 0, 1); NaN -> 0
        const double x = 255.99 * g;
        q[ch] = (x != x) ? 0u : (uint32_t)(int32_t)x; // `as i32`; in [0,255] after the clamp
    }
    tx.r = lin[0]; tx.g = lin[1]; tx.b = lin[2];
    tx.rgb8 = q[0] | (q[1] << 8) | (q[2] << 16);
    out[tid] = tx;
}

// ---- device evaluation of the arithmetic contract, for parity tests --------------------
// op: 0 sin, 1 log, 2 atan2(x,y), 3 asin, 4 x/y, 5 sqrt, 6 u01(bits of x)
__global__ void rtmi_math_probe_kernel(int op, const float *x, const float *y, float *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
    case 0: r = rtmi_sinf(x[i]); break;
    case 1: r = rtmi_logf(x[i]); break;
    case 2: r = rtmi_atan2f(x[i], y[i]); break;
    case 3: r = rtmi_asinf(x[i]); break;
    case 4: r = x[i] / y[i]; break;
    case 5: r = __builtin_sqrtf(x[i]); break;
    default: r = rtmi_u01(__float_as_uint(x[i])); break;
    }
    out[i] = r;
}
__global__ void rtmi_philox_probe_kernel(const uint32_t *ctr, const uint32_t *key, uint32_t *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o0, o1, o2, o3;
    philox(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[2 * i], key[2 * i + 1], o0, o1, o2, o3);
    out[4 * i] = o0; out[4 * i + 1] = o1; out[4 * i + 2] = o2; out[4 * i + 3] = o3;
}

// ======================================================================================
// host side of the C ABI
// ======================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(RTMI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

struct rtmi_scene {
    int device = 0;
    DevScene dev{};
    std::vector<void *> allocs;
    rtmi_scene_desc meta{}; // counts only (pointers nulled)
    double *partial = nullptr;
    size_t partial_bytes = 0;
    unsigned int *status = nullptr; // device word: cooperative-traversal pool overflows (must stay 0)
    rtmi_texel *texels = nullptr; // scratch for the blocking host API
    size_t texel_count = 0;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

extern "C" const char *rtmi_last_error(void) { return g_err.c_str(); }

extern "C" int rtmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <typename T>
static int upload(rtmi_scene *s, const T *src, size_t n, const T **dst) {
    *dst = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (n) HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = reinterpret_cast<const T *>(p);
    return RTMI_OK;
}

static int validate(const rtmi_scene_desc *d) {
    if (!d) return fail(RTMI_ERR_INVALID, "desc is NULL");
    if (d->abi_version != RTMI_ABI_VERSION) return fail(RTMI_ERR_INVALID, "abi_version mismatch");
    if (d->n_items == 0 || !d->items) return fail(RTMI_ERR_INVALID, "scene has no items");
    if (d->max_bvh_depth > RTMI_MAX_BVH_DEPTH)
        return fail(RTMI_ERR_UNSUPPORTED, "BVH deeper than RTMI_MAX_BVH_DEPTH");
    if (d->n_prims >= (1u << 28)) return fail(RTMI_ERR_UNSUPPORTED, "too many primitives");
    auto prim_ok = [&](int64_t i) { return i >= 0 && (uint64_t)i < d->n_prims; };
    for (uint32_t i = 0; i < d->n_prims; i++) {
        const rtmi_prim_meta &m = d->prim_meta[i];
        if (m.type < 0 || m.type > RTMI_PRIM_CUBE) return fail(RTMI_ERR_INVALID, "bad primitive type");
        if (m.material < 0 || (uint32_t)m.material >= d->n_materials)
            return fail(RTMI_ERR_INVALID, "primitive material out of range");
    }
    for (uint32_t i = 0; i < d->n_nodes; i++) {
        const int32_t ch[2] = {d->nodes[i].left, d->nodes[i].right};
        for (int c = 0; c < 2; c++) {
            if (ch[c] >= 0) {
                if ((uint32_t)ch[c] >= d->n_nodes) return fail(RTMI_ERR_INVALID, "BVH child out of range");
            } else {
                uint32_t type = ((uint32_t)ch[c] >> 28) & 7u, idx = (uint32_t)ch[c] & 0x0fffffffu;
                if (type > RTMI_PRIM_CUBE || !prim_ok(idx)) return fail(RTMI_ERR_INVALID, "BVH leaf out of range");
                if ((int)type != d->prim_meta[idx].type) return fail(RTMI_ERR_INVALID, "BVH leaf type mismatch");
            }
        }
    }
    for (uint32_t i = 0; i < d->n_items; i++) {
        const rtmi_item &it = d->items[i];
        if (it.kind == RTMI_ITEM_LIST) {
            if (it.count < 0) return fail(RTMI_ERR_INVALID, "item primitive count is negative");
            if (it.count > 0 && (!prim_ok(it.first) || !prim_ok((int64_t)it.first + it.count - 1)))
                return fail(RTMI_ERR_INVALID, "item primitive range out of bounds");
        } else if (it.kind == RTMI_ITEM_BVH) {
            if (it.first < 0 || (uint32_t)it.first >= d->n_nodes) return fail(RTMI_ERR_INVALID, "item BVH root out of range");
        } else {
            return fail(RTMI_ERR_INVALID, "bad item kind");
        }
        if (it.xform_count < 0 || it.xform_first < 0 || (uint32_t)(it.xform_first + it.xform_count) > d->n_xforms)
            return fail(RTMI_ERR_INVALID, "item transform range out of bounds");
        if ((it.flags & RTMI_ITEMFLAG_MEDIUM) &&
            (it.medium_material < 0 || (uint32_t)it.medium_material >= d->n_materials))
            return fail(RTMI_ERR_INVALID, "medium material out of range");
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rtmi_material &m = d->materials[i];
        if (m.kind < 0 || m.kind > RTMI_MAT_ISOTROPIC) return fail(RTMI_ERR_INVALID, "bad material kind");
        if (m.kind != RTMI_MAT_DIELECTRIC && (m.tex < 0 || (uint32_t)m.tex >= d->n_textures))
            return fail(RTMI_ERR_INVALID, "material texture out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rtmi_texture &t = d->textures[i];
        switch (t.kind) {
        case RTMI_TEX_SOLID: break;
        case RTMI_TEX_CHECKER:
            if (t.i0 < 0 || t.i1 < 0 || (uint32_t)t.i0 >= d->n_textures || (uint32_t)t.i1 >= d->n_textures)
                return fail(RTMI_ERR_INVALID, "checker child out of range");
            break;
        case RTMI_TEX_NOISE:
            if (t.i0 < 0 || (uint32_t)t.i0 >= d->n_perlin) return fail(RTMI_ERR_INVALID, "perlin table out of range");
            break;
        case RTMI_TEX_IMAGE:
            if (t.i0 < 0 || (uint32_t)t.i0 >= d->n_images) return fail(RTMI_ERR_INVALID, "image out of range");
            break;
        default: return fail(RTMI_ERR_INVALID, "bad texture kind");
        }
    }
    for (uint32_t i = 0; i < d->n_images; i++) {
        const rtmi_image &im = d->images[i];
        if (im.nx == 0 || im.ny == 0 || im.offset + 3ull * im.nx * im.ny > d->image_bytes)
            return fail(RTMI_ERR_INVALID, "image outside image_data");
    }
    return RTMI_OK;
}

extern "C" int rtmi_scene_create(const rtmi_scene_desc *d, int device, rtmi_scene **out) {
    if (!out) return fail(RTMI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate(d);
    if (rc) return rc;
    int ndev = rtmi_device_count();
    if (ndev <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available (the rtmi path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RTMI_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    rtmi_scene *s = new (std::nothrow) rtmi_scene();
    if (!s) return fail(RTMI_ERR_NOMEM, "out of host memory");
    s->device = device;
    s->meta = *d;
    const float4 *nodes4 = nullptr;
    rc = upload(s, d->items, d->n_items, &s->dev.items);
    if (!rc) rc = upload(s, reinterpret_cast<const float4 *>(d->prim_a), d->n_prims, &s->dev.prim_a);
    if (!rc) rc = upload(s, reinterpret_cast<const float4 *>(d->prim_b), d->n_prims, &s->dev.prim_b);
    if (!rc) rc = upload(s, d->prim_meta, d->n_prims, &s->dev.meta);
    if (!rc) rc = upload(s, reinterpret_cast<const float4 *>(d->nodes), (size_t)d->n_nodes * 4, &nodes4);
    if (!rc) rc = upload(s, d->xforms, d->n_xforms, &s->dev.xforms);
    if (!rc) rc = upload(s, d->materials, d->n_materials, &s->dev.mats);
    if (!rc) rc = upload(s, d->textures, d->n_textures, &s->dev.texs);
    if (!rc) rc = upload(s, d->perlin, d->n_perlin, &s->dev.perlin);
    if (!rc) rc = upload(s, d->images, d->n_images, &s->dev.images);
    if (!rc) rc = upload(s, d->image_data, (size_t)d->image_bytes, &s->dev.image_data);
    if (rc) {
        rtmi_scene_destroy(s);
        return rc;
    }
    s->dev.nodes = nodes4;
    s->dev.n_items = d->n_items;
    if (hipMalloc(reinterpret_cast<void **>(&s->status), sizeof(unsigned int)) != hipSuccess ||
        hipMemset(s->status, 0, sizeof(unsigned int)) != hipSuccess) {
        rtmi_scene_destroy(s);
        return fail(RTMI_ERR_DEVICE, "allocating the status word failed");
    }
    for (int i = 0; i < 3; i++)
        if (hipEventCreate(&s->ev[i]) != hipSuccess) {
            rtmi_scene_destroy(s);
            return fail(RTMI_ERR_DEVICE, "hipEventCreate failed");
        }
    *out = s;
    return RTMI_OK;
}

extern "C" void rtmi_scene_destroy(rtmi_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->partial) (void)hipFree(s->partial);
    if (s->texels) (void)hipFree(s->texels);
    if (s->status) (void)hipFree(s->status);
    for (int i = 0; i < 3; i++)
        if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
    delete s;
}

static inline uint32_t tiles_x_of(const rtmi_render_params *p) { return (p->nx + RTMI_TILE - 1) / RTMI_TILE; }
static inline uint32_t tiles_y_of(const rtmi_render_params *p) { return (p->ny + RTMI_TILE - 1) / RTMI_TILE; }
static inline uint32_t local_tiles_of(const rtmi_render_params *p, uint32_t rank) {
    const uint32_t T = tiles_x_of(p) * tiles_y_of(p);
    return rank < T ? (T - rank + p->tile_world - 1) / p->tile_world : 0;
}

extern "C" uint32_t rtmi_local_tiles(const rtmi_render_params *p) {
    if (!p || p->tile_world == 0) return 0;
    return local_tiles_of(p, p->tile_rank);
}

static int check_params(const rtmi_render_params *p) {
    if (!p) return fail(RTMI_ERR_INVALID, "params is NULL");
    if (p->nx == 0 || p->ny == 0 || p->ns == 0) return fail(RTMI_ERR_INVALID, "nx, ny and ns must be positive");
    if ((uint64_t)p->nx * p->ny > 0xffffffffull) return fail(RTMI_ERR_UNSUPPORTED, "image too large");
    if (p->tile_world == 0 || p->tile_rank >= p->tile_world) return fail(RTMI_ERR_INVALID, "bad tile_rank/tile_world");
    return RTMI_OK;
}

extern "C" int rtmi_render_device(rtmi_scene *s, const rtmi_camera *cam, const rtmi_render_params *p, void *d_texels,
                                  void *stream_, rtmi_stats *stats) {
    if (!s || !cam || !d_texels) return fail(RTMI_ERR_INVALID, "NULL argument");
    int rc = check_params(p);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);

    DevParams P{};
    P.nx = p->nx; P.ny = p->ny; P.ns = p->ns; P.max_depth = p->max_depth; P.t_min = p->t_min;
    P.key0 = (uint32_t)p->seed; P.key1 = (uint32_t)(p->seed >> 32);
    P.tile_rank = p->tile_rank; P.tile_world = p->tile_world; P.tiles_x = tiles_x_of(p);
    P.ntiles_local = local_tiles_of(p, p->tile_rank);
    if (P.ntiles_local == 0) {
        if (stats) memset(stats, 0, sizeof(*stats));
        return RTMI_OK;
    }
    // split the sample range so that the grid holds many more wavefronts than the chip
    // has slots (256 CUs x 16): balances tiles of very different path lengths
    uint32_t chunks = p->spp_chunks;
    if (chunks == 0) {
        const uint32_t target_items = 256u * 16u * 8u;
        chunks = (target_items + P.ntiles_local - 1) / P.ntiles_local;
        if (chunks > 64u) chunks = 64u;
    }
    if (chunks > p->ns) chunks = p->ns;
    if (chunks == 0) chunks = 1;
    P.nchunks = chunks;
    P.path_sig = reinterpret_cast<unsigned long long *>(p->path_sig);
    if (p->flags & RTMI_FLAG_PATH_SIG) {
        if (!p->path_sig) return fail(RTMI_ERR_INVALID, "RTMI_FLAG_PATH_SIG needs params.path_sig");
        HIP_TRY(hipMemsetAsync(P.path_sig, 0, (size_t)P.ntiles_local * 64 * sizeof(unsigned long long), stream));
    }

    const size_t need = (size_t)P.ntiles_local * chunks * 64 * 3 * sizeof(double);
    if (need > s->partial_bytes) {
        if (s->partial) { HIP_TRY(hipFree(s->partial)); s->partial = nullptr; s->partial_bytes = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->partial), need));
        s->partial_bytes = need;
    }

    DevCamera C;
    C.origin = F3{cam->origin[0], cam->origin[1], cam->origin[2]};
    C.llc = F3{cam->lower_left_corner[0], cam->lower_left_corner[1], cam->lower_left_corner[2]};
    C.horizontal = F3{cam->horizontal[0], cam->horizontal[1], cam->horizontal[2]};
    C.vertical = F3{cam->vertical[0], cam->vertical[1], cam->vertical[2]};
    C.u = F3{cam->u[0], cam->u[1], cam->u[2]};
    C.v = F3{cam->v[0], cam->v[1], cam->v[2]};
    C.time0 = cam->time0; C.time1 = cam->time1; C.lens_radius = cam->lens_radius;

    const uint32_t nitems = P.ntiles_local * chunks;
    const uint32_t blocks = (nitems + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (stats) HIP_TRY(hipEventRecord(s->ev[0], stream));
    const bool fast = (p->flags & RTMI_FLAG_FAST_CULL) != 0u, sigf = (p->flags & RTMI_FLAG_PATH_SIG) != 0u;
    const dim3 grid(blocks), block(64 * WAVES_PER_BLOCK);
    P.stack_depth = s->meta.max_bvh_depth + 1u;
    P.shade_threshold = p->shade_threshold ? (p->shade_threshold > 64u ? 64u : p->shade_threshold) : 1u;
    const size_t dyn_lds = (size_t)WAVES_PER_BLOCK * 2u * P.stack_depth * 64u * sizeof(uint32_t);
    const bool prof = (p->flags & RTMI_FLAG_PROFILE) != 0u, sync = (p->flags & RTMI_FLAG_SYNC) != 0u;
    if (prof) {
        if (!p->prof) return fail(RTMI_ERR_INVALID, "RTMI_FLAG_PROFILE needs params.prof");
        P.prof = reinterpret_cast<unsigned long long *>(p->prof);
        HIP_TRY(hipMemsetAsync(P.prof, 0, 2 * RTMI_PROF_SLOTS * sizeof(unsigned long long), stream));
    }
    // kernel selection: default = two-phase schedule, cooperative traversal when fast-cull is on
    // (it implements the fast-cull semantics); RTMI_FLAG_SYNC = per-lane traversal; RTMI_FLAG_ASYNC =
    // per-lane state machine (kept for comparison)
    const bool async = (p->flags & RTMI_FLAG_ASYNC) != 0u;
    const bool coop_ok = s->meta.n_prims < (1u << 22) && s->meta.n_nodes < (1u << 25);
    const bool coop = fast && !sync && !async && coop_ok;
    P.status = s->status;
    P.coop_cap = 64u * (s->meta.max_bvh_depth + 2u);
    const size_t coop_lds = (size_t)WAVES_PER_BLOCK * (2u * P.coop_cap + 64u * 12u + 128u) * sizeof(uint32_t);
#define RTMI_LAUNCH(KERN, F, S, PR, LDS) hipLaunchKernelGGL((KERN<F, S, PR>), grid, block, LDS, stream, s->dev, C, P, s->partial)
#define RTMI_LAUNCH_COOP(S, PR, W)                                                                                       \
    do {                                                                                                                 \
        if (coop_lds > 48u * 1024u)                                                                                      \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&rtmi_render_coop<S, PR, W>),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds));                     \
        hipLaunchKernelGGL((rtmi_render_coop<S, PR, W>), grid, block, coop_lds, stream, s->dev, C, P, s->partial);       \
    } while (0)
    if (coop) {
        const uint32_t wps = (p->flags >> 8) & 7u; // experiment knob: requested waves per SIMD (0 = default)
        if (prof) RTMI_LAUNCH_COOP(false, true, 3);
        else if (sigf) RTMI_LAUNCH_COOP(true, false, 4);
        else if (wps == 3) RTMI_LAUNCH_COOP(false, false, 3);
        else if (wps == 5) RTMI_LAUNCH_COOP(false, false, 5);
        else RTMI_LAUNCH_COOP(false, false, 4);
    } else if (!async) {
        if (prof) { if (fast) RTMI_LAUNCH(rtmi_render_kernel, true, false, true, 0); else RTMI_LAUNCH(rtmi_render_kernel, false, false, true, 0); }
        else if (fast && sigf) RTMI_LAUNCH(rtmi_render_kernel, true, true, false, 0);
        else if (fast) RTMI_LAUNCH(rtmi_render_kernel, true, false, false, 0);
        else if (sigf) RTMI_LAUNCH(rtmi_render_kernel, false, true, false, 0);
        else RTMI_LAUNCH(rtmi_render_kernel, false, false, false, 0);
    } else {
        if (prof) { if (fast) RTMI_LAUNCH(rtmi_render_async, true, false, true, dyn_lds); else RTMI_LAUNCH(rtmi_render_async, false, false, true, dyn_lds); }
        else if (fast && sigf) RTMI_LAUNCH(rtmi_render_async, true, true, false, dyn_lds);
        else if (fast) RTMI_LAUNCH(rtmi_render_async, true, false, false, dyn_lds);
        else if (sigf) RTMI_LAUNCH(rtmi_render_async, false, true, false, dyn_lds);
        else RTMI_LAUNCH(rtmi_render_async, false, false, false, dyn_lds);
    }
#undef RTMI_LAUNCH
#undef RTMI_LAUNCH_COOP
    HIP_TRY(hipGetLastError());
    if (stats) HIP_TRY(hipEventRecord(s->ev[1], stream));
    const uint32_t ntex = P.ntiles_local * 64u;
    hipLaunchKernelGGL(rtmi_resolve_kernel, dim3((ntex + 255) / 256), dim3(256), 0, stream, s->partial,
                       reinterpret_cast<rtmi_texel *>(d_texels), P);
    HIP_TRY(hipGetLastError());
    if (stats) {
        HIP_TRY(hipEventRecord(s->ev[2], stream));
        HIP_TRY(hipEventSynchronize(s->ev[2]));
        float ms_r = 0.f, ms_all = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms_r, s->ev[0], s->ev[1]));
        HIP_TRY(hipEventElapsedTime(&ms_all, s->ev[0], s->ev[2]));
        stats->render_ms = ms_r;
        stats->kernel_ms = ms_all;
        // samples actually traced: pixels inside the image that belong to local tiles
        uint64_t pix = 0;
        const uint32_t txn = tiles_x_of(p);
        for (uint32_t lt = 0; lt < P.ntiles_local; lt++) {
            const uint32_t t = lt * p->tile_world + p->tile_rank;
            const uint32_t ty = t / txn, tx = t % txn;
            const uint32_t w = (tx * RTMI_TILE + RTMI_TILE <= p->nx) ? RTMI_TILE : p->nx - tx * RTMI_TILE;
            const uint32_t h = (ty * RTMI_TILE + RTMI_TILE <= p->ny) ? RTMI_TILE : p->ny - ty * RTMI_TILE;
            pix += (uint64_t)w * h;
        }
        stats->samples = pix * p->ns;
        stats->tiles = P.ntiles_local; stats->chunks = chunks; stats->blocks = blocks; stats->reserved = 0;
        unsigned int st = 0;
        HIP_TRY(hipMemcpy(&st, s->status, sizeof(st), hipMemcpyDeviceToHost));
        if (st != 0) return fail(RTMI_ERR_DEVICE, "cooperative traversal pool overflow (results invalid): use RTMI_FLAG_SYNC");
    }
    return RTMI_OK;
}

extern "C" int rtmi_untile(const rtmi_render_params *p, const rtmi_texel *g, float *out_linear, uint8_t *out_rgb8) {
    int rc = check_params(p);
    if (rc) return rc;
    if (!g) return fail(RTMI_ERR_INVALID, "gathered buffer is NULL");
    const uint32_t txn = tiles_x_of(p), tyn = tiles_y_of(p);
    const size_t stride = (size_t)local_tiles_of(p, 0) * 64; // every rank padded to rank 0's size
    for (uint32_t ty = 0; ty < tyn; ty++)
        for (uint32_t tx = 0; tx < txn; tx++) {
            const uint32_t t = ty * txn + tx;
            const uint32_t rank = t % p->tile_world, lt = t / p->tile_world;
            const rtmi_texel *src = g + rank * stride + (size_t)lt * 64;
            for (uint32_t ly = 0; ly < RTMI_TILE; ly++) {
                const uint32_t row = ty * RTMI_TILE + ly;
                if (row >= p->ny) break;
                for (uint32_t lx = 0; lx < RTMI_TILE; lx++) {
                    const uint32_t px = tx * RTMI_TILE + lx;
                    if (px >= p->nx) break;
                    const rtmi_texel &e = src[ly * RTMI_TILE + lx];
                    const size_t o = ((size_t)row * p->nx + px) * 3;
                    if (out_linear) { out_linear[o] = e.r; out_linear[o + 1] = e.g; out_linear[o + 2] = e.b; }
                    if (out_rgb8) {
                        out_rgb8[o] = (uint8_t)(e.rgb8 & 255u);
                        out_rgb8[o + 1] = (uint8_t)((e.rgb8 >> 8) & 255u);
                        out_rgb8[o + 2] = (uint8_t)((e.rgb8 >> 16) & 255u);
                    }
                }
            }
        }
    return RTMI_OK;
}

extern "C" int rtmi_render(rtmi_scene *s, const rtmi_camera *cam, const rtmi_render_params *p_in, float *out_linear,
                           uint8_t *out_rgb8, uint64_t *out_path_sig, rtmi_stats *stats) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    int rc = check_params(p_in);
    if (rc) return rc;
    if (p_in->tile_world != 1) return fail(RTMI_ERR_INVALID, "rtmi_render renders the whole image: tile_world must be 1");
    HIP_TRY(hipSetDevice(s->device));
    rtmi_render_params p = *p_in;
    const size_t ntex = (size_t)rtmi_local_tiles(&p) * 64;
    if (ntex > s->texel_count) {
        if (s->texels) { HIP_TRY(hipFree(s->texels)); s->texels = nullptr; s->texel_count = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->texels), ntex * sizeof(rtmi_texel)));
        s->texel_count = ntex;
    }
    unsigned long long *d_sig = nullptr;
    p.flags &= ~RTMI_FLAG_PATH_SIG;
    p.path_sig = 0;
    if (out_path_sig) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_sig), ntex * sizeof(unsigned long long)));
        p.flags |= RTMI_FLAG_PATH_SIG;
        p.path_sig = reinterpret_cast<uint64_t>(d_sig);
    }
    rtmi_stats local{};
    rc = rtmi_render_device(s, cam, &p, s->texels, nullptr, stats ? stats : &local);
    if (rc) { if (d_sig) (void)hipFree(d_sig); return rc; }
    std::vector<rtmi_texel> host(ntex);
    HIP_TRY(hipMemcpy(host.data(), s->texels, ntex * sizeof(rtmi_texel), hipMemcpyDeviceToHost));
    if (out_path_sig) {
        std::vector<unsigned long long> hs(ntex);
        hipError_t e = hipMemcpy(hs.data(), d_sig, ntex * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipFree(d_sig);
        if (e != hipSuccess) return fail(RTMI_ERR_DEVICE, "copying the path signature failed");
        const uint32_t txn = tiles_x_of(&p);
        for (uint32_t row = 0; row < p.ny; row++)
            for (uint32_t px = 0; px < p.nx; px++) {
                const uint32_t t = (row / RTMI_TILE) * txn + px / RTMI_TILE;
                out_path_sig[(size_t)row * p.nx + px] = hs[(size_t)t * 64 + (row % RTMI_TILE) * RTMI_TILE + px % RTMI_TILE];
            }
    }
    return rtmi_untile(&p, host.data(), out_linear, out_rgb8);
}

// P3 writer — tests/test.rs:59,79
extern "C" size_t rtmi_ppm_p3(uint32_t nx, uint32_t ny, const uint8_t *rgb8, char *buf, size_t cap) {
    const size_t need = 40 + (size_t)nx * ny * 12;
    if (!buf || cap < need || !rgb8) return need;
    size_t n = (size_t)snprintf(buf, cap, "P3\n%u %u\n255\n", nx, ny);
    static const char digits[] = "0123456789";
    const size_t npx = (size_t)nx * ny;
    char *w = buf + n;
    for (size_t i = 0; i < npx * 3; i++) {
        const unsigned v = rgb8[i];
        if (v >= 100) { *w++ = digits[v / 100]; *w++ = digits[(v / 10) % 10]; *w++ = digits[v % 10]; }
        else if (v >= 10) { *w++ = digits[v / 10]; *w++ = digits[v % 10]; }
        else { *w++ = digits[v]; }
        *w++ = (i % 3 == 2) ? '\n' : ' ';
    }
    return (size_t)(w - buf);
}

// ---- probes (parity tests call these through the C ABI) ------------------------------
extern "C" int rtmi_probe_math(int op, const float *x, const float *y, float *out, uint32_t n) {
    if (rtmi_device_count() <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available");
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dx), n * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dy), n * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dout), n * 4));
    HIP_TRY(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dy, y ? y : x, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtmi_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, dx, dy, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return RTMI_OK;
}
extern "C" int rtmi_probe_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out, uint32_t n) {
    if (rtmi_device_count() <= 0) return fail(RTMI_ERR_DEVICE, "no HIP device available");
    uint32_t *dc = nullptr, *dk = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dc), n * 16));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dk), n * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dout), n * 16));
    HIP_TRY(hipMemcpy(dc, ctr, n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dk, key, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rtmi_philox_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dc, dk, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * 16, hipMemcpyDeviceToHost));
    (void)hipFree(dc); (void)hipFree(dk); (void)hipFree(dout);
    return RTMI_OK;
}
