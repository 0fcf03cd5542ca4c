// rtmi_shade.hpp — textures, Perlin noise, materials, camera sample, medium sample, shading of one hit.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include "rtmi_rng.hpp"
#include "rtmi_geom.hpp"

// ----------------------------------------------------------------------------------
// textures — src/texture.rs, src/perlin.rs
// ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t as_usize_u32(float x) { // Rust `as usize`, see DESIGN.md
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0u; // fp32 values >= 2^32 are multiples of 512: low 8 bits are 0
    return (uint32_t)x;
}
// ImageTexture's `(x) as usize` followed by the clamp to n - 1 (texture.rs:91-101): Rust's cast saturates, so any
// x >= n — including +inf and values beyond 2^32, where as_usize_u32's wrap to 0 would pick the FIRST texel — lands
// on n - 1; negative and NaN give 0.
__device__ __forceinline__ uint32_t as_image_index(float x, uint32_t n) {
    if (!(x > 0.0f)) return 0u;
    if (x >= (float)n) return n - 1u;
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
// One corner term of Perlin::noise for octave point tp (perlin.rs:38-56, 76-97), as perlin_noise evaluates it: the
// cooperative turbulence below computes the 56 terms of a turb(p, 7) on 56 lanes with the same operations.
__device__ __forceinline__ float perlin_corner_term(const rtmi_perlin *pn, F3 tp, int di, int dj, int dk) {
    const float fx = __builtin_floorf(tp.x), fy = __builtin_floorf(tp.y), fz = __builtin_floorf(tp.z);
    const float u = tp.x - fx, v = tp.y - fy, w = tp.z - fz;
    const uint32_t i = as_usize_u32(fx), j = as_usize_u32(fy), k = as_usize_u32(fz);
    const float uu = u * u * (3.0f - 2.0f * u);
    const float vv = v * v * (3.0f - 2.0f * v);
    const float ww = w * w * (3.0f - 2.0f * w);
    const float4 *rv = reinterpret_cast<const float4 *>(pn->ranvec);
    const int h = pn->perm[(i + (uint32_t)di) & 255u] ^ pn->perm[256 + ((j + (uint32_t)dj) & 255u)] ^
                  pn->perm[512 + ((k + (uint32_t)dk) & 255u)];
    const float4 c = rv[h];
    const float wx = u - (float)di, wy = v - (float)dj, wz = w - (float)dk;
    const float fi = di ? uu : (1.0f - uu);
    const float fj = dj ? vv : (1.0f - vv);
    const float fk = dk ? ww : (1.0f - ww);
    return fi * fj * fk * (c.x * wx + c.y * wy + c.z * wz);
}
// Perlin::turb(p, 7) of ONE point for the whole wavefront (every lane calls it with the same pn and p; every lane
// receives the value).  Lane 8*o + c evaluates corner c of octave o (tp = p * 2^o: the repeated doubling of
// perlin.rs:107 is exact), lanes 0..6 add the eight terms of their octave in the reference's order (di, dj, dk
// ascending, starting from 0.0), and the octaves are combined in order with the weights 1, 1/2, ... — the same
// operations on the same values as perlin_turb, so the same bits; 2 dependent table look-ups instead of 14.
// `scratch`: 64 floats of LDS private to the wavefront.
// out of line: inlined, its registers cost every scene 1.5-2 % (spills in the shading code) whether it has a
// NoiseTexture or not; the call is taken only when some lane of the batch wants turbulence
__device__ __attribute__((noinline)) float perlin_turb7_wave(const rtmi_perlin *pn, F3 p, float *scratch) {
    const int lane = threadIdx.x & 63;
    const int o = lane >> 3, c = lane & 7;
    float term = 0.0f;
    if (lane < 56) {
        const float s = (float)(1u << o);
        term = perlin_corner_term(pn, f3(p.x * s, p.y * s, p.z * s), (c >> 2) & 1, (c >> 1) & 1, c & 1);
    }
    scratch[lane] = term;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float oct = 0.0f;
    if (lane < 7) {
#pragma unroll
        for (int k = 0; k < 8; k++) oct += scratch[lane * 8 + k];
    }
    float accum = 0.0f, weight = 1.0f;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        accum += weight * __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(oct), i));
        weight *= 0.5f;
    }
    __builtin_amdgcn_wave_barrier(); // scratch may be rewritten by the next call
    return __builtin_fabsf(accum);
}
#ifndef RTMI_COOP_NOISE_MAX
#define RTMI_COOP_NOISE_MAX 10 // more lanes than this want turbulence at once: every lane evaluates its own (cheaper)
#endif
// Texture::value — texture.rs:21-25 (Solid), :39-48 (Checker), :65-71 (Noise), :86-108 (Image) — for the whole
// wavefront: ALL 64 lanes call it, lanes with want = true receive their value.  Perlin turbulence (7 octaves x 8
// corners, two dependent table look-ups each: ~1700 instructions and 14 memory latencies per lane) is what a few
// lanes of a shading batch need while the others wait, so those few points are evaluated one after the other by
// all 64 lanes together (perlin_turb7_wave); measured on final_scene: Perlin cost 4.3 % of the frame.
__device__ __forceinline__ F3 tex_value_wave(const DevScene &sc, bool want, rtmi_texture t, float u, float v, F3 p, float *scratch) {
    F3 out = f3(1, 1, 1);
    bool noise = false;
    int table = 0;
    float nscale = 0.0f;
    if (want) { // t = the material's texture record (it came with the shading record)
        for (int guard = 0; guard < 16 && t.kind == RTMI_TEX_CHECKER; guard++) {
            const float s = rtmi_sinf(10.0f * p.x) * rtmi_sinf(10.0f * p.y) * rtmi_sinf(10.0f * p.z);
            t = sc.texs[s < 0.0f ? t.i0 : t.i1];
        }
        if (t.kind == RTMI_TEX_NOISE) {
            noise = true; table = t.i0; nscale = t.f0;
        } else if (t.kind == RTMI_TEX_IMAGE) {
            const rtmi_image im = sc.images[t.i0];
            uint32_t i = as_image_index(u * (float)im.nx, im.nx);
            uint32_t j = as_image_index((1.0f - v) * (float)im.ny, im.ny);
            const uint8_t *px = sc.image_data + im.offset + 3ull * i + 3ull * im.nx * j;
            out = f3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
        } else {
            out = f3(t.f0, t.f1, t.f2);
        }
    }
    unsigned long long m = __ballot(noise);
    if (m != 0ull) { // wave-uniform
        float turb = 0.0f;
        if (__popcll(m) > RTMI_COOP_NOISE_MAX) {
            if (noise) turb = perlin_turb(sc.perlin + table, p, 7);
        } else {
            const int lane = threadIdx.x & 63;
            while (m != 0ull) { // wave-uniform loop over the lanes that want turbulence
                const int L = (int)__ffsll((long long)m) - 1;
                m &= m - 1ull;
                const F3 q = f3(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(p.x), L)),
                                __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(p.y), L)),
                                __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(p.z), L)));
                const int tab = __builtin_amdgcn_readlane(table, L);
                const float r = perlin_turb7_wave(sc.perlin + tab, q, scratch);
                if (lane == L) turb = r;
            }
        }
        if (noise) {
            const float g = 0.5f * (1.0f + rtmi_sinf(nscale * p.x + 5.0f * turb));
            out = f3(g, g, g);
        }
    }
    return out;
}

// get_sphere_uv — sphere.rs:9-15 (FRAC_2_PI, sic)
// book = RTMI_FLAG_UV_BOOK (opt-in): theta + pi/2 as in the book instead of the reference's 2/pi
__device__ __forceinline__ void sphere_uv(F3 n, bool book, float &u, float &v) {
    const float phi = rtmi_atan2f(n.z, n.x);
    const float theta = rtmi_asinf(n.y);
    u = 1.0f - (phi + RTMI_PI_F) / (2.0f * RTMI_PI_F);
    v = (theta + (book ? RTMI_PIO2_F : RTMI_2_OVER_PI_F)) / RTMI_PI_F;
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
// opt-in extension (RTMI_FLAG_SKY): the background gradient the reference keeps commented out at
// color.rs:18-20; a path that leaves the scene adds T * sky(direction) instead of black (:21)
__device__ __forceinline__ F3 sky_color(F3 d) {
    const F3 unit = normalize(d);
    const float t = 0.5f * (unit.y + 1.0f);
    const float a = 1.0f - t;
    return f3(a * 1.0f + t * 0.5f, a * 1.0f + t * 0.7f, a * 1.0f + t * 1.0f);
}

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
template <typename RngT>
__device__ __forceinline__ void camera_sample(const DevCamera &cam, const DevParams &P, RngT &g, uint32_t k0, uint32_t k1,
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
// `dn` = |direction| of the WORLD ray (medium.rs:38 `ray.direction().norm()`), passed by the caller as sqrt(W.a): W.a is
// dot(d, d) already (ray_derive), the same expression norm() squares — same bits without a second dot product.
// |direction| of the ray a ConstantMedium sees (medium.rs:38): the world ray's, or — for a medium that sits INSIDE the
// first `outer` transforms of its item (rtmi.h, RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) — the ray those wrappers hand down.
// Wave-uniform `flags`; the rare path reads the transforms from memory.  INST = false: the instantiation for scenes
// without such media (and without instanced primitives) carries none of this — merely present, the rare path cost
// final_scene 1.7 % (profiles/r03_experiments/medium_outer_ab.log).
template <bool INST = true>
__device__ __forceinline__ float medium_dir_norm(const DevScene &sc, uint32_t flags, int xform_first, const RayF &W) {
    if (!INST) return __builtin_sqrtf(W.a);
    const int outer = (int)((flags >> RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) & 15u);
    if (outer == 0) return __builtin_sqrtf(W.a); // W.a is dot(d, d): the expression norm() squares
    F3 o = W.o, d = W.d;
    if (!xform_ray(sc.xforms, xform_first, outer, o, d)) return __builtin_sqrtf(W.a); // translations only: same direction
    return norm(d);
}
// Gate of a DEFERRED medium item — a ConstantMedium that was a child of a BVHNode (rtmi.h): the box of that node, tested as
// BVHNode::hit tests its own box (bvh.rs:71; a child is reached iff every ancestor's box passes, i.e. iff its parent's
// does) with the ray the first G transforms of the item hand down — the transforms of the enclosing BVH item — and the
// interval (t_min, T0) the BVH was entered with.  The box travels as the gate of the item's first primitive.
__device__ __forceinline__ bool deferred_gate(const DevScene &sc, const rtmi_item &I, const RayF &W, float t_min, float t0) {
    const int G = (int)((I.flags >> RTMI_ITEMFLAG_GATE_OUTER_SHIFT) & 15u);
    RayF Rg = W;
    if (G > 0 && xform_ray(sc.xforms, I.xform_first, G, Rg.o, Rg.d)) ray_derive(Rg);
    float4 g0, g1;
    if (I.kind == RTMI_ITEM_LIST) { // a medium around primitives (a primitive of a list scan): the gate of its first primitive
        const float4 *rec = sc.leaf_rec + (size_t)I.first * 5;
        g0 = rec[3]; g1 = rec[4];
    } else { // geometry = a BVHNode (an instanced subtree, a medium's boundary): the two records behind its chain ({kind, x, y, z})
        const float4 *rec = reinterpret_cast<const float4 *>(sc.xforms + I.xform_first + I.xform_count);
        g0 = make_float4(rec[0].y, rec[0].z, rec[0].w, 0.0f); g1 = make_float4(rec[1].y, rec[1].z, rec[1].w, 0.0f);
    }
    return aabb_hit(g0.x, g0.y, g0.z, g1.x, g1.y, g1.z, Rg, t_min, t0);
}
// A hit of a DEFERRED BVH item (an instanced subtree that was a child of a BVHNode) against the closest hit so far: the
// fold of bvh.rs:75-81 — closer wins, an exact tie goes to the LATER child — between the subtree (position: `rank` leaves
// of the enclosing tree precede it; the same for the result of a list scan that was a child of a BVHNode, LISTSCAN_END) and whatever holds the closest hit: something before the group (the list scan accepts
// every hit its t_max lets through, hittable.rs:40-44), a leaf of the enclosing tree (later iff its index >= I.count), or
// an earlier deferred item of the group (earlier).
__device__ __forceinline__ bool deferred_bvh_wins(int rank, float t, float closest, int best_item, int best_pf, int grp_first,
                                                  bool grp_tree) {
    if (t < closest) return true;
    if (t != closest) return false;
    if (grp_tree && best_item == grp_first) return (best_pf >> 3) < rank; // a leaf of the enclosing tree holds the closest hit
    return true; // something before the group, or an earlier deferred item of it
}
// The scan of a HittableList that was a child of a BVHNode and holds media (LISTSCAN items, rtmi.h): its closest hit so far
// and who holds it; `fold` at the terminator item.
struct ListScan {
    float cl;
    int item, pf;
    bool medium, has;
};
__device__ __forceinline__ void listscan_fold(const ListScan &ls, int rank, float &closest, int &best_item, int &best_pf, bool &best_medium,
                                              int grp_first, bool grp_tree) {
    if (!ls.has) return;
    const bool wins = ls.medium ? ls.cl < closest : deferred_bvh_wins(rank, ls.cl, closest, best_item, best_pf, grp_first, grp_tree);
    if (wins) { closest = ls.cl; best_item = ls.item; best_pf = ls.pf; best_medium = ls.medium; }
}
template <typename RngT>
__device__ __forceinline__ bool medium_sample(float t1, float t2, float t_min, float closest, float dn,
                                              float neg_inv_density, RngT &g, uint32_t k0, uint32_t k1, float &t_out) {
    if (t1 < t_min) t1 = t_min;
    if (t2 > closest) t2 = closest;
    if (t1 < t2) {
        const float dist_inside = (t2 - t1) * dn;
        const float hit_distance = neg_inv_density * rtmi_logf(rng_uniform(g, k0, k1));
        if (hit_distance < dist_inside) {
            t_out = t1 + hit_distance / dn;
            return true;
        }
    }
    return false;
}

// A ConstantMedium whose boundary is a ConstantMedium (RTMI_ITEMFLAG_NESTED_MEDIUM, rtmi.h).  The outer medium's two boundary
// queries (medium.rs:30-31) are two calls of the inner medium's hit(): (ray, -MAX, MAX) and (ray, hit1.t + 0.0001, MAX).
// Either call queries the geometry with the same two intervals — (t1, t2) on entry, found once by the caller — clamps, and
// draws one random number when its interval is not empty; the second call happens only when the first returned a hit.  On
// return (t1, t2) are the two random distances: the interval the outer medium then clamps and samples (its own draw).
// The inner medium sees the same ray as the outer one (no wrappers in between: refused by the lowerings), hence `dn`.
template <typename RngT>
__device__ __forceinline__ bool nested_medium_interval(const DevScene &sc, const rtmi_item &I, float dn, RngT &g, uint32_t k0, uint32_t k1,
                                                       float &t1, float &t2) {
    const int at = I.xform_first + I.xform_count + (((I.flags & RTMI_ITEMFLAG_DEFERRED) && I.kind == RTMI_ITEM_BVH) ? 2 : 0);
    const float inner_nid = sc.xforms[at].x;
    float ta, tb;
    if (!medium_sample(t1, t2, -RTMI_FLT_MAX, RTMI_FLT_MAX, dn, inner_nid, g, k0, k1, ta)) return false;
    if (!medium_sample(t1, t2, ta + 0.0001f, RTMI_FLT_MAX, dn, inner_nid, g, k0, k1, tb)) return false;
    t1 = ta; t2 = tb;
    return true;
}

// HitRecord of the closest hit (hittable.rs:9-16), built once, then
// color(): emitted + attenuation * color(scattered) — color.rs:8-15, in throughput form.
// ALL 64 lanes call this together (the texture lookup is a wavefront operation, tex_value_wave); lanes with
// active = true hold a hit to shade.  Returns true when the lane's path continues (pa holds the scattered ray),
// false when it ended (or the lane was not active).  `scratch`: 64 floats of LDS private to the wavefront.
template <typename RngT, bool INST = true>
__device__ __forceinline__ bool shade_hit(const DevScene &sc, uint32_t max_depth, uint32_t ext, RngT &g, uint32_t k0, uint32_t k1,
                                          bool active, float closest, int best_item, int best_pf, bool best_medium, Path &pa,
                                          float *scratch) {
    F3 hp = f3(0, 0, 0), hn = f3(1, 0, 0);
    float hu = 0.0f, hv = 0.0f;
    rtmi_material M;
    M.kind = -1; M.tex = 0; M.param = 0.0f; M.flags = 0u;
    bool can_scatter = false, textured = false;
    F3 rs = f3(0, 0, 0);
    rtmi_texture T0;
    T0.kind = RTMI_TEX_SOLID; T0.i0 = 0; T0.i1 = 0; T0.pad = 0; T0.f0 = 1.0f; T0.f1 = 1.0f; T0.f2 = 1.0f; T0.f3 = 0.0f;
    if (active) {
        // of the item only {flags, xform_first, xform_count, medium_material} are needed: one 16-B fetch
        struct __attribute__((aligned(4))) ItemWords { int x, y, z, w; }; // 4-byte aligned: offset 12 of a 64-B record
        const char *ditem = reinterpret_cast<const char *>(sc.items + best_item);
        const ItemWords IW = *reinterpret_cast<const ItemWords *>(ditem + 12);
        // the item's first two transforms travel with it (DevItem): no fetch that depends on IW
        const rtmi_xform IX0 = *reinterpret_cast<const rtmi_xform *>(ditem + 64), IX1 = *reinterpret_cast<const rtmi_xform *>(ditem + 80);
        const uint32_t iflags = (uint32_t)IW.x;
        const int xform_first = IW.y, xform_count = IW.z;
        float4 rec_mat, rec_t0, rec_t1; // material record + its texture record, fetched with the hit's geometry
        if (best_medium) {
            const float4 *rec = sc.shade_mat + (size_t)IW.w * 4; // medium_material
            rec_mat = rec[1]; rec_t0 = rec[2]; rec_t1 = rec[3];
            hp = pa.ro + pa.rd * closest;      // ray.pointing_at(t) — medium.rs:47
            hn = f3(1.0f, 0.0f, 0.0f);         // medium.rs:48
            const int outer = INST ? (int)((iflags >> RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) & 15u) : 0;
            if (outer > 0) { // the medium sits inside `outer` wrappers: its point is taken on THEIR ray and handed back
                F3 lo = pa.ro, ld = pa.rd;
                xform_ray(sc.xforms, xform_first, outer, lo, ld);
                hp = lo + ld * closest;
                xform_hit(sc.xforms, xform_first, outer, hp, hn);
            }
        } else {
            const int idx = best_pf >> 3, face = best_pf & 7;
            const float4 *rec = sc.shade_prim + (size_t)idx * 4;
            const float4 A = rec[0];
            rec_mat = rec[1]; rec_t0 = rec[2]; rec_t1 = rec[3];
            // (meta and plane B from the leaf record; fetched with the rest, not after the type is known: one latency, not two)
            const PrimRec *pr = reinterpret_cast<const PrimRec *>(sc.leaf_rec + (size_t)idx * 5);
            const rtmi_prim_meta PM = pr->M;
            const float4 PB = pr->B;
            F3 lo = pa.ro, ld = pa.rd;
            if (xform_count > 0) xform_ray_item(sc.xforms, xform_first, xform_count, IX0, IX1, lo, ld);
            // an instanced primitive's own chain, inside the item's frame (rtmi.h)
            const int pxf_count = (INST && sc.has_prim_xf) ? (int)((PM.flags >> RTMI_PRIMFLAG_XF_COUNT_SHIFT) & RTMI_PRIM_XF_MAX) : 0;
            const int pxf_first = (int)(PM.flags >> RTMI_PRIMFLAG_XF_FIRST_SHIFT);
            if (pxf_count > 0) xform_ray(sc.xforms, pxf_first, pxf_count, lo, ld);
            const bool needs_uv = (__float_as_uint(rec_mat.w) & RTMI_MATFLAG_NEEDS_UV) != 0u;
            hp = lo + ld * closest; // ray.pointing_at(t)
            if (PM.type == RTMI_PRIM_SPHERE || PM.type == RTMI_PRIM_MSPHERE) {
                F3 c = f3(A.x, A.y, A.z);
                if (PM.type == RTMI_PRIM_MSPHERE) c = moving_center(A, PB, PM.inv_dt, pa.rtime);
                hn = vdiv(hp - c, A.w); // sphere.rs:50 — outward, never face-forwarded
                if (needs_uv) {
                    // the flag is re-read here, behind a compiler barrier: left alone, the constant it selects is hoisted
                    // out of the path loop into a VGPR at kernel entry and — the 129th of 128 — spilled to scratch
                    uint32_t ext_here = ext;
                    asm volatile("" : "+s"(ext_here));
                    sphere_uv(hn, (ext_here & RTMI_EXT_UV_BOOK) != 0u, hu, hv);
                }
            } else {
                int plane;
                float x0, y0, x1, y1;
                if (PM.type == RTMI_PRIM_RECT) {
                    plane = (int)((PM.flags >> RTMI_PRIMFLAG_PLANE_SHIFT) & 3u);
                    x0 = A.x; y0 = A.y; x1 = A.z; y1 = A.w;
                } else { // cube face -> its rect (cube.rs:21-74)
                    const float ax = A.x, ay = A.y, az = A.z, bx = A.w, by = PB.x, bz = PB.y;
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
            if (pxf_count > 0) xform_hit(sc.xforms, pxf_first, pxf_count, hp, hn); // innermost frames first
            if (xform_count > 0) xform_hit_item(sc.xforms, xform_first, xform_count, IX0, IX1, hp, hn);
            if (((PM.flags ^ iflags) & 1u) != 0u) hn = -hn; // FlipNormals — hittable.rs:78-83
        }

        // Material::emitted / Material::scatter (material.rs).  The rejection sampler and the texture
        // lookup are needed by several materials; they are evaluated ONCE here for all lanes that need
        // them (a per-material copy would run the same long code serially for each lane subset).  The
        // draw order per lane is unchanged: Lambertian/Isotropic/fuzzy Metal draw only inside the sampler,
        // Dielectric draws its single uniform, DiffuseLight draws nothing.
        M.kind = (int32_t)__float_as_uint(rec_mat.x); M.tex = (int32_t)__float_as_uint(rec_mat.y);
        M.param = rec_mat.z; M.flags = __float_as_uint(rec_mat.w);
        T0.kind = (int32_t)__float_as_uint(rec_t0.x); T0.i0 = (int32_t)__float_as_uint(rec_t0.y);
        T0.i1 = (int32_t)__float_as_uint(rec_t0.z); T0.pad = 0;
        T0.f0 = rec_t1.x; T0.f1 = rec_t1.y; T0.f2 = rec_t1.z; T0.f3 = rec_t1.w;
        can_scatter = pa.depth < max_depth; // color.rs:9
        textured = M.kind == RTMI_MAT_LAMBERTIAN || M.kind == RTMI_MAT_METAL || M.kind == RTMI_MAT_ISOTROPIC;
        const bool want_sample = can_scatter && (M.kind == RTMI_MAT_LAMBERTIAN || M.kind == RTMI_MAT_ISOTROPIC ||
                                                 (M.kind == RTMI_MAT_METAL && M.param > 0.0f));
        if (want_sample) rs = random_in_unit_sphere(g, k0, k1);
    }
    const int kind = M.kind;
    const bool want_tex = active && (kind == RTMI_MAT_DIFFUSE_LIGHT || (can_scatter && textured));
    const F3 tv = tex_value_wave(sc, want_tex, T0, hu, hv, hp, scratch); // every lane of the wavefront
    bool scattered = false;
    if (active) {
        if (kind == RTMI_MAT_DIFFUSE_LIGHT) pa.L = pa.L + pa.T * tv; // material.rs:148-150
        const F3 rd = pa.rd;
        F3 nd = rd, att = f3(1, 1, 1);
        // opt-in RTMI_FLAG_FACE_FORWARD (wave-uniform): the opaque materials see the normal turned against the ray
        if ((ext & RTMI_EXT_FACE_FORWARD) && kind != RTMI_MAT_DIELECTRIC && dot(rd, hn) > 0.0f) hn = -hn;
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
    }
    return scattered;
}

// ----------------------------------------------------------------------------------
// Work distribution.  A UNIT = (local tile, sample chunk): the items (pixel of the tile, sample of
// the chunk), numbered sample-major (item k -> sample s_begin + k / n_valid, k % n_valid-th pixel
// of the tile that lies inside the image).  Two levels, both dynamic:
//   * lanes: a lane whose path ended takes the next item of the wavefront's unit, whatever its
//     pixel (a static lane = pixel assignment leaves fast lanes idle ~2/sqrt(spp) of the time);
//   * wavefronts are persistent: when its unit is exhausted the wavefront takes the next unit from
//     a global counter while its other lanes are still finishing paths of the previous one, so a
//     wavefront drains only once, at the end of the launch, and units can be small (short tail).
// Every finished path stores its radiance in the per-sample buffer samples[tile][sample][pixel];
// the resolve kernel adds them in sample order (tests/test.rs:65-70), so the result does not
// depend on which lane rendered what, nor on chunking or scheduling.
// ----------------------------------------------------------------------------------
struct WaveWork { // wave-uniform (kept in SGPRs: every field passes through readfirstlane)
    uint32_t ltile, ps_base, obase, x0, y0, cols, n_valid, next, total;
};
__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ WaveWork wave_work(const DevParams &P, uint32_t unit) {
    WaveWork w;
    const bool has_work = unit < P.ntiles_local * P.nchunks;
    const uint32_t chunk = unit / P.ntiles_local; // chunk-major: all tiles of chunk 0 first
    const uint32_t ltile = unit - chunk * P.ntiles_local;
    const uint32_t tile = ltile * P.tile_world + P.tile_rank;
    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const uint32_t x0 = tx * RTMI_TILE, y0 = ty * RTMI_TILE;
    const uint32_t cols = x0 < P.nx ? (P.nx - x0 < RTMI_TILE ? P.nx - x0 : RTMI_TILE) : 0u;
    const uint32_t rows = y0 < P.ny ? (P.ny - y0 < RTMI_TILE ? P.ny - y0 : RTMI_TILE) : 0u;
    const uint32_t s_begin = P.pass_s0 + chunk * P.chunk_spp;
    const uint32_t pass_end = P.pass_s0 + P.pass_cnt;
    uint32_t s_end = s_begin + P.chunk_spp;
    if (s_end > pass_end) s_end = pass_end;
    w.ltile = rfl(ltile);
    w.x0 = rfl(x0);
    w.y0 = rfl(y0);
    w.cols = rfl(cols);
    w.n_valid = rfl(cols * rows);
    w.ps_base = rfl(s_begin << 6);
    // slot of item ps = sample << 6 | pixel in the per-sample buffer: (ltile * stride + sample - pass_s0) * 64 + pixel
    // = obase + ps in uint32 arithmetic (the host keeps the buffer below 2^32 slots)
    w.obase = rfl((ltile * P.pass_stride - P.pass_s0) * 64u);
    w.total = rfl((has_work && s_end > s_begin) ? (s_end - s_begin) * cols * rows : 0u);
    w.next = 0u;
    return w;
}
// Persistent wavefront; all 64 lanes call this together.  Every lane with want = true receives the next item
// (sample s of pixel (px, j) of local tile ltile; oidx = its slot in the per-sample buffer) from the current
// unit or, when that is exhausted, from the next units of the global queue.  Returns false for lanes that
// wanted but found the queue empty: they are done for good.  The loop ends for every wavefront: the
// counter only grows and the number of units is fixed.
__device__ __forceinline__ bool work_take(WaveWork &w, bool &queue_empty, bool want, const DevParams &P, uint32_t &oidx,
                                          uint32_t &ltile, uint32_t &smp, uint32_t &px, uint32_t &j) {
    bool got = false;
    for (;;) {
        const bool still = want && !got;
        const unsigned long long m = __ballot(still);
        if (m == 0ull) break;
        if (w.next >= w.total) { // wave-uniform: unit exhausted, take the next one
            if (queue_empty) break;
            uint32_t u = 0u;
            if ((threadIdx.x & 63) == 0) u = atomicAdd(P.queue, 1u);
            u = rfl(u);
            if (u >= P.ntiles_local * P.nchunks) { queue_empty = true; break; }
            w = wave_work(P, u);
            continue;
        }
        const uint32_t k = w.next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        w.next = rfl(w.next + (uint32_t)__popcll(m));
        if (still && k < w.total) {
            uint32_t ps; // sample << 6 | pixel-in-tile (row * 8 + column)
            if (w.n_valid == 64u) { // wave-uniform: items are sample-major, 64 pixels per sample
                ps = w.ps_base + k;
            } else { // ragged tile on the right / bottom edge: a cols x rows rectangle of the 8x8 tile
                const uint32_t ds = k / w.n_valid, idx = k - ds * w.n_valid;
                const uint32_t row = idx / w.cols, col = idx - row * w.cols;
                ps = w.ps_base + (ds << 6) + (row * RTMI_TILE + col);
            }
            oidx = w.obase + ps;
            ltile = w.ltile;
            smp = ps >> 6;
            px = w.x0 + (ps & 7u);
            j = P.ny - 1u - (w.y0 + ((ps >> 3) & 7u)); // `for j in (0..ny).rev()` — tests/test.rs:62
            got = true;
        }
    }
    return got;
}
// `col += color(..)` — tests/test.rs:69: the path's radiance goes to its slot of the per-sample buffer
__device__ __forceinline__ void path_end(const DevParams &P, uint32_t oidx, const Path &pa) {
    P.samples[oidx] = Rad3{pa.L.x, pa.L.y, pa.L.z};
}
