// rtmi_geom.hpp — instance transforms, the per-frame ray, box / sphere / rect / cube tests.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include "rtmi_types.hpp"

// Wave-uniform reads (the item record, its transforms, the primitives of a list item: every lane reads the same
// address) go through the constant address space: the compiler then issues scalar loads (s_load_dwordx4 into
// SGPRs) instead of 64 identical vector loads — fewer VALU address computations, fewer VGPRs, the vector memory
// pipe left to the divergent gathers.  Legal because the scene arrays are never written while a render kernel
// runs.  Measured: final_scene +2.1 %, cornell_box +6 %.
#if defined(__HIP_DEVICE_COMPILE__) /* (the host pass of hipcc parses this header too) */
#define RTMI_UNIFORM_LOAD(T, ptr) (*reinterpret_cast<const T __attribute__((address_space(4))) *>(reinterpret_cast<uintptr_t>(ptr)))
#else
#define RTMI_UNIFORM_LOAD(T, ptr) (*(ptr))
#endif

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
// world -> object; returns true when the direction changed (a rotation was applied).  UNIFORM: `first` and `count`
// are the same in every lane (the item loop), so the records come through scalar loads.
template <bool UNIFORM = false>
__device__ __forceinline__ bool xform_ray(const rtmi_xform *xf, int first, int count, F3 &o, F3 &d) {
    bool rotated = false;
    for (int k = 0; k < count; k++) {
        const rtmi_xform X = UNIFORM ? RTMI_UNIFORM_LOAD(rtmi_xform, xf + first + k) : xf[first + k];
        switch (X.kind) {
        case RTMI_XF_TRANSLATE: o = o - f3(X.x, X.y, X.z); break;
        case RTMI_XF_ROTATE_X: rot_fwd(X.x, X.y, o.y, o.z); rot_fwd(X.x, X.y, d.y, d.z); rotated = true; break;
        case RTMI_XF_ROTATE_Y: rot_fwd(X.x, X.y, o.z, o.x); rot_fwd(X.x, X.y, d.z, d.x); rotated = true; break;
        default: rot_fwd(X.x, X.y, o.x, o.y); rot_fwd(X.x, X.y, d.x, d.y); rotated = true; break;
        }
    }
    return rotated;
}
// one transform, world -> object (traslate.rs:18-20, rotate.rs:85-92); returns true for a rotation
__device__ __forceinline__ bool xform_ray_one(const rtmi_xform &X, F3 &o, F3 &d) {
    switch (X.kind) {
    case RTMI_XF_TRANSLATE: o = o - f3(X.x, X.y, X.z); return false;
    case RTMI_XF_ROTATE_X: rot_fwd(X.x, X.y, o.y, o.z); rot_fwd(X.x, X.y, d.y, d.z); return true;
    case RTMI_XF_ROTATE_Y: rot_fwd(X.x, X.y, o.z, o.x); rot_fwd(X.x, X.y, d.z, d.x); return true;
    default: rot_fwd(X.x, X.y, o.x, o.y); rot_fwd(X.x, X.y, d.x, d.y); return true;
    }
}
__device__ __forceinline__ void xform_hit_one(const rtmi_xform &X, F3 &p, F3 &n) {
    switch (X.kind) {
    case RTMI_XF_TRANSLATE: p = p + f3(X.x, X.y, X.z); break;
    case RTMI_XF_ROTATE_X: rot_inv(X.x, X.y, p.y, p.z); rot_inv(X.x, X.y, n.y, n.z); break;
    case RTMI_XF_ROTATE_Y: rot_inv(X.x, X.y, p.z, p.x); rot_inv(X.x, X.y, n.z, n.x); break;
    default: rot_inv(X.x, X.y, p.x, p.y); rot_inv(X.x, X.y, n.x, n.y); break;
    }
}
// An item's chain with its first two transforms at hand (DevItem): same operations in the same order as xform_ray /
// xform_hit over xforms[first .. first + count)
template <bool UNIFORM = false>
__device__ __forceinline__ bool xform_ray_item(const rtmi_xform *xf, int first, int count, const rtmi_xform &X0, const rtmi_xform &X1,
                                               F3 &o, F3 &d) {
    bool rotated = false;
    if (count > 0) rotated = xform_ray_one(X0, o, d);
    if (count > 1) rotated = xform_ray_one(X1, o, d) || rotated;
    if (count > 2) rotated = xform_ray<UNIFORM>(xf, first + 2, count - 2, o, d) || rotated;
    return rotated;
}
__device__ __forceinline__ void xform_hit_item(const rtmi_xform *xf, int first, int count, const rtmi_xform &X0, const rtmi_xform &X1,
                                               F3 &p, F3 &n);
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

__device__ __forceinline__ void xform_hit_item(const rtmi_xform *xf, int first, int count, const rtmi_xform &X0, const rtmi_xform &X1,
                                               F3 &p, F3 &n) {
    if (count > 2) xform_hit(xf, first + 2, count - 2, p, n);
    if (count > 1) xform_hit_one(X1, p, n);
    if (count > 0) xform_hit_one(X0, p, n);
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
    // aabb.rs:41-43 returns false as soon as t_max <= t_min after an axis.  t_min never decreases and t_max never
    // increases from axis to axis (fmaxf / fminf, which also ignore a NaN slab distance exactly like f64::max / min),
    // so "after some axis" <=> "after the last axis": one comparison instead of three and their two mask ORs.
    float t0 = (mnx - r.o.x) * r.inv_d.x, t1 = (mxx - r.o.x) * r.inv_d.x;
    bool neg = r.inv_d.x < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    t0 = (mny - r.o.y) * r.inv_d.y; t1 = (mxy - r.o.y) * r.inv_d.y;
    neg = r.inv_d.y < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    t0 = (mnz - r.o.z) * r.inv_d.z; t1 = (mxz - r.o.z) * r.inv_d.z;
    neg = r.inv_d.z < 0.0f;
    t_min = fmaxf(t_min, neg ? t1 : t0);
    t_max = fminf(t_max, neg ? t0 : t1);
    return !(t_max <= t_min);
}

// Discriminant of Sphere::hit (sphere.rs:43: b*b - a*c with c = oc.oc - r*r) — arithmetic contract, substitution 5 (r04).
// Algebraically  b^2 - a (|oc|^2 - r^2)  =  a (r^2 - |oc - (b/a) d|^2):  the squared distance of the centre from the ray
// instead of the difference of two numbers of size |oc|^2 a.  In fp32 the literal form loses the hit point of a small
// sphere seen from afar: r = 10 at |oc| = 800 gives disc = 100 a +- 0.1 a, the hit point lands up to 5e-3 inside or
// outside the surface, and a scattered ray that starts inside meets the far side of the same sphere beyond t_min = 0.001 —
// measured against the f64 literal restatement of the reference (tests/test_gpu_f64_tolerance.py): final_scene's small
// spheres 10 % too dark, the image mean 0.55 % low.  With this form the error of disc is relative to r^2 a:
// the same image is within 0.003 % of the f64 mean.  Three explicit fma for l, two for |l|^2 (single roundings on both
// sides of the parity test, like rtmi_math.h): the instruction count of the literal form.
__device__ __forceinline__ float sphere_disc(const RayF &r, const F3 &oc, float b, float radius) {
    const float q = b * r.inv_a;
    const float lx = __builtin_fmaf(-q, r.d.x, oc.x), ly = __builtin_fmaf(-q, r.d.y, oc.y), lz = __builtin_fmaf(-q, r.d.z, oc.z);
    const float l2 = __builtin_fmaf(lz, lz, __builtin_fmaf(ly, ly, lx * lx));
    return r.a * (radius * radius - l2);
}
// Sphere::hit / MovingSphere::hit — src/sphere.rs:37-77, 122-164 (t only; the record is
// built once for the closest hit in finalize_hit)
__device__ __forceinline__ bool sphere_test(const RayF &r, F3 c, float radius, float t_min, float t_max, float &t_out) {
    F3 oc = r.o - c;
    float b = dot(oc, r.d);
    const float disc = sphere_disc(r, oc, b, radius);
    // select form (r03): both roots are evaluated and the first one inside (t_min, t_max) is kept, as sphere.rs:44-74
    // does with two early returns; the same comparisons on the same values (sqrt of a non-positive discriminant gives
    // NaN or 0 and is masked by disc > 0), without the three nested exec-mask regions
    const bool pos = disc > 0.0f;
    // (a wave-uniform skip of the square root when no lane has disc > 0 measured -0.5 % on final_scene, +0.2 % on
    // random_spheres: profiles/r03_experiments/sph_ab.log — not kept)
    const float sq = __builtin_sqrtf(disc);
    const float t1 = (-b - sq) * r.inv_a, t2 = (-b + sq) * r.inv_a;
    const bool ok1 = pos & (t1 < t_max) & (t1 > t_min);
    const bool ok2 = pos & (t2 < t_max) & (t2 > t_min);
    t_out = ok1 ? t1 : (ok2 ? t2 : t_out);
    return ok1 | ok2;
}
// ConstantMedium::hit's two boundary queries (medium.rs:29-30) against ONE static sphere:
//   boundary.hit(ray, -MAX, MAX) -> t1, then boundary.hit(ray, t1 + 0.0001, MAX) -> t2.
// Both evaluate the same roots; sphere_test's conditions are applied to them twice, in its order.
__device__ __forceinline__ void sphere_two_queries(const RayF &r, float4 A, bool &h1, float &t1, bool &h2, float &t2) {
    const F3 oc = r.o - f3(A.x, A.y, A.z);
    const float b = dot(oc, r.d);
    const float disc = sphere_disc(r, oc, b, A.w);
    h1 = false; h2 = false;
    if (disc > 0.0f) {
        const float sq = __builtin_sqrtf(disc);
        const float ta = (-b - sq) * r.inv_a, tb = (-b + sq) * r.inv_a;
        if (ta < RTMI_FLT_MAX && ta > -RTMI_FLT_MAX) { t1 = ta; h1 = true; }
        else if (tb < RTMI_FLT_MAX && tb > -RTMI_FLT_MAX) { t1 = tb; h1 = true; }
        if (h1) {
            const float lo = t1 + 0.0001f;
            if (ta < RTMI_FLT_MAX && ta > lo) { t2 = ta; h2 = true; }
            else if (tb < RTMI_FLT_MAX && tb > lo) { t2 = tb; h2 = true; }
        }
    }
}
// MovingSphere::center — src/sphere.rs:115-118 (contract: (time - t0) * inv_dt)
__device__ __forceinline__ F3 moving_center(float4 A, float4 B, float inv_dt, float time) {
    float f = (time - B.w) * inv_dt;
    return f3(A.x, A.y, A.z) + f3(B.x, B.y, B.z) * f;
}

// Rect::hit — src/rect.rs:39-69 with (k,a,b) = YZ:(0,1,2) ZX:(1,2,0) XY:(2,0,1)
// Branch-free form (r03).  The reference rejects iff  t < t_min || t > t_max || x < x0 || x > x1 || y < y0 || y > y1.
// In IEEE arithmetic with denormals kept  a < b  <=>  b - a > 0  exactly (a difference is zero only for equal operands,
// its sign is never lost), and a NaN operand makes the comparison false just as fmaxf ignores a NaN difference; so the
// same decision is  max(t_min - t, t - t_max, x0 - x, x - x1, y0 - y, y - y1) > 0  (all NaN: not rejected, like the six
// false comparisons).  Six subtractions and three v_max3 replace six compares, their five scalar mask combinations and
// the two divergent early-outs: on gfx950 a scalar instruction occupies its issue port for 4 cycles, shared by the
// wavefronts of a SIMD, a vector instruction for 2 (tools/micro/issue_share.hip) — the scalar side is what binds here.
template <int P>
__device__ __forceinline__ bool rect_test(float x0, float y0, float x1, float y1, float k, const RayF &r, float t_min,
                                          float t_max, float &t_out) {
    constexpr int K = P == 0 ? 0 : (P == 1 ? 1 : 2);
    constexpr int A = P == 0 ? 1 : (P == 1 ? 2 : 0);
    constexpr int B = P == 0 ? 2 : (P == 1 ? 0 : 1);
    float t = (k - comp<K>(r.o)) * comp<K>(r.inv_d);
    const float x = comp<A>(r.o) + t * comp<A>(r.d);
    const float y = comp<B>(r.o) + t * comp<B>(r.d);
    const float m = fmaxf(fmaxf(fmaxf(t_min - t, t - t_max), fmaxf(x0 - x, x - x1)), fmaxf(y0 - y, y - y1));
    if (m > 0.0f) return false;
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
    float cl = t_max, t = 0.0f;
    // select form: every face is evaluated, the closest so far and its face number move by v_cndmask (no exec-mask
    // regions); the scan order and the shrinking interval are those of the list scan, so ties resolve the same way
    int f = -1;
    bool h;
    h = rect_test<2>(ax, ay, bx, by, bz, r, t_min, cl, t); cl = h ? t : cl; f = h ? 0 : f;
    h = rect_test<2>(ax, ay, bx, by, az, r, t_min, cl, t); cl = h ? t : cl; f = h ? 1 : f;
    h = rect_test<1>(az, ax, bz, bx, by, r, t_min, cl, t); cl = h ? t : cl; f = h ? 2 : f;
    h = rect_test<1>(az, ax, bz, bx, ay, r, t_min, cl, t); cl = h ? t : cl; f = h ? 3 : f;
    h = rect_test<0>(ay, az, by, bz, bx, r, t_min, cl, t); cl = h ? t : cl; f = h ? 4 : f;
    h = rect_test<0>(ay, az, by, bz, ax, r, t_min, cl, t); cl = h ? t : cl; f = h ? 5 : f;
    t_out = cl;
    face = f < 0 ? 0 : f;
    return f >= 0;
}

// Instanced primitive (rtmi.h, RTMI_PRIMFLAG_XF_*): the ray of the surrounding frame taken into the primitive's own
// frame through its chain — Traslate::hit / Rotate::hit (traslate.rs:18-20, rotate.rs:85-92) — with the per-frame
// derived values renewed when a rotation changed the direction.  t is the same in every frame.
__device__ __forceinline__ RayF prim_frame(const rtmi_xform *xf, uint32_t flags, const RayF &r) {
    RayF L = r;
    const int cnt = (int)((flags >> RTMI_PRIMFLAG_XF_COUNT_SHIFT) & RTMI_PRIM_XF_MAX);
    if (xform_ray(xf, (int)(flags >> RTMI_PRIMFLAG_XF_FIRST_SHIFT), cnt, L.o, L.d)) ray_derive(L);
    return L;
}
#define RTMI_PRIM_HAS_XF(flags) ((((flags) >> RTMI_PRIMFLAG_XF_COUNT_SHIFT) & RTMI_PRIM_XF_MAX) != 0u)

// one primitive against (t_min, t_max); pf = prim << 3 | face.  The three planes are loaded up
// front (independent addresses): one memory latency instead of up to three dependent ones.
// INST = false: an instantiation for scenes without instanced primitives (the cooperative kernel's headline build keeps
// the transform code out of its traversal loop: inlined there it costs registers whether it runs or not)
template <bool INST = true>
__device__ __forceinline__ bool prim_test(const DevScene &sc, int type, int idx, const RayF &r0, float time,
                                          float t_min, float t_max, float &t_out, int &pf) {
    // planes and meta from the primitive's leaf record (one base pointer for everything the trace phase reads of a
    // primitive: the separate plane arrays cost six more live scalar registers in a kernel that spills them)
    const PrimRec *pr = reinterpret_cast<const PrimRec *>(sc.leaf_rec + (size_t)idx * 5);
    const float4 A = pr->A;
    const float4 B = pr->B;
    const rtmi_prim_meta M = pr->M;
    RayF r = r0;
    if (INST && sc.has_prim_xf && RTMI_PRIM_HAS_XF(M.flags)) r = prim_frame(sc.xforms, M.flags, r0); // first test wave-uniform
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
// a primitive whose planes and meta words are at hand (leaf record of a gated tree)
template <bool INST = true>
__device__ __forceinline__ bool prim_test_vals(const DevScene &sc, int type, int idx, float4 A, float4 B, float m_inv_dt, uint32_t m_flags,
                                               const RayF &r0, float time, float t_min, float t_max, float &t_out, int &pf) {
    RayF r = r0;
    if (INST && sc.has_prim_xf && RTMI_PRIM_HAS_XF(m_flags)) r = prim_frame(sc.xforms, m_flags, r0);
    bool h = false;
    int face = 0;
    if (type <= RTMI_PRIM_MSPHERE) {
        // one sphere test for both kinds (one exec-mask region less, one copy of the test): the centre is selected,
        // not blended — a static sphere's centre is plane A as stored
        const F3 cm = moving_center(A, B, m_inv_dt, time);
        const bool mv = type == RTMI_PRIM_MSPHERE;
        h = sphere_test(r, f3(mv ? cm.x : A.x, mv ? cm.y : A.y, mv ? cm.z : A.z), A.w, t_min, t_max, t_out);
    } else if (type == RTMI_PRIM_RECT) {
        const int plane = (int)((m_flags >> RTMI_PRIMFLAG_PLANE_SHIFT) & 3u);
        h = rect_test_rt(plane, A, B.x, r, t_min, t_max, t_out);
    } else {
        h = cube_test(A, B, r, t_min, t_max, t_out, face);
    }
    pf = (idx << 3) | face;
    return h;
}
// the same for a wave-uniform primitive index (list items)
template <bool INST = true>
__device__ __forceinline__ bool prim_test_uniform(const DevScene &sc, int idx, const RayF &r0, float time,
                                                  float t_min, float t_max, float &t_out, int &pf) {
    const PrimRec PR = RTMI_UNIFORM_LOAD(PrimRec, reinterpret_cast<const PrimRec *>(sc.leaf_rec + (size_t)idx * 5));
    const float4 A = PR.A, B = PR.B;
    const rtmi_prim_meta M = PR.M;
    const int type = M.type;
    RayF r = r0;
    if (INST && RTMI_PRIM_HAS_XF(M.flags)) { // wave-uniform: the member of a nested list is wrapped in Traslate / Rotate
        const int cnt = (int)((M.flags >> RTMI_PRIMFLAG_XF_COUNT_SHIFT) & RTMI_PRIM_XF_MAX);
        if (xform_ray<true>(sc.xforms, (int)(M.flags >> RTMI_PRIMFLAG_XF_FIRST_SHIFT), cnt, r.o, r.d)) ray_derive(r);
    }
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
