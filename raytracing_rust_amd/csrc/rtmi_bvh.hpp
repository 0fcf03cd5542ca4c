// rtmi_bvh.hpp — per-lane BVH traversal (reference order, and fast-cull) and the per-lane hit query.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include "rtmi_geom.hpp"
#include "rtmi_prof.hpp"

// AABB::hit as above, additionally returning the entry distance max(t_min, near slabs).
__device__ __forceinline__ bool aabb_hit_t(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                           const RayF &r, float t_min, float t_max, float &t_enter) {
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
    const bool fail = t_max <= t_min; // one comparison after the last axis: see aabb_hit (rtmi_geom.hpp)
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
        const int type = reinterpret_cast<const PrimRec *>(sc.leaf_rec + (size_t)idx * 5)->M.type;
        float t;
        int pf;
        prof_tick<PROF>(prof, 13, true);            // list primitive tests
        if (prim_test(sc, type, idx, r, time, q_min, cl, t, pf)) { cl = t; any = true; pf_out = pf; }
    }
    t_out = cl;
    return any;
}
