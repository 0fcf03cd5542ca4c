// rtmi_bvh_coop.hpp — wave-cooperative BVH traversal over a shared LDS work pool.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include "rtmi_bvh.hpp"

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
#define RTMI_COOP_DUMMY_WORDS 128u /* one uint2 per lane behind `best`: target of the stores of children that are not published */
#define COOP_SENTINEL 0xffffffffffffffffull
__device__ __forceinline__ uint32_t f2sort(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sort2f(uint32_t s) {
    return __uint_as_float((s & 0x80000000u) ? (s & 0x7fffffffu) : ~s);
}
// Pool encoding of a node child reference (26 bits, the ray id takes the upper 6): an internal node is its index
// (< 2^25); a leaf is 1<<25 | type<<22 | primitive (< 2^22).  rtmi_scene_create stores both children of every
// node in this encoding in the node record's reserved words.

// Work space of one wavefront.  LDS (uint32 words): pool [cap][2] | ctx [64][12] floats | best [64] uint64 | dummy [64][2].
// The pool in LDS is the TOP of the logical LIFO; when it runs full its older half moves to `spill` (global
// memory, private to the wavefront) and comes back when the LDS part is empty, so the logical stack and
// its depth-first bound of 64 * (tree depth + 1) entries are unchanged while the LDS footprint does not
// depend on the tree depth.
struct CoopWork {
    uint32_t *wlds;
    int cap;        // LDS pool entries (multiple of 64, >= 256)
    uint2 *spill;   // global part of the stack
    int spill_cap;  // its capacity in entries: 64 * (deepest tree + 2)
};

// All 64 lanes must call this together.  EXT = false is the lean instantiation for scenes that need neither
// alternative trees nor a stack beyond LDS (pool = the full bound 64 * (depth + 2), no gate, no spill code: the
// extra code costs 2-3 % everywhere through register allocation); EXT = true has both.
// W4: `root` is a node of the 4-wide alternative tree (sc.nodes4, always gated); a visit tests four child boxes,
// keeps the nearest surviving child and pushes up to three.  Half the chain length of a binary tree: the
// traversals are bound by their longest chain, not by their work.
#define RTMI_COOP_INLINE __forceinline__
template <bool PROF, bool EXT, bool W4, bool INST>
__device__ __forceinline__ void coop_bvh_query(const DevScene &sc, int root, bool gated, float scale, bool active, const RayF &R,
                                               float time, float q_min, float q_max, const CoopWork &cw,
                                               bool &have, float &t_out, int &pf_out, bool &overflow,
                                               unsigned long long *prof, int slot) {
    const int lane = threadIdx.x & 63;
    uint32_t *wlds = cw.wlds;
    const int cap = cw.cap;
    // Plain (non-volatile) LDS accesses: every exchange between lanes is separated by a wavefront-scope
    // fence + wave barrier, and plain accesses let the compiler keep the LDS address space (ds_read/write_b64;
    // volatile ones became flat loads/stores with a full wait each — five per iteration).
    uint2 *pool = reinterpret_cast<uint2 *>(wlds);
    float4 *ctx = reinterpret_cast<float4 *>(wlds + 2 * cap);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(wlds + 2 * cap + 64 * 12);

    const unsigned long long m_act = __ballot(active);
    have = false;
    if (m_act == 0ull) return; // wave-uniform: nobody enters this tree
    // ---- publish ray contexts; every owner starts as the worker of its own root entry (no pool round trip, its ray
    // context is already in registers)
    best[lane] = COOP_SENTINEL;
    uint32_t cur = COOP_NONE; // 26-bit node/leaf encoding of the entry this worker holds
    int ray = 0, cray = -1, aray = -1; // ray of the held entry; ray whose context / whose a, inv_a are loaded
    float tent = 0.0f;
    RayF W;                   // context of ray `cray`
    W.o = f3(0, 0, 0); W.d = f3(0, 0, 1); W.inv_d = f3(0, 0, 0); W.a = 1.0f; W.inv_a = 1.0f;
    float wtime = 0.0f, wqmin = 0.0f, wqmax = 0.0f, wmabs = 0.0f;
    if (active) {
        // 48 B per ray keeps the wave's LDS under 10 KB; a and inv_a are recomputed by a worker when it
        // first tests a sphere of this ray (same operations on the same values: bit-identical)
        ctx[lane * 3 + 0] = make_float4(R.o.x, R.o.y, R.o.z, time);
        ctx[lane * 3 + 1] = make_float4(R.d.x, R.d.y, R.d.z, q_min);
        ctx[lane * 3 + 2] = make_float4(R.inv_d.x, R.inv_d.y, R.inv_d.z, q_max);
        cur = (uint32_t)root; ray = lane; cray = lane;
        tent = q_min; // root entry distance: conservative
        W.o = R.o; W.d = R.d; W.inv_d = R.inv_d;
        wtime = time; wqmin = q_min; wqmax = q_max;
        wmabs = scale * (1.0f / 8192.0f) *
                fminf(fminf(__builtin_fabsf(W.inv_d.x), __builtin_fabsf(W.inv_d.y)), __builtin_fabsf(W.inv_d.z));
    }
    int top = 0, gtop = 0; // entries in the LDS part / in the spilled (older) part of the stack
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // children of the 4-wide node visited this round and their entry distances; only read under a bit of `wkeep`, so
    // they live outside the loop and are not re-initialised every round (16 moves per round otherwise)
    uint32_t wch[4] = {0u, 0u, 0u, 0u};
    float wtn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    // One visit of a node of the 4-wide tree held in `cur` (ray context W): tests the four child boxes, keeps the
    // nearest surviving child in `cur` / `tent`, marks the others in `wkeep` for publication.
    // These boxes only cull (acceptance is the primitive's own test and its gate) and are padded beyond every
    // primitive, so the slab test need not follow the reference's operation order.
    const auto visit4 = [&](float limit, uint32_t &wkeep, int &npush) {
        // near / far plane of every slab picked by the sign of the ray direction (an index into the record, no
        // min/max pair per axis and box; measured final_scene +1.7 %, random_spheres +2.7 %).  A NaN from 0 * inf on a plane
        // the ray lies in is ignored by fmaxf / fminf: no constraint from that plane, which only culls less.
        const uint32_t base = cur * 8u; // minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4] child[4] -
        const uint32_t sx = W.inv_d.x < 0.0f ? 3u : 0u, sy = W.inv_d.y < 0.0f ? 3u : 0u, sz = W.inv_d.z < 0.0f ? 3u : 0u;
        const float4 nx4 = sc.nodes4[base + sx], fx4 = sc.nodes4[base + 3u - sx];
        const float4 ny4 = sc.nodes4[base + 1u + sy], fy4 = sc.nodes4[base + 4u - sy];
        const float4 nz4 = sc.nodes4[base + 2u + sz], fz4 = sc.nodes4[base + 5u - sz];
        const float4 chf = sc.nodes4[base + 6u];
        const float anx[4] = {nx4.x, nx4.y, nx4.z, nx4.w}, afx[4] = {fx4.x, fx4.y, fx4.z, fx4.w};
        const float any_[4] = {ny4.x, ny4.y, ny4.z, ny4.w}, afy[4] = {fy4.x, fy4.y, fy4.z, fy4.w};
        const float anz[4] = {nz4.x, nz4.y, nz4.z, nz4.w}, afz[4] = {fz4.x, fz4.y, fz4.z, fz4.w};
        wch[0] = __float_as_uint(chf.x); wch[1] = __float_as_uint(chf.y); wch[2] = __float_as_uint(chf.z); wch[3] = __float_as_uint(chf.w);
        int nearest = -1;
        float tnear = RTMI_FLT_MAX;
        // the pruning limit joins the far side of the interval: "tn < tf and tn <= limit" becomes "tn <= min(tf, limit)",
        // which also lets a zero-thickness overlap through — a superset, and these boxes only cull (+0.9 % / +1.2 %)
        const float far0 = fminf(wqmax, limit);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float tn = fmaxf(fmaxf(fmaxf(wqmin, (anx[c] - W.o.x) * W.inv_d.x), (any_[c] - W.o.y) * W.inv_d.y), (anz[c] - W.o.z) * W.inv_d.z);
            const float tf = fminf(fminf(fminf(far0, (afx[c] - W.o.x) * W.inv_d.x), (afy[c] - W.o.y) * W.inv_d.y), (afz[c] - W.o.z) * W.inv_d.z);
            wtn[c] = tn;
            // (the explicit test of the child word stays: a ray with NaN components passes every min/max slab
            // test, and an empty slot's reference must never reach the pool)
            const bool okc = !(tn > tf) && wch[c] != COOP_NONE;
            if (okc) wkeep |= 1u << c;
            // (!(tn >= tnear), not tn < tnear: with a NaN entry distance — a NaN q_min — some surviving child still
            // becomes the nearest, so at most three are ever left to publish)
            if (okc && !(tn >= tnear)) { nearest = c; tnear = tn; }
        }
        if (nearest >= 0) { cur = wch[nearest]; tent = tnear; wkeep &= ~(1u << nearest); } else cur = COOP_NONE;
        npush = __popc(wkeep); // what is left gets published
    };
    // Publication of the marked children of every worker's visit (all 64 lanes call): exclusive prefix sum of the
    // per-worker counts (0..3) from two ballots.
    const auto publish4 = [&](uint32_t wkeep, int npush) {
        const unsigned long long b0 = __ballot((npush & 1) != 0), b1 = __ballot((npush & 2) != 0);
        const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                           2 * (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
        int at = top + before;
        // Unconditional stores: a child that is not published goes to the worker's own dummy entry behind `best` — a
        // v_cndmask on the address instead of an exec-mask region (three scalar instructions) per child.
        const int dummy = cap + 64 * 6 + 64 + lane; // in uint2 units from `pool`: behind ctx (64 x 48 B) and best (64 x 8 B)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const bool on = (wkeep & (1u << c)) != 0u;
            pool[on ? at : dummy] = make_uint2(((uint32_t)ray << 26) | wch[c], __float_as_uint(wtn[c]));
            at += on ? 1 : 0;
        }
        top += __popcll(b0) + 2 * __popcll(b1);
    };
    // The hot loop runs until the LDS part is empty with all workers idle, or too full for 64 more pushes; the
    // rare handling of both (refill from / spill to global memory) sits in the outer loop, outside the hot
    // loop's register allocation.
    for (;;) {
    for (;;) {
        // ---- idle workers take the deepest pending entries
        const bool needw = cur == COOP_NONE;
        const unsigned long long m_need = __ballot(needw);
        const int n_need = __popcll(m_need);
        if (top == 0 && n_need == 64) break;
        if (top > cap - (W4 ? 192 : 64)) { // no room for this round's pushes (one per worker, three in a 4-wide tree)
            if (!EXT) overflow = true; // cannot happen: the LDS pool is the depth-first bound itself
            break;
        }
        const int take = n_need < top ? n_need : top;
        {
            const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_need, 0u));
            if (needw && r < take) { // one exec-mask region (scalar instructions are the dear ones here: rtmi_geom.hpp)
                const int idx = top - 1 - r;
                const uint2 e = pool[idx];
                tent = __uint_as_float(e.y);
                cur = e.x & 0x03ffffffu;
                ray = (int)(e.x >> 26);
            }
        }
        top -= take;
        prof_tick<PROF>(prof, slot, cur != COOP_NONE);
        bool push = false;
        uint32_t push_ref = 0u;
        float push_t = 0.0f;
        int npush = 0;                    // W4: up to three of the four children are published
        uint32_t wkeep = 0u;              // bit c: child c is published (wch / wtn: declared outside the loop)
        {
            if (cur != COOP_NONE && ray != cray) { // switch ray context
                const float4 c0 = ctx[ray * 3 + 0], c1 = ctx[ray * 3 + 1], c2 = ctx[ray * 3 + 2];
                W.o = f3(c0.x, c0.y, c0.z); wtime = c0.w;
                W.d = f3(c1.x, c1.y, c1.z); wqmin = c1.w;
                W.inv_d = f3(c2.x, c2.y, c2.z); wqmax = c2.w;
                // pruning margin scale/8192/|d|, over-estimated (never under: pruning less is always safe) by
                // 1/|d| <= min_i 1/|d_i| — no division, no square root on a context switch
                wmabs = scale * (1.0f / 8192.0f) *
                        fminf(fminf(__builtin_fabsf(W.inv_d.x), __builtin_fabsf(W.inv_d.y)), __builtin_fabsf(W.inv_d.z));
                cray = ray;
            }
            // the ray's best hit so far -> pruning limit (idle workers read the slot of the ray they held last: harmless)
            const unsigned long long key = best[ray];
            // select form (no exec-mask region): the sentinel decodes to a NaN pattern whose "limit" is discarded
            const float bt = sort2f((uint32_t)(key >> 32));
            const float lim = bt + (__builtin_fabsf(bt) * (1.0f / 128.0f) + wmabs);
            const float limit = key != COOP_SENTINEL ? lim : RTMI_FLT_MAX;
            if (tent > limit) cur = COOP_NONE; // (an idle worker's cur is COOP_NONE already: all its bits are set)
            if (W4 && !(cur & (1u << 25))) { // internal node of the 4-wide tree
                visit4(limit, wkeep, npush);
            } else if (!W4 && !(cur & (1u << 25))) { // internal node
                const float4 *n = sc.nodes + (size_t)cur * 4;
                const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
                // child references in pool encoding (filled in by rtmi_scene_create from left/right = n3.x, n3.y)
                const uint32_t left = __float_as_uint(n3.z), right = __float_as_uint(n3.w);
                float tl, tr;
                bool vl = aabb_hit_t(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, W, wqmin, wqmax, tl);
                bool vr = aabb_hit_t(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, W, wqmin, wqmax, tr);
                vl = vl && !(tl > limit);
                vr = vr && !(tr > limit) && right != left;
                if (vl && vr) {
                    const bool lfirst = !(tr < tl);
                    push = true;
                    push_ref = lfirst ? right : left;
                    push_t = lfirst ? tr : tl;
                    cur = lfirst ? left : right;
                    tent = lfirst ? tl : tr;
                } else if (vl) { cur = left; tent = tl; }
                else if (vr) { cur = right; tent = tr; }
                else cur = COOP_NONE;
            }
        }
        // Leaf part, after the node part instead of beside it: a worker whose visit just kept a LEAF as its nearest
        // child tests it in this same round (the leaf code runs for the workers that popped a leaf anyway), which
        // takes one pop round off every chain (measured: final_scene +3.3 %, random_spheres +8 %).
        if (cur != COOP_NONE && (cur & (1u << 25))) {
            {
                const int type = (int)((cur >> 22) & 7u);
                const int idx = (int)(cur & 0x003fffffu);
                float t;
                int pf;
                if (type <= RTMI_PRIM_MSPHERE && aray != ray) { // sphere.rs:40 needs d.d
                    W.a = dot(W.d, W.d);
                    W.inv_a = 1.0f / W.a;
                    aray = ray;
                }
                // the gate box is fetched together with the primitive's planes, not after its test: one memory latency
                // per leaf instead of two on the path of every accepted hit
                bool hit;
                if (EXT && gated) { // one 80-B record: planes, meta and the gate box arrive together
                    const float4 *rec = sc.leaf_rec + (size_t)idx * 5;
                    const float4 A = rec[0], B = rec[1], M = rec[2], g0 = rec[3], g1 = rec[4];
                    hit = prim_test_vals<INST>(sc, type, idx, A, B, M.z, __float_as_uint(M.y), W, wtime, wqmin, wqmax, t, pf);
                    // alternative tree: the reference reaches this leaf iff its parent's box passes
                    if (hit) hit = aabb_hit(g0.x, g0.y, g0.z, g1.x, g1.y, g1.z, W, wqmin, wqmax);
                } else {
                    hit = prim_test<INST>(sc, type, idx, W, wtime, wqmin, wqmax, t, pf);
                }
                if (hit) {
                    const unsigned long long k = ((unsigned long long)f2sort(t) << 32) | (unsigned long long)(0x7fffffffu - (uint32_t)pf);
                    atomicMin(&best[ray], k);
                }
                cur = COOP_NONE;
            }
        }
        // ---- publish the far children
        if (W4) {
            if (__ballot(npush != 0) != 0ull) publish4(wkeep, npush); // leaf-only rounds publish nothing
        } else {
        const unsigned long long m_push = __ballot(push);
        if (push) {
            const int pos = top + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_push >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_push, 0u));
            pool[pos] = make_uint2(((uint32_t)ray << 26) | push_ref, __float_as_uint(push_t));
        }
        top += __popcll(m_push);
        }
        if (PROF && lane == 0) atomicMax(&prof[2 * 18], (unsigned long long)top); // deepest pool seen (diagnostics)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
        if (!EXT) break;
        if (top > cap - (W4 ? 192 : 64)) { // no room for the pushes: the older half goes to global memory, order kept
            const int m = top - 64; // > 0 (top > cap - 192 >= 64): keep the newest 64, so the hot loop can go on
            if (gtop + m > cw.spill_cap) { overflow = true; break; } // cannot happen by the depth-first bound
            for (int i = lane; i < m; i += 64) cw.spill[gtop + i] = pool[i];
            const int rest = top - m;
            for (int base = 0; base < rest; base += 64) { // slide the newer part down; chunk k writes [64k, 64k+64)
                const int i = base + lane;                // and later chunks read above m + 64k + 64: no overlap
                uint2 e = make_uint2(0u, 0u);
                if (i < rest) e = pool[m + i];
                __builtin_amdgcn_wave_barrier();
                if (i < rest) pool[i] = e;
            }
            gtop += m;
            top = rest;
        } else if (gtop > 0) { // LDS part empty, every worker idle: bring back the newest spilled entries
            const int n = gtop < cap / 2 ? gtop : cap / 2;
            for (int i = lane; i < n; i += 64) pool[i] = cw.spill[gtop - n + i];
            gtop -= n;
            top = n;
        } else {
            break; // nothing pending anywhere
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (active) {
        const unsigned long long key = best[lane];
        if (key != COOP_SENTINEL) {
            have = true;
            t_out = sort2f((uint32_t)(key >> 32));
            pf_out = (int)(0x7fffffffu - (uint32_t)key);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// geometry of one item for the whole wavefront: every lane calls it; `active` lanes own a query
template <bool PROF, bool EXT, bool W4, bool INST>
__device__ __forceinline__ bool geom_query_coop(const DevScene &sc, const rtmi_item &I, bool use_alt, bool active, const RayF &r,
                                                float time, float q_min, float q_max, const CoopWork &cw,
                                                float &t_out, int &pf_out, bool &overflow, unsigned long long *prof,
                                                int slot) {
    if (I.kind == RTMI_ITEM_BVH) { // wave-uniform branch
        // BVHNode::hit of the root: its own bbox first (bvh.rs:71)
        const bool enter = active && aabb_hit(I.root_min[0], I.root_min[1], I.root_min[2], I.root_max[0], I.root_max[1],
                                              I.root_max[2], r, q_min, q_max);
        bool have = false;
        const bool alt = EXT && W4 && use_alt && I.alt_first >= 0; // wave-uniform
        if (alt) coop_bvh_query<PROF, EXT, W4, INST>(sc, I.alt_first, true, I.scale, enter, r, time, q_min, q_max, cw, have, t_out, pf_out, overflow, prof, slot);
        else coop_bvh_query<PROF, EXT, false, INST>(sc, I.first, false, I.scale, enter, r, time, q_min, q_max, cw, have, t_out, pf_out, overflow, prof, slot);
        return have;
    }
    // HittableList::hit — hittable.rs:37-47
    float cl = q_max;
    bool any = false;
    if (active) {
        for (int k = 0; k < I.count; k++) {
            const int idx = I.first + k; // wave-uniform
            float t;
            int pf;
            prof_tick<PROF>(prof, 13, true);
            if (prim_test_uniform<INST>(sc, idx, r, time, q_min, cl, t, pf)) { cl = t; any = true; pf_out = pf; }
        }
    }
    t_out = cl;
    return any;
}
