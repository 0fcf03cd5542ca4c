// rtmi_bvh_block.hpp — workgroup-cooperative BVH traversal: the wavefronts of a workgroup share ONE work stack.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include "rtmi_bvh_coop.hpp"

// ----------------------------------------------------------------------------------
// Why (r03): in the wave-cooperative traversal (rtmi_bvh_coop.hpp) a pop round costs its wavefront the same
// instructions whether 60 or 6 of its lanes hold an entry, and on final_scene the rounds run at 53 % (ground BVH)
// and 28 % (sphere BVH) of their lanes: the tail of every call — a few long chains — is paid by a whole wavefront.
// The kernel is bound by instruction issue, and a wavefront that waits at a barrier issues nothing.  So here the
// RTMI_BLK_WAVES wavefronts of a workgroup put the entries of all their rays on one LIFO in LDS, and every round
// hands the pending entries out DENSELY: entry k of the round goes to thread k, wavefronts whose 64 threads got
// nothing skip the visit and wait at the barrier.  Four tails of 10 entries are one visit of one wavefront instead
// of four visits.
//   * no worker keeps a node for itself (that is what left the lanes thin): every surviving child of a visit is
//     published; the one exception is a LEAF that is the nearest surviving child, tested in the same visit;
//   * a ray's context (origin, direction, reciprocals, interval, d.d, pruning margin: 64 B) is read from LDS by
//     whoever pops an entry of it; the best hit is folded by the same LDS atomicMin on (t, inverted index) — the
//     fold is order-independent, so the result does not depend on who visits what, when;
//   * pool entry = {child reference (26 bits, encoding of rtmi_bvh_coop.hpp), entry distance with its low 8 mantissa
//     bits replaced by the ray id}; the distance is only a pruning bound and is decoded to a value <= the true one;
//   * two barriers per round: (A) the publications of the previous round are complete — every thread then derives
//     the same stack height from the same LDS counter, so all control flow around the barriers is workgroup-uniform;
//     (B) every thread holds its entry in registers — the popped range may be overwritten by this round's pushes.
//     Push positions come from one LDS atomicAdd per wavefront and round on one of two alternating counters.
// Alternative (gated 4-wide) trees only: the host selects this kernel when every BVH item has one.
// ----------------------------------------------------------------------------------
// (RTMI_BLK_WAVES / RTMI_BLK_THREADS / RTMI_BLK_LDS_WORDS: rtmi_kernels.hpp, next to the kernel declarations the launcher uses)

struct BlockWork {
    uint2 *pool;              // [cap] shared LIFO
    float4 *ctx;              // [4][T] ray contexts, one plane per float4 of the context
    unsigned long long *best; // [T] best hit of every ray of the running call
    uint32_t *sync;           // [0..1] push counters | [4..4+2*WAVES) two vote buffers
    int cap;
    int par;      // counter the next publication adds to          (workgroup-uniform, in registers)
    int vpar;     // vote buffer the next vote writes              (workgroup-uniform)
    uint32_t rot; // rotates which wavefront receives the first 64 entries of a round
};

__device__ __forceinline__ void block_work_init(BlockWork &bw, uint32_t *lds, int cap) {
    bw.pool = reinterpret_cast<uint2 *>(lds);
    bw.ctx = reinterpret_cast<float4 *>(lds + 2 * cap);
    bw.best = reinterpret_cast<unsigned long long *>(lds + 2 * cap + RTMI_BLK_THREADS * 16);
    bw.sync = lds + 2 * cap + RTMI_BLK_THREADS * 20;
    bw.cap = cap;
    bw.par = 0; bw.vpar = 0;
    bw.rot = blockIdx.x;
    if (threadIdx.x < 16) bw.sync[threadIdx.x] = 0u;
    __syncthreads();
}
// this wavefront's 64 dummy entries double as its private scratch outside the traversal (128 words)
__device__ __forceinline__ uint32_t *block_wave_scratch(const BlockWork &bw) {
    return reinterpret_cast<uint32_t *>(bw.best + RTMI_BLK_THREADS) + (threadIdx.x >> 6) * 128;
}

// Sum over the wavefronts of the workgroup of a per-wavefront word (all threads call; one barrier).  Two buffers in
// turn: a wavefront can write vote k + 2 only after every wavefront passed the barrier of vote k + 1, i.e. after
// every wavefront has read vote k.
__device__ __forceinline__ uint32_t block_vote(BlockWork &bw, uint32_t wave_word) {
    uint32_t *v = bw.sync + 4 + bw.vpar * RTMI_BLK_WAVES;
    bw.vpar ^= 1;
    if ((threadIdx.x & 63) == 0) v[threadIdx.x >> 6] = wave_word;
    __syncthreads();
    uint32_t s = 0u;
#pragma unroll
    for (int k = 0; k < RTMI_BLK_WAVES; k++) s += v[k];
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
}

// All threads of the workgroup call this together, the same number of times in the same order.
template <bool INST>
__device__ __forceinline__ void block_bvh_query(const DevScene &sc, int root, float scale, bool active, const RayF &R, float time,
                                                float q_min, float q_max, BlockWork &bw, bool &have, float &t_out, int &pf_out,
                                                bool &overflow) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint2 *pool = bw.pool;
    float4 *ctx = bw.ctx;
    unsigned long long *best = bw.best;
    uint32_t *cnt = bw.sync;
    const int cap = bw.cap;
    const int dummy = cap + RTMI_BLK_THREADS * 8 + RTMI_BLK_THREADS + tid; // in uint2 units from `pool`
    int p = bw.par;
    have = false;
    // ---- owners publish their ray context and the root entry
    best[tid] = COOP_SENTINEL;
    {
        const unsigned long long m_act = __ballot(active);
        const int n_act = __popcll(m_act);
        int base = 0;
        if (n_act != 0) { // wave-uniform
            if (lane == 0) base = (int)atomicAdd(&cnt[p], (uint32_t)n_act);
            base = __builtin_amdgcn_readfirstlane(base);
        }
        if (active) {
            const float wmabs = scale * (1.0f / 8192.0f) *
                                fminf(fminf(__builtin_fabsf(R.inv_d.x), __builtin_fabsf(R.inv_d.y)), __builtin_fabsf(R.inv_d.z));
            // four planes of T float4 each, not one 64-B record per ray: records at a 64-B stride put the 16-B reads of a
            // wavefront on 8 of the 32 banks (measured: six times the bank-conflict cycles of the wavefront kernel)
            ctx[0 * RTMI_BLK_THREADS + tid] = make_float4(R.o.x, R.o.y, R.o.z, time);
            ctx[1 * RTMI_BLK_THREADS + tid] = make_float4(R.d.x, R.d.y, R.d.z, q_min);
            ctx[2 * RTMI_BLK_THREADS + tid] = make_float4(R.inv_d.x, R.inv_d.y, R.inv_d.z, q_max);
            ctx[3 * RTMI_BLK_THREADS + tid] = make_float4(R.a, R.inv_a, wmabs, 0.0f);
            const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_act, 0u));
            // root entry distance: -FLT_MAX, never pruned (packed like every entry distance: ray id in the low 8 bits)
            pool[base + r] = make_uint2((uint32_t)root, 0xff7fff00u | (uint32_t)tid);
        }
    }
    int ebase = 0;
    for (;;) {
        __syncthreads(); // (A) the publications of the previous round are in the pool and counted
        const int height = ebase + __builtin_amdgcn_readfirstlane((int)cnt[p]);
        if (height == 0) break;
        int n_take = height < RTMI_BLK_THREADS ? height : RTMI_BLK_THREADS;
        if (height + 3 * RTMI_BLK_THREADS > cap) { // a visit replaces one entry by up to four: keep the pushes inside the pool
            const int room = (cap - height) / 3;
            if (n_take > room) n_take = room;
            if (n_take <= 0) { // cannot go on: reported loudly by the host, like an overflow of the wavefront pool
                overflow = true;
                __syncthreads();
                if (tid == 0) cnt[p] = 0u;
                __syncthreads();
                break;
            }
        }
        const int vt = ((((wave + (int)bw.rot) & (RTMI_BLK_WAVES - 1)) << 6) | lane);
        const bool work = vt < n_take;
        uint2 e = make_uint2(COOP_NONE, 0u);
        if (work) e = pool[height - 1 - vt];
        __syncthreads(); // (B) every entry of this round is in registers
        if (tid == 0) cnt[p] = 0u; // read by everybody before (B); the next additions to it come after (B) of the next round
        ebase = height - n_take;
        p ^= 1;
        bw.rot++;
        if (__ballot(work) == 0ull) continue; // this wavefront got nothing: it issues nothing until the next barrier

        uint32_t cur = work ? (e.x & 0x03ffffffu) : COOP_NONE;
        const int ray = (int)(e.y & 0xffu);
        // entry distance as a lower bound of the one that was published (low 8 bits of the mantissa carry the ray)
        const float tent = __uint_as_float((e.y & 0x80000000u) ? (e.y | 0xffu) : (e.y & 0xffffff00u));
        RayF W;
        const float4 c0 = ctx[0 * RTMI_BLK_THREADS + ray], c1 = ctx[1 * RTMI_BLK_THREADS + ray], c2 = ctx[2 * RTMI_BLK_THREADS + ray],
                     c3 = ctx[3 * RTMI_BLK_THREADS + ray];
        W.o = f3(c0.x, c0.y, c0.z); W.d = f3(c1.x, c1.y, c1.z); W.inv_d = f3(c2.x, c2.y, c2.z);
        W.a = c3.x; W.inv_a = c3.y;
        const float wtime = c0.w, wqmin = c1.w, wqmax = c2.w, wmabs = c3.z;
        const unsigned long long key = best[ray];
        const float bt = sort2f((uint32_t)(key >> 32));
        const float lim = bt + (__builtin_fabsf(bt) * (1.0f / 128.0f) + wmabs);
        const float limit = key != COOP_SENTINEL ? lim : RTMI_FLT_MAX;
        if (tent > limit) cur = COOP_NONE;

        uint32_t wkeep = 0u;
        uint32_t wch[4] = {0u, 0u, 0u, 0u};
        float wtn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!(cur & (1u << 25))) { // internal node of the 4-wide tree (COOP_NONE has the bit set)
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
            const float far0 = fminf(wqmax, limit); // same culling rule as the wavefront traversal (rtmi_bvh_coop.hpp, visit4)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float tn = fmaxf(fmaxf(fmaxf(wqmin, (anx[c] - W.o.x) * W.inv_d.x), (any_[c] - W.o.y) * W.inv_d.y), (anz[c] - W.o.z) * W.inv_d.z);
                const float tf = fminf(fminf(fminf(far0, (afx[c] - W.o.x) * W.inv_d.x), (afy[c] - W.o.y) * W.inv_d.y), (afz[c] - W.o.z) * W.inv_d.z);
                wtn[c] = tn;
                const bool okc = !(tn > tf) && wch[c] != COOP_NONE;
                if (okc) wkeep |= 1u << c;
                if (okc && !(tn >= tnear)) { nearest = c; tnear = tn; }
            }
            // the nearest surviving child, when it is a leaf, is tested right here (the leaf code runs for the threads
            // that popped a leaf anyway); everything else is published
            cur = COOP_NONE;
            if (nearest >= 0 && (wch[nearest] & (1u << 25))) { cur = wch[nearest]; wkeep &= ~(1u << nearest); }
        }
        if (cur != COOP_NONE) { // a leaf (popped, or just kept)
            const int type = (int)((cur >> 22) & 7u);
            const int idx = (int)(cur & 0x003fffffu);
            float t;
            int pf;
            const float4 *rec = sc.leaf_rec + (size_t)idx * 5;
            const float4 A = rec[0], B = rec[1], M = rec[2], g0 = rec[3], g1 = rec[4];
            bool hit = prim_test_vals<INST>(sc, type, idx, A, B, M.z, __float_as_uint(M.y), W, wtime, wqmin, wqmax, t, pf);
            // alternative tree: the reference reaches this leaf iff its parent's box passes
            if (hit) hit = aabb_hit(g0.x, g0.y, g0.z, g1.x, g1.y, g1.z, W, wqmin, wqmax);
            if (hit) {
                const unsigned long long k = ((unsigned long long)f2sort(t) << 32) | (unsigned long long)(0x7fffffffu - (uint32_t)pf);
                atomicMin(&best[ray], k);
            }
        }
        // ---- publish the marked children: 0..4 per thread
        const int npush = __popc(wkeep);
        const unsigned long long b0 = __ballot((npush & 1) != 0), b1 = __ballot((npush & 2) != 0), b2 = __ballot((npush & 4) != 0);
        const int total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
        if (total != 0) { // wave-uniform
            int base = 0;
            if (lane == 0) base = (int)atomicAdd(&cnt[p], (uint32_t)total);
            base = __builtin_amdgcn_readfirstlane(base);
            int at = ebase + base +
                     (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                     2 * (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
                     4 * (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
#pragma unroll
            for (int c = 0; c < 4; c++) { // unconditional stores: an unmarked child goes to the thread's dummy entry
                const bool on = (wkeep & (1u << c)) != 0u;
                pool[on ? at : dummy] = make_uint2(wch[c], (__float_as_uint(wtn[c]) & 0xffffff00u) | (uint32_t)ray);
                at += on ? 1 : 0;
            }
        }
    }
    bw.par = p ^ 1; // a wavefront still reading the last counter must not see the next call's additions: take the other one
    if (active) {
        const unsigned long long key = best[tid];
        if (key != COOP_SENTINEL) {
            have = true;
            t_out = sort2f((uint32_t)(key >> 32));
            pf_out = (int)(0x7fffffffu - (uint32_t)key);
        }
    }
}

// geometry of one item for the whole workgroup: every thread calls it; `active` threads own a query
template <bool INST>
__device__ __forceinline__ bool geom_query_block(const DevScene &sc, const rtmi_item &I, bool active, const RayF &r, float time,
                                                 float q_min, float q_max, BlockWork &bw, float &t_out, int &pf_out, bool &overflow) {
    if (I.kind == RTMI_ITEM_BVH) { // workgroup-uniform branch
        const bool enter = active && aabb_hit(I.root_min[0], I.root_min[1], I.root_min[2], I.root_max[0], I.root_max[1],
                                              I.root_max[2], r, q_min, q_max); // BVHNode::hit of the root: bvh.rs:71
        bool have = false;
        block_bvh_query<INST>(sc, I.alt_first, I.scale, enter, r, time, q_min, q_max, bw, have, t_out, pf_out, overflow);
        return have;
    }
    float cl = q_max; // HittableList::hit — hittable.rs:37-47
    bool any = false;
    if (active) {
        for (int k = 0; k < I.count; k++) {
            float t;
            int pf;
            if (prim_test_uniform<INST>(sc, I.first + k, r, time, q_min, cl, t, pf)) { cl = t; any = true; pf_out = pf; }
        }
    }
    t_out = cl;
    return any;
}
