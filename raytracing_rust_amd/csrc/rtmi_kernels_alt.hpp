// rtmi_kernels_alt.hpp — the two render kernels that are NOT the product's default path, kept as independent
// implementations of the same per-lane program for the parity tests (four schedules of one fold must give the same bits):
//   rtmi_render_bcoop  workgroup-cooperative traversal (RTMI_FLAG_BLOCK_COOP; measured -23 % against the default)
//   rtmi_render_async  per-lane state machine          (RTMI_FLAG_ASYNC;      measured 0.7x)
// They live in a translation unit of their own (rtmi_alt.hip), so the headline build neither compiles nor carries them;
// rtmi_device.hip launches them through the declarations at the end of rtmi_kernels.hpp.
#pragma once
#include "rtmi_kernels.hpp"
#include "rtmi_bvh_block.hpp"

// ----------------------------------------------------------------------------------
// render kernel, two-phase form with WORKGROUP-cooperative BVH traversal (rtmi_bvh_block.hpp).
// Same per-lane program as rtmi_render_coop (items in list order, media draws in order: same bits); what changes is
// who decides: the phase switches and the end of the kernel are votes of the whole workgroup, so that its wavefronts
// reach every traversal call together.  A wavefront whose lanes are all done keeps voting and keeps serving as
// workers in the shared traversal until the workgroup is done.
// ----------------------------------------------------------------------------------
template <bool SIG, bool INST>
__global__ __launch_bounds__(RTMI_BLK_THREADS, 4) void rtmi_render_bcoop(DevScene sc, DevCamera cam, DevParams P) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
    const int lane = threadIdx.x & 63;
    BlockWork bw;
    block_work_init(bw, lds_dyn, (int)P.coop_cap);
    float *scratch = reinterpret_cast<float *>(block_wave_scratch(bw));
    unsigned long long sig = 0ull;
    WaveWork w;
    w.ltile = 0u; w.ps_base = 0u; w.obase = 0u; w.x0 = 0u; w.y0 = 0u; w.cols = 0u; w.n_valid = 0u; w.next = 0u; w.total = 0u;
    bool queue_empty = false;
    const uint32_t k0 = P.key0, k1 = P.key1;
    const uint32_t threshold = P.shade_threshold * RTMI_BLK_WAVES;

    uint32_t oidx = 0u, ltile = 0u;
    bool alive = false, done = false, have_hit = false, overflow = false;
    RngReg g;
    rng_init(g, 0, 0);
    Path pa;
    pa.ro = f3(0, 0, 0); pa.rd = f3(0, 0, 1); pa.rtime = 0.0f; pa.T = f3(1, 1, 1); pa.L = f3(0, 0, 0); pa.depth = 0;
    float closest = RTMI_FLT_MAX;
    int best_item = -1, best_pf = 0;
    bool best_medium = false;

    for (;;) {
        // ================= phase A =================
        uint32_t n_hit;
        for (;;) {
            // one vote: lanes that still trace (low half) and lanes that hold a hit (high half)
            const uint32_t v = block_vote(bw, (uint32_t)__popcll(__ballot(!have_hit && !done)) | ((uint32_t)__popcll(__ballot(have_hit)) << 16));
            n_hit = v >> 16;
            if ((v & 0xffffu) == 0u || n_hit >= threshold) break;
            { // lanes whose path ended take the next (sample, pixel) item of the chunk (per wavefront, as in rtmi_render_coop)
                const bool want = !have_hit && !done && !alive;
                if (__ballot(want) != 0ull) {
                    uint32_t smp = 0u, px = 0u, j = 0u;
                    if (work_take(w, queue_empty, want, P, oidx, ltile, smp, px, j)) {
                        camera_sample(cam, P, g, k0, k1, smp, j * P.nx + px, px, j, pa);
                        alive = true;
                    } else if (want) {
                        done = true;
                    }
                }
            }
            const bool need = !have_hit && !done;
            RayF W;
            W.o = pa.ro; W.d = pa.rd;
            ray_derive(W);
            if (need) { closest = RTMI_FLT_MAX; best_item = -1; best_pf = 0; best_medium = false; }
            for (uint32_t it = 0; it < sc.n_items; it++) { // executed by all threads of the workgroup
                const rtmi_item I = RTMI_UNIFORM_LOAD(rtmi_item, &sc.items[it].it);
                RayF R = W;
                if (I.xform_count > 0) {
                    struct XPair { rtmi_xform x0, x1; };
                    const XPair XP = RTMI_UNIFORM_LOAD(XPair, reinterpret_cast<const XPair *>(&sc.items[it].x0));
                    if (xform_ray_item<true>(sc.xforms, I.xform_first, I.xform_count, XP.x0, XP.x1, R.o, R.d)) ray_derive(R);
                }
                if (!(I.flags & RTMI_ITEMFLAG_MEDIUM)) {
                    float t;
                    int pf;
                    if (geom_query_block<INST>(sc, I, need, R, pa.rtime, P.t_min, closest, bw, t, pf, overflow)) {
                        closest = t; best_item = (int)it; best_pf = pf; best_medium = false;
                    }
                } else { // ConstantMedium::hit — medium.rs:28-56
                    float t1 = 0.0f, t2 = 0.0f, tm;
                    int pf;
                    bool h1, h2;
                    if (I.flags & RTMI_ITEMFLAG_DEV_MEDIUM_SPHERE) {
                        h1 = false; h2 = false;
                        if (need) sphere_two_queries(R, make_float4(I.root_min[0], I.root_min[1], I.root_min[2], I.root_max[0]), h1, t1, h2, t2);
                    } else {
                        h1 = geom_query_block<INST>(sc, I, need, R, pa.rtime, -RTMI_FLT_MAX, RTMI_FLT_MAX, bw, t1, pf, overflow);
                        h2 = geom_query_block<INST>(sc, I, need && h1, R, pa.rtime, t1 + 0.0001f, RTMI_FLT_MAX, bw, t2, pf, overflow);
                    }
                    if (need && h1 && h2) {
                        if (medium_sample(t1, t2, P.t_min, closest, medium_dir_norm<INST>(sc, I.flags, I.xform_first, W), I.neg_inv_density, g, k0, k1, tm)) {
                            closest = tm; best_item = (int)it; best_medium = true;
                        }
                    }
                }
            }
            if (need) {
                if (best_item >= 0) {
                    have_hit = true;
                } else { // miss: black background (color.rs:21)
                    if (P.sky) pa.L = pa.L + pa.T * sky_color(pa.rd);
                    path_end(P, oidx, pa);
                    if (SIG) { atomicAdd(P.path_sig + (size_t)ltile * 64 + (oidx & 63u), sig); sig = 0ull; }
                    alive = false;
                }
            }
        }
        // ================= phase B =================
        if (n_hit == 0u) break; // nobody traces, nobody holds a hit: the workgroup is done
        {
            const bool shading = have_hit;
            have_hit = false;
            if (SIG && shading) sig += (unsigned long long)sig_mix(__float_as_uint(closest), pa.depth);
            const bool goes_on = shade_hit<RngReg, INST>(sc, P.max_depth, P.ext, g, k0, k1, shading, closest, best_item, best_pf, best_medium, pa, scratch);
            if (shading && !goes_on) {
                path_end(P, oidx, pa);
                if (SIG) { atomicAdd(P.path_sig + (size_t)ltile * 64 + (oidx & 63u), sig); sig = 0ull; }
                alive = false;
            }
        }
    }
    if (P.ext & RTMI_EXT_TEST_OVERFLOW) overflow = true; // test knob: exercise the error path
    if (__ballot(overflow) != 0ull && lane == 0) atomicAdd(P.status, 1u); // reported loudly by the host
}

// ----------------------------------------------------------------------------------
// render kernel, asynchronous form (RTMI_FLAG_ASYNC; a measured negative result kept as an independent
// implementation for the parity tests: bit-identical, ~0.7x the two-phase kernels).
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
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void rtmi_render_async(DevScene sc, DevCamera cam, DevParams P) {
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
    const WaveWork w = wave_work(P, blockIdx.x * WAVES_PER_BLOCK + (uint32_t)wave); // one unit per wavefront, no queue
    if (!PROF && w.total == 0u) return;
    const uint32_t k0 = P.key0, k1 = P.key1;
    const int n_items = (int)sc.n_items;

    // static assignment lane = pixel (this kernel predates the dynamic hand-out of the two-phase kernels)
    const bool in_image = w.total != 0u && (uint32_t)(lane & 7) < w.cols && (uint32_t)(lane >> 3) * w.cols < w.n_valid;
    const uint32_t s_begin = w.ps_base >> 6;
    const uint32_t s_end = in_image ? s_begin + w.total / w.n_valid : s_begin;
    const uint32_t px = w.x0 + (uint32_t)(lane & 7), j = P.ny - 1u - (w.y0 + (uint32_t)(lane >> 3));
    uint32_t s = s_begin;
    RngReg g;
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
                                if (medium_sample(t1, bt, P.t_min, closest, medium_dir_norm(sc, iflags, sc.items[it].it.xform_first, W), sc.items[it].it.neg_inv_density, g, k0, k1, tm)) {
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
                            if (P.sky) pa.L = pa.L + pa.T * sky_color(pa.rd);
                            path_end(P, w.obase + ((s << 6) | (uint32_t)lane), pa);
                            s++; st = ST_NEW;
                        }
                        break;
                    }
                    // ---- enter item `it`
                    const rtmi_item I = sc.items[it].it;
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
            const bool shading = st == ST_SHADE;
            if (SIG && shading) sig += (unsigned long long)sig_mix(__float_as_uint(closest), pa.depth);
            // all lanes call (wavefront texture lookup).  Scratch = entry 0 of the traversal stacks of all lanes: the
            // lanes in ST_NODE / ST_PRIM hold live entries there, so the 64 words are saved around the call.
            const uint32_t saved0 = stack[0];
            const bool goes_on = shade_hit(sc, P.max_depth, P.ext, g, k0, k1, shading, closest, best_item, best_pf, best_medium, pa,
                                           reinterpret_cast<float *>(stack - lane));
            stack[0] = saved0;
            if (shading) {
                if (goes_on) {
                    W.o = pa.ro; W.d = pa.rd;
                    ray_derive(W);
                    it = 0; ph = 0; pending = false; closest = RTMI_FLT_MAX; best_item = -1; best_medium = false;
                    st = ST_ITEM;
                } else {
                    path_end(P, w.obase + ((s << 6) | (uint32_t)lane), pa);
                    s++; st = ST_NEW;
                }
            }
        } else { // ST_NEW
            if (st == ST_NEW) {
                if (s >= s_end) { st = ST_DONE; }
                else {
                    camera_sample(cam, P, g, k0, k1, s, j * P.nx + px, px, j, pa);
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
    }
    if (SIG && in_image) atomicAdd(P.path_sig + (size_t)w.ltile * 64 + lane, sig); // integer add: order-independent
}

