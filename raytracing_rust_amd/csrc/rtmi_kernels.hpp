// rtmi_kernels.hpp — render kernels (per-lane two-phase, wave-cooperative, async state machine), resolve, probes.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include "rtmi_bvh_coop.hpp"
#include "rtmi_shade.hpp"

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
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void rtmi_render_kernel(DevScene sc, DevCamera cam, DevParams P) {
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
    WaveWork w;
    w.ltile = 0u; w.ps_base = 0u; w.obase = 0u; w.x0 = 0u; w.y0 = 0u; w.cols = 0u; w.n_valid = 0u; w.next = 0u; w.total = 0u;
    bool queue_empty = false;
    const uint32_t k0 = P.key0, k1 = P.key1;
    const int threshold = (int)P.shade_threshold;

    uint32_t oidx = 0u, ltile = 0u; // slot of this lane's path in the per-sample buffer; its local tile (SIG only)
    bool alive = false, done = false, have_hit = false;
    RngReg g;
    rng_init(g, 0, 0);
    Path pa;
    pa.ro = f3(0, 0, 0); pa.rd = f3(0, 0, 1); pa.rtime = 0.0f; pa.T = f3(1, 1, 1); pa.L = f3(0, 0, 0); pa.depth = 0;
    float closest = RTMI_FLT_MAX;
    int best_item = -1, best_pf = 0;
    bool best_medium = false;

    for (;;) {
        // ================= phase A: trace until enough lanes hold a hit =================
        for (;;) {
            if (__ballot(!have_hit && !done) == 0ull) break;
            { // lanes whose path ended take the next (sample, pixel) item of the chunk
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
            prof_tick<PROF>(prof, 0, need);
            if (need) {
                // ---- world.hit(ray, 0.001, f64::MAX): scan of the top-level list (hittable.rs:37-47)
                RayF W;
                W.o = pa.ro; W.d = pa.rd;
                ray_derive(W);
                closest = RTMI_FLT_MAX;
                best_item = -1; best_pf = 0; best_medium = false;
                float t0_saved = RTMI_FLT_MAX; // the closest hit before a BVH item whose media / instanced-subtree children follow as DEFERRED items
                int grp_first = 0x7fffffff;    // ... the index of that item (or of the first deferred one), and whether it is the enclosing tree
                bool grp_tree = false;
                ListScan ls; // a list with media that was a child of a BVHNode (rtmi.h, LISTSCAN)
                ls.cl = RTMI_FLT_MAX; ls.item = -1; ls.pf = 0; ls.medium = false; ls.has = false;
                for (uint32_t it = 0; it < sc.n_items; it++) {
                    const rtmi_item I = sc.items[it].it;
                    if (I.flags & RTMI_ITEMFLAG_SAVE_T0) {
                        t0_saved = closest; grp_first = (int)it; grp_tree = I.kind == RTMI_ITEM_BVH && !(I.flags & RTMI_ITEMFLAG_DEFERRED);
                    }
                    if (I.flags & RTMI_ITEMFLAG_LISTSCAN_END) { // the terminator: the scan's result meets the closest hit so far
                        listscan_fold(ls, I.first, closest, best_item, best_pf, best_medium, grp_first, grp_tree);
                        continue;
                    }
                    if (I.flags & RTMI_ITEMFLAG_LISTSCAN_BEGIN) { ls.cl = t0_saved; ls.has = false; }
                    const bool scan = (I.flags & RTMI_ITEMFLAG_LISTSCAN_MEMBER) != 0u;
                    RayF R = W;
                    if (I.xform_count > 0) {
                        if (xform_ray(sc.xforms, I.xform_first, I.xform_count, R.o, R.d)) ray_derive(R);
                    }
                    const int slot = 1 + (it < 11u ? (int)it : 11);
                    if (!(I.flags & RTMI_ITEMFLAG_MEDIUM)) {
                        float t;
                        int pf;
                        if (scan) { // a primitive member of the list scan: t_max = the scan's closest hit so far
                            if (deferred_gate(sc, I, W, P.t_min, t0_saved) &&
                                geom_query<FAST, PROF>(sc, I, R, pa.rtime, P.t_min, ls.cl, stack, t, pf, prof, slot)) {
                                ls.cl = t; ls.item = (int)it; ls.pf = pf; ls.medium = false; ls.has = true;
                            }
                        } else if (I.flags & RTMI_ITEMFLAG_DEFERRED) { // an instanced subtree that was a child of a BVHNode (rtmi.h)
                            if (deferred_gate(sc, I, W, P.t_min, t0_saved) &&
                                geom_query<FAST, PROF>(sc, I, R, pa.rtime, P.t_min, t0_saved, stack, t, pf, prof, slot) &&
                                deferred_bvh_wins(I.count, t, closest, best_item, best_pf, grp_first, grp_tree)) {
                                closest = t; best_item = (int)it; best_pf = pf; best_medium = false;
                            }
                        } else if (geom_query<FAST, PROF>(sc, I, R, pa.rtime, P.t_min, closest, stack, t, pf, prof, slot)) {
                            closest = t; best_item = (int)it; best_pf = pf; best_medium = false;
                        }
                    } else {
                        // ConstantMedium::hit — medium.rs:28-56
                        float t1, t2, tm;
                        int pf;
                        // a medium that was a child of a BVHNode (rtmi.h, DEFERRED): reached through its parent's box, its
                        // interval clamped to the t_max the BVH was entered with, accepted when closer than the tree's hit
                        const bool dfr = (I.flags & RTMI_ITEMFLAG_DEFERRED) != 0u;
                        const float qmax = scan ? ls.cl : (dfr ? t0_saved : closest);
                        if (!dfr || deferred_gate(sc, I, W, P.t_min, t0_saved)) {
                        if (geom_query<FAST, PROF>(sc, I, R, pa.rtime, -RTMI_FLT_MAX, RTMI_FLT_MAX, stack, t1, pf, prof, slot)) {
                            if (geom_query<FAST, PROF>(sc, I, R, pa.rtime, t1 + 0.0001f, RTMI_FLT_MAX, stack, t2, pf, prof, slot)) {
                                const float dn = medium_dir_norm(sc, I.flags, I.xform_first, W);
                                if ((I.flags & RTMI_ITEMFLAG_NESTED_MEDIUM) && !nested_medium_interval(sc, I, dn, g, k0, k1, t1, t2)) {
                                    // the inner medium returned no hit to one of the outer medium's two queries
                                } else
                                if (medium_sample(t1, t2, P.t_min, qmax, dn, I.neg_inv_density, g, k0, k1, tm)) {
                                    if (scan) { ls.cl = tm; ls.item = (int)it; ls.medium = true; ls.has = true; }
                                    else if (!dfr || tm < closest) { closest = tm; best_item = (int)it; best_medium = true; }
                                }
                            }
                        }
                        }
                    }
                }
                if (best_item >= 0) {
                    have_hit = true;
                } else { // miss: black background (color.rs:21); the path ends
                    if (P.sky) pa.L = pa.L + pa.T * sky_color(pa.rd);
                    path_end(P, oidx, pa);
                    if (SIG) { atomicAdd(P.path_sig + (size_t)ltile * 64 + (oidx & 63u), sig); sig = 0ull; }
                    alive = false;
                }
            }
            if (__popcll(__ballot(have_hit)) >= threshold) break;
        }
        // ================= phase B: shade every lane that holds a hit =================
        if (__ballot(have_hit) == 0ull) break; // nobody holds a hit and nobody can trace: all done
        prof_tick<PROF>(prof, 16, have_hit);
        {
            const bool shading = have_hit;
            have_hit = false;
            if (SIG && shading) sig += (unsigned long long)sig_mix(__float_as_uint(closest), pa.depth);
            // all lanes call (wavefront texture lookup); the traversal stacks are idle now: LDS scratch
            const bool goes_on = shade_hit(sc, P.max_depth, P.ext, g, k0, k1, shading, closest, best_item, best_pf, best_medium, pa,
                                           reinterpret_cast<float *>(&lds_stack[wave][0][0][0]));
            if (shading && !goes_on) {
                // absorbed, emitter or depth limit: the path ends
                path_end(P, oidx, pa);
                if (SIG) { atomicAdd(P.path_sig + (size_t)ltile * 64 + (oidx & 63u), sig); sig = 0ull; }
                alive = false;
            }
        }
    }

    if (PROF) {
        __syncthreads();
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS && prof_lds[threadIdx.x] != 0ull) atomicAdd(P.prof + threadIdx.x, prof_lds[threadIdx.x]);
    }
}

// ----------------------------------------------------------------------------------
// render kernel, two-phase form with wave-cooperative BVH traversal (default).
// Same schedule as rtmi_render_kernel; the item scan of phase A is executed by ALL lanes (lanes
// without a pending query are workers for the others' BVH traversals).
// ----------------------------------------------------------------------------------
// INSTL: 0 = the instantiations of the common scenes: no transform code in their loops; 1 = scenes with instanced primitives or
// media inside transforms (rtmi.h); 2 = also DEFERRED items, list scans, nested media (children of a BVHNode that are not
// primitives).  Forced on the BASELINE scenes level 2 costs 10-14 % (profiles/r04_experiments/force_inst_ab.log), hence two levels.
template <bool SIG, bool PROF, int WPS, bool EXT, int INSTL>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, WPS) void rtmi_render_coop(DevScene sc, DevCamera cam, DevParams P) {
    constexpr bool INST = INSTL >= 1; // instanced primitives, media inside transforms
    constexpr bool INSD = INSTL >= 2; // DEFERRED items, list scans, nested media
    __shared__ unsigned long long prof_lds[PROF ? 2 * RTMI_PROF_SLOTS : 1];
    unsigned long long *prof = prof_lds;
    if (PROF) {
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS) prof_lds[threadIdx.x] = 0ull;
        __syncthreads();
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[]; // per wave: pool | ctx | best
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    CoopWork cw;
    cw.cap = (int)P.coop_cap;
    // per wave: pool | ctx | best | dummy | (lean: word ring) | (INSD: group state, RTMI_COOP_PARK_WORDS)
    constexpr uint32_t lds_tail = (EXT ? 0u : RTMI_RNG_RING_WORDS) + (INSD ? RTMI_COOP_PARK_WORDS : 0u);
    cw.wlds = lds_dyn + (size_t)wave * (2u * cw.cap + 64u * 12u + 128u + RTMI_COOP_DUMMY_WORDS + lds_tail);
    // INSD: the state of a group of DEFERRED items lives in LDS, not in registers that would stay live through every
    // traversal of every scene (parked: 20 VGPRs spilled -> see DESIGN.md §8d): per lane t0 | scan cl | scan holder | scan pf
    uint32_t *park = cw.wlds + 2u * cw.cap + 64u * 12u + 128u + RTMI_COOP_DUMMY_WORDS + (EXT ? 0u : RTMI_RNG_RING_WORDS) + lane;
    cw.spill_cap = (int)P.spill_cap;
    cw.spill = P.spill + (size_t)(blockIdx.x * WAVES_PER_BLOCK + wave) * P.spill_cap;
    unsigned long long sig = 0ull;
    WaveWork w;
    w.ltile = 0u; w.ps_base = 0u; w.obase = 0u; w.x0 = 0u; w.y0 = 0u; w.cols = 0u; w.n_valid = 0u; w.next = 0u; w.total = 0u;
    bool queue_empty = false;
    const uint32_t k0 = P.key0, k1 = P.key1;
    const int threshold = (int)P.shade_threshold;
    // Three scalars of the trace loop in registers of their OWN: as members of the kernel-argument block they sit in
    // eight-register tuples, and when the allocator spills such a tuple (this kernel keeps ~100 scalars in VGPR lanes) it
    // restores all eight registers wherever one member is read — found as 48 v_readlane per list scan for the sake of t_min.
    // The copy through an asm statement is a live range the coalescer cannot fold back into the tuple (r04: final_scene
    // +1.5 %, cornell_box +4 % together with the one-pointer primitive records; giving every scene POINTER its own pair
    // the same way lost 2 % / 7 % — profiles/r04_experiments/own_sgprs_ab.log: the allocator is not to be out-guessed twice)
    float t_min;
    uint32_t n_items, use_alt_w;
    asm volatile("s_mov_b32 %0, %3\n s_mov_b32 %1, %4\n s_mov_b32 %2, %5" : "=&s"(t_min), "=&s"(n_items), "=&s"(use_alt_w) : "s"(P.t_min), "s"(sc.n_items), "s"(P.use_alt));
    const bool use_alt = use_alt_w != 0u;

    uint32_t oidx = 0u, ltile = 0u; // slot of this lane's path in the per-sample buffer; its local tile (SIG only)
    bool alive = false, done = false, have_hit = false, overflow = false;
    // lean instantiation (scenes without alternative trees): word ring in LDS behind pool | ctx | best
    typename std::conditional<EXT, RngReg, RngRing>::type g;
    rng_attach(g, cw.wlds + 2u * cw.cap + 64u * 12u + 128u + RTMI_COOP_DUMMY_WORDS);
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
            if (__ballot(!have_hit && !done) == 0ull) break;
            prof_time<PROF>(prof, 31, tstamp); // loop overhead / phase switching
            { // lanes whose path ended take the next (sample, pixel) item of the chunk
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
            prof_tick<PROF>(prof, 0, need);
            prof_time<PROF>(prof, 25, tstamp); // camera samples
            RayF W;
            W.o = pa.ro; W.d = pa.rd;
            ray_derive(W);
            if (need) { closest = RTMI_FLT_MAX; best_item = -1; best_pf = 0; best_medium = false; }
            // INSD, parked in LDS: [0] the closest hit before a BVH item whose media / instanced-subtree children follow as DEFERRED
            // items (T0); [64] / [128] / [192] the scan of a list with media that was a child of a BVHNode (rtmi.h, LISTSCAN): its
            // closest hit so far, who holds it (item, bit 31: a medium; RTMI_PARK_NONE: nobody) and the primitive
            int grp_first = 0x7fffffff;    // the index of the SAVE_T0 item (or of the first deferred one), and whether it is the enclosing tree
            bool grp_tree = false;
            if (INSD) park[0] = __float_as_uint(RTMI_FLT_MAX);
            for (uint32_t it = 0; it < n_items; it++) { // executed by all 64 lanes
                const rtmi_item I = RTMI_UNIFORM_LOAD(rtmi_item, &sc.items[it].it);
                if (INSD && (I.flags & RTMI_ITEMFLAG_SAVE_T0)) {
                    park[0] = __float_as_uint(closest); grp_first = (int)it; grp_tree = I.kind == RTMI_ITEM_BVH && !(I.flags & RTMI_ITEMFLAG_DEFERRED);
                }
                if (INSD && (I.flags & RTMI_ITEMFLAG_LISTSCAN_END)) { // wave-uniform: the terminator of a list scan
                    const uint32_t holder = park[128];
                    if (need && holder != RTMI_PARK_NONE) {
                        ListScan ls;
                        ls.cl = __uint_as_float(park[64]); ls.item = (int)(holder & 0x7fffffffu); ls.pf = (int)park[192]; ls.medium = (holder >> 31) != 0u; ls.has = true;
                        listscan_fold(ls, I.first, closest, best_item, best_pf, best_medium, grp_first, grp_tree);
                    }
                    continue;
                }
                if (INSD && (I.flags & RTMI_ITEMFLAG_LISTSCAN_BEGIN)) { park[64] = park[0]; park[128] = RTMI_PARK_NONE; }
                const bool scan = INSD && (I.flags & RTMI_ITEMFLAG_LISTSCAN_MEMBER) != 0u; // wave-uniform
                RayF R = W;
                if (I.xform_count > 0) { // both transforms of a chain of two in ONE scalar fetch (they follow the item record)
                    struct XPair { rtmi_xform x0, x1; };
                    const XPair XP = RTMI_UNIFORM_LOAD(XPair, reinterpret_cast<const XPair *>(&sc.items[it].x0));
                    if (xform_ray_item<true>(sc.xforms, I.xform_first, I.xform_count, XP.x0, XP.x1, R.o, R.d)) ray_derive(R);
                }
                const int slot = 1 + (it < 11u ? (int)it : 11);
                if (!(I.flags & RTMI_ITEMFLAG_MEDIUM)) {
                    float t;
                    int pf;
                    // ONE query site (three inlined copies of the traversal made the INSD instantiations twice the code of the
                    // others): a DEFERRED item — an instanced subtree that was a child of a BVHNode, or a primitive member of a
                    // list scan (rtmi.h) — is reached through its gate and queried up to the t_max its group was entered with
                    // (the scan's closest hit so far); what its hit means is decided afterwards
                    bool reach = need;
                    float qmax = closest;
                    const bool dfi = INSD && (I.flags & RTMI_ITEMFLAG_DEFERRED) != 0u; // wave-uniform
                    if (dfi) {
                        const float t0_saved = __uint_as_float(park[0]);
                        reach = need && deferred_gate(sc, I, W, t_min, t0_saved);
                        qmax = scan ? __uint_as_float(park[64]) : t0_saved;
                    }
                    if (geom_query_coop<PROF, EXT, EXT, INST>(sc, I, use_alt, reach, R, pa.rtime, t_min, qmax, cw, t, pf, overflow, prof, slot)) {
                        if (scan) {
                            park[64] = __float_as_uint(t); park[128] = it; park[192] = (uint32_t)pf;
                        } else if (!dfi || deferred_bvh_wins(I.count, t, closest, best_item, best_pf, grp_first, grp_tree)) {
                            closest = t; best_item = (int)it; best_pf = pf; best_medium = false;
                        }
                    }
                    prof_time<PROF>(prof, I.kind == RTMI_ITEM_BVH ? (it == 0 ? 27 : 28) : 26, tstamp);
                } else {
                    // ConstantMedium::hit — medium.rs:28-56
                    float t1 = 0.0f, t2 = 0.0f, tm;
                    int pf;
                    bool h1, h2;
                    if (I.flags & RTMI_ITEMFLAG_DEV_MEDIUM_SPHERE) {
                        // boundary = one static sphere (wave-uniform test; the sphere travels in the item record): both
                        // boundary queries are roots of the same quadratic, evaluated once (same expressions as two
                        // Sphere::hit calls: same bits)
                        h1 = false; h2 = false;
                        if (need) sphere_two_queries(R, make_float4(I.root_min[0], I.root_min[1], I.root_min[2], I.root_max[0]), h1, t1, h2, t2);
                    } else {
                        // INSD: a medium that was a child of a BVHNode (rtmi.h, DEFERRED) is reached through its parent's box
                        const bool dfr = INSD && (I.flags & RTMI_ITEMFLAG_DEFERRED) != 0u; // wave-uniform
                        bool reach = need;
                        float t0_saved = RTMI_FLT_MAX;
                        if (dfr) {
                            t0_saved = scan ? __uint_as_float(park[64]) : __uint_as_float(park[0]); // the interval's end: the scan's closest hit, or T0
                            reach = need && deferred_gate(sc, I, W, t_min, __uint_as_float(park[0]));
                        }
                        h1 = geom_query_coop<PROF, EXT, false, INST>(sc, I, use_alt, reach, R, pa.rtime, -RTMI_FLT_MAX, RTMI_FLT_MAX, cw, t1, pf, overflow, prof, slot);
                        h2 = geom_query_coop<PROF, EXT, false, INST>(sc, I, use_alt, reach && h1, R, pa.rtime, t1 + 0.0001f, RTMI_FLT_MAX, cw, t2, pf, overflow, prof, slot);
                        if (INSD && (I.flags & RTMI_ITEMFLAG_NESTED_MEDIUM)) { // wave-uniform: the boundary is itself a medium (rtmi.h)
                            if (reach && h1 && h2) h1 = nested_medium_interval(sc, I, medium_dir_norm<INST>(sc, I.flags, I.xform_first, W), g, k0, k1, t1, t2);
                        }
                        if (dfr) { // its interval ends at the t_max the BVH was entered with; its hit must beat what the tree found
                            if (reach && h1 && h2 && medium_sample(t1, t2, t_min, t0_saved, medium_dir_norm<INST>(sc, I.flags, I.xform_first, W), I.neg_inv_density, g, k0, k1, tm)) {
                                if (scan) { park[64] = __float_as_uint(tm); park[128] = it | 0x80000000u; }
                                else if (tm < closest) { closest = tm; best_item = (int)it; best_medium = true; }
                            }
                            h1 = false; // done
                        }
                    }
                    if (need && h1 && h2) {
                        // (one square root per medium: sharing it between the media of a query measured -1.1 %,
                        // profiles/r03_experiments/wdn_shared_norm_ab.log)
                        if (medium_sample(t1, t2, t_min, closest, medium_dir_norm<INST>(sc, I.flags, I.xform_first, W), I.neg_inv_density, g, k0, k1, tm)) {
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
                    if (P.sky) pa.L = pa.L + pa.T * sky_color(pa.rd);
                    path_end(P, oidx, pa);
                    if (SIG) { atomicAdd(P.path_sig + (size_t)ltile * 64 + (oidx & 63u), sig); sig = 0ull; }
                    alive = false;
                }
            }
            if (__popcll(__ballot(have_hit)) >= threshold) break;
        }
        // ================= phase B =================
        if (__ballot(have_hit) == 0ull) break;
        prof_tick<PROF>(prof, 16, have_hit);
        {
            const bool shading = have_hit;
            have_hit = false;
            if (SIG && shading) sig += (unsigned long long)sig_mix(__float_as_uint(closest), pa.depth);
            // all lanes call (wavefront texture lookup); the traversal pool is idle now: LDS scratch
            const bool goes_on = shade_hit<decltype(g), INST>(sc, P.max_depth, P.ext, g, k0, k1, shading, closest, best_item, best_pf, best_medium, pa,
                                           reinterpret_cast<float *>(cw.wlds));
            if (shading && !goes_on) {
                path_end(P, oidx, pa);
                if (SIG) { atomicAdd(P.path_sig + (size_t)ltile * 64 + (oidx & 63u), sig); sig = 0ull; }
                alive = false;
            }
        }
        prof_time<PROF>(prof, 30, tstamp); // shading
    }
    if (P.ext & RTMI_EXT_TEST_OVERFLOW) overflow = true; // test knob: exercise the error path
    if (__ballot(overflow) != 0ull && lane == 0) atomicAdd(P.status, 1u); // reported loudly by the host

    if (PROF) {
        __syncthreads();
        if (threadIdx.x < 2 * RTMI_PROF_SLOTS && prof_lds[threadIdx.x] != 0ull) {
            if (threadIdx.x == 2 * 18) atomicMax(P.prof + threadIdx.x, prof_lds[threadIdx.x]); // slot 18: a maximum
            else atomicAdd(P.prof + threadIdx.x, prof_lds[threadIdx.x]);
        }
    }
}

#ifndef RTMI_LEAN_TU
// the lean instantiations (EXT = false) are compiled in rtmi_lean.hip, with another machine-scheduler strategy
extern template __global__ void rtmi_render_coop<false, false, 4, false, 0>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_coop<false, false, 4, false, 1>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_coop<false, false, 4, false, 2>(DevScene, DevCamera, DevParams);
#endif

// ----------------------------------------------------------------------------------
// The workgroup-cooperative kernel (RTMI_FLAG_BLOCK_COOP) and the asynchronous state-machine kernel (RTMI_FLAG_ASYNC)
// are defined in rtmi_kernels_alt.hpp and instantiated in rtmi_alt.hip — measured slower than the default, kept as
// independent implementations for the parity tests, out of the headline translation unit.  Declarations for the launch:
// ----------------------------------------------------------------------------------
#define RTMI_BLK_WAVES 4
#define RTMI_BLK_THREADS (64 * RTMI_BLK_WAVES)
// LDS of a workgroup in uint32 words: pool [cap][2] | ctx [4][T][4] | best [T][2] | dummy [T][2] | sync [16]
#define RTMI_BLK_LDS_WORDS(cap) (2u * (cap) + RTMI_BLK_THREADS * 20u + 16u)
template <bool SIG, bool INST>
__global__ void rtmi_render_bcoop(DevScene sc, DevCamera cam, DevParams P);
template <bool FAST, bool SIG, bool PROF>
__global__ void rtmi_render_async(DevScene sc, DevCamera cam, DevParams P);
#ifndef RTMI_ALT_TU
extern template __global__ void rtmi_render_bcoop<true, false>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_bcoop<false, false>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_async<true, false, true>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_async<false, false, true>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_async<true, true, false>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_async<true, false, false>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_async<false, true, false>(DevScene, DevCamera, DevParams);
extern template __global__ void rtmi_render_async<false, false, false>(DevScene, DevCamera, DevParams);
#endif

#ifndef RTMI_LEAN_TU /* plain (non-template) kernels: defined once, in rtmi_device.hip */
// `col += color(..)` in sample order, then `col /= ns; sqrt; clamp; (255.99*c) as i32` — tests/test.rs:69-78,
// per local texel.  One thread per (local tile, pixel): adds this pass's samples to the f64 sum (`acc`, carried
// between passes when the per-sample buffer does not hold all ns samples at once) and, on the last pass, writes
// the texel.  A wavefront reads 1 KB contiguous per sample.
__global__ __launch_bounds__(256) void rtmi_resolve_kernel(const Rad3 *__restrict__ samples, double *__restrict__ acc,
                                                           rtmi_texel *__restrict__ out, DevParams P, int first, int last,
                                                           int partial) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P.ntiles_local * 64u) return;
    // a traversal-pool overflow in any pass of this call invalidates its texels: poison them (rtmi_untile and
    // rtmi_scene_status report it) instead of handing out plausible numbers
    const bool poisoned = P.status[0] != 0u;
    const uint32_t ltile = tid >> 6, lane = tid & 63u;
    const uint32_t tile = ltile * P.tile_world + P.tile_rank;
    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const bool in_image = tx * RTMI_TILE + (lane & 7u) < P.nx && ty * RTMI_TILE + (lane >> 3) < P.ny;
    double sum[3] = {0.0, 0.0, 0.0};
    double *a = acc + ((size_t)ltile * 3) * 64 + lane;
    if (!first) { sum[0] = a[0]; sum[1] = a[64]; sum[2] = a[128]; }
    if (in_image) {
        const Rad3 *src = samples + ((size_t)ltile * P.pass_stride) * 64u + lane;
        uint32_t s = 0;
        for (; s + 8u <= P.pass_cnt; s += 8u) { // 8 independent loads in flight, additions in sample order
            Rad3 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = src[(size_t)(s + k) * 64u];
#pragma unroll
            for (int k = 0; k < 8; k++) { sum[0] += (double)v[k].r; sum[1] += (double)v[k].g; sum[2] += (double)v[k].b; }
        }
        for (; s < P.pass_cnt; s++) {
            const Rad3 v = src[(size_t)s * 64u];
            sum[0] += (double)v.r; sum[1] += (double)v.g; sum[2] += (double)v.b;
        }
    }
    if (!last) {
        a[0] = sum[0]; a[64] = sum[1]; a[128] = sum[2];
        if (!partial) return; // RTMI_FLAG_PROGRESSIVE: the texel of the samples so far is written after every pass
    }
    // col /= ns of the samples summed so far: all of them after the last pass (tests/test.rs:71)
    const double n_so_far = last ? (double)P.ns : (double)(P.pass_s0 + P.pass_cnt);
    rtmi_texel tx_out;
    uint32_t q[3];
    float lin[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const double m = sum[ch] / n_so_far;
        lin[ch] = (float)m;
        double g = sqrt(m);
        g = (g > 0.0) ? ((g < 1.0) ? g : 1.0) : 0.0; // nalgebra::clamp(val, 0, 1); NaN -> 0
        const double x = 255.99 * g;
        q[ch] = (x != x) ? 0u : (uint32_t)(int32_t)x; // `as i32`; in [0,255] after the clamp
    }
    tx_out.r = lin[0]; tx_out.g = lin[1]; tx_out.b = lin[2];
    tx_out.rgb8 = q[0] | (q[1] << 8) | (q[2] << 16);
    if (poisoned) {
        const float nan = __uint_as_float(0x7fc00000u);
        tx_out.r = nan; tx_out.g = nan; tx_out.b = nan;
        tx_out.rgb8 = RTMI_TEXEL_POISON;
    }
    out[tid] = tx_out;
}
// end of a pass (one thread): the unit counter goes back to zero for the next pass, the pass's units are added to
// the progress word, and after the last pass the call's overflow count joins the sticky word
__global__ void rtmi_pass_end_kernel(unsigned int *status, unsigned int units, unsigned int pass_spp, int last) {
    status[3] += units;
    status[4] += pass_spp;
    status[1] = 0u;
    if (last) status[2] += status[0];
}

// ---- device evaluation of the arithmetic contract, for parity tests --------------------
// op: 0 sin, 1 log, 2 atan2(x,y), 3 asin, 4 x/y, 5 sqrt, 6 u01(bits of x), 16.. instance transforms
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
    case 6: r = rtmi_u01(__float_as_uint(x[i])); break;
    default: { // 16 + 4*(axis) + which: instance transforms (rotate.rs:85-113) by sin/cos of 33 degrees
        const int axis = ((op - 16) >> 2) % 3, which = (op - 16) & 3;
        rtmi_xform X;
        X.kind = RTMI_XF_ROTATE_X + axis; X.x = 0.5446390509605408f; X.y = 0.838670551776886f; X.z = 0.0f;
        F3 o, d = f3(0, 0, 0);
        // put (x, y) into the (a, b) components of this axis: X -> (y,z), Y -> (z,x), Z -> (x,y)
        if (axis == 0) o = f3(0.25f, x[i], y[i]); else if (axis == 1) o = f3(y[i], 0.25f, x[i]); else o = f3(x[i], y[i], 0.25f);
        F3 n = o;
        if (which < 2) xform_ray(&X, 0, 1, o, d); else xform_hit(&X, 0, 1, o, n);
        if (op >= 28) o = n; // 28..39: the normal instead of the point
        const float a = axis == 0 ? o.y : (axis == 1 ? o.z : o.x), b = axis == 0 ? o.z : (axis == 1 ? o.x : o.y);
        r = (which & 1) ? b : a;
        break;
    }
    }
    out[i] = r;
}
// instance transforms exactly as the render kernels call them: xforms in global memory, runtime count
__global__ void rtmi_xform_probe_kernel(const rtmi_xform *xf, int count, const float *a, const float *b, float *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 o = f3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), d = f3(b[3 * i], b[3 * i + 1], b[3 * i + 2]);
    F3 p = o, nn = d;
    const bool rot = xform_ray(xf, 0, count, o, d);
    xform_hit(xf, 0, count, p, nn);
    float *r = out + 13 * (size_t)i;
    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
    r[6] = p.x; r[7] = p.y; r[8] = p.z; r[9] = nn.x; r[10] = nn.y; r[11] = nn.z; r[12] = rot ? 1.0f : 0.0f;
}
__global__ void rtmi_philox_probe_kernel(const uint32_t *ctr, const uint32_t *key, uint32_t *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o0, o1, o2, o3;
    philox(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[2 * i], key[2 * i + 1], o0, o1, o2, o3);
    out[4 * i] = o0; out[4 * i + 1] = o1; out[4 * i + 2] = o2; out[4 * i + 3] = o3;
}
#endif // RTMI_LEAN_TU
