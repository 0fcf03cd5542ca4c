/* rtmi.h — C ABI of the MI355X (gfx950) device path for the per-pixel render loop of
 * DrStiev/raytracing_rust.
 *
 * What this boundary replaces.  The reference has no FFI; the seam is the pair
 *   create_image(ny, nx, ns, cam, world) -> String      (tests/test.rs:55-85)
 *   color(ray, world, depth) -> Vector3<f64>            (src/color.rs:6-23)
 * plus everything they call through the traits Hittable (src/hittable.rs:18-21),
 * Material (src/material.rs:30-33) and Texture (src/texture.rs:4-6).  A host (the C++
 * mirror in raytracing_rust_amd/host/, or a Rust shim — see INTEGRATION.md) lowers its
 * object graph into the flat, plain-old-data scene description below and calls
 * rtmi_render*, which runs the triple loop of create_image on the GPU.
 *
 * Conventions: plain pointers and sizes only; inputs are borrowed for the duration of
 * the call; outputs are caller-allocated; every entry point returns 0 on success or an
 * RTMI_ERR_* code (message via rtmi_last_error()); nothing throws across the boundary.
 * Thread model.  A scene handle (rtmi_scene: one device; rtmi_multi: a device list) owns the device copy of the
 * scene — read-only after creation — AND the scratch its render calls work in (unit queue and status words, the
 * per-sample radiance buffer, f64 sums, the cached texel / signature buffers of the blocking calls).  Render calls on
 * ONE handle therefore serialise: on the host by a per-handle mutex, on the device by an event chain (a render
 * enqueued on any stream starts after the previous render of that handle has finished).  Any number of threads may
 * call into one handle; each gets the image a single thread would get.  Different handles — also on the same device —
 * are independent and render concurrently.  The reference's convention being replaced: one thread, Rc (bvh.rs:11-12).
 *
 * Arithmetic: fp32 on the device under the contract written in DESIGN.md ("fp32
 * arithmetic contract"); transcendental functions are those of rtmi_math.h; random
 * numbers are Philox4x32-10 streams keyed by `seed` with counter
 * (block, sample, pixel = j*nx+i with j counted from the bottom row, 0).
 */
#ifndef RTMI_H
#define RTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTMI_ABI_VERSION 7u /* the scene description (6: instanced primitives, rtmi_prim_meta.flags bits 4..7, 12..31;
                             * 7: deferred media, RTMI_ITEMFLAG_SAVE_T0 / _DEFERRED) */
#define RTMI_MAX_BVH_DEPTH 24u /* per-lane LDS traversal stack entries */
#define RTMI_TILE 8u           /* a wavefront renders an 8x8 pixel tile: lane = pixel */
#define RTMI_SAMPLE_SLOT_BYTES 12u /* per-sample radiance buffer: three fp32 per finished path (budget arithmetic of
                                   * rtmi_render_params.sample_buffer_bytes) */

enum { RTMI_OK = 0, RTMI_ERR_INVALID = 1, RTMI_ERR_UNSUPPORTED = 2, RTMI_ERR_DEVICE = 3, RTMI_ERR_NOMEM = 4, RTMI_ERR_CANCELLED = 5 };

/* ---- textures: Texture::value (src/texture.rs) -------------------------------- */
enum { RTMI_TEX_SOLID = 0, RTMI_TEX_CHECKER = 1, RTMI_TEX_NOISE = 2, RTMI_TEX_IMAGE = 3 };
typedef struct {
    int32_t kind;
    int32_t i0; /* CHECKER: odd texture index | NOISE: perlin table index | IMAGE: image index */
    int32_t i1; /* CHECKER: even texture index.  Checkers may nest (texture.rs:28-48 is generic over its children) up to
                 * 16 levels; a deeper nest or a checker that reaches itself is RTMI_ERR_INVALID at rtmi_scene_create */
    int32_t pad;
    float f0, f1, f2; /* SOLID: r,g,b | NOISE: f0 = scale */
    float f3;
} rtmi_texture; /* 32 B */

/* Perlin tables of one NoiseTexture (src/perlin.rs:59-74): 256 unit vectors + 3 permutations */
typedef struct {
    float ranvec[256 * 4]; /* x,y,z,0 */
    int32_t perm[3 * 256]; /* perm_x, perm_y, perm_z */
} rtmi_perlin;

typedef struct {
    uint64_t offset; /* byte offset of row-major RGB8 texels inside image_data */
    uint32_t nx, ny;
} rtmi_image;

/* ---- materials: Material::{scatter,emitted} (src/material.rs) ----------------- */
enum {
    RTMI_MAT_LAMBERTIAN = 0,
    RTMI_MAT_METAL = 1,
    RTMI_MAT_DIELECTRIC = 2,
    RTMI_MAT_DIFFUSE_LIGHT = 3,
    RTMI_MAT_ISOTROPIC = 4
};
#define RTMI_MATFLAG_NEEDS_UV 1u /* texture tree contains an IMAGE texture */
typedef struct {
    int32_t kind;
    int32_t tex;
    float param; /* METAL: fuzz (already clamped to <= 1, material.rs:70) | DIELECTRIC: ref_idx */
    uint32_t flags;
} rtmi_material; /* 16 B */

/* ---- primitives: SoA planes of float4 ------------------------------------------
 * plane A[i]: SPHERE/MSPHERE {c0.x, c0.y, c0.z, radius}   (sphere.rs:20-24, 87-94)
 *             RECT           {x0, y0, x1, y1}             (rect.rs:14-22)
 *             CUBE           {min.x, min.y, min.z, max.x} (cube.rs:8-12)
 * plane B[i]: MSPHERE        {c1-c0 (x,y,z), time0}
 *             RECT           {k, 0, 0, 0}
 *             CUBE           {max.y, max.z, 0, 0}
 * meta[i]   : material, flags, inv_dt (MSPHERE: 1/(time1-time0)), type              */
enum { RTMI_PRIM_SPHERE = 0, RTMI_PRIM_MSPHERE = 1, RTMI_PRIM_RECT = 2, RTMI_PRIM_CUBE = 3 };
#define RTMI_PRIMFLAG_FLIP 1u /* FlipNormals (hittable.rs:67-88) folded into the primitive */
#define RTMI_PRIMFLAG_PLANE_SHIFT 8 /* RECT: Plane YZ=0, ZX=1, XY=2 (rect.rs:8-12) in bits 8..9 */
/* Instanced primitive: Traslate<H> / Rotate<H> (src/traslate.rs:6-9, src/rotate.rs:21-28) are generic over any
 * Hittable, so the reference lets them wrap a single primitive anywhere — as a member of a nested list or as a child of
 * a BVHNode (bvh.rs:11-12).  Such a primitive carries its own transform chain: bits 4..7 = number of transforms
 * (0 = none, at most RTMI_PRIM_XF_MAX), bits 12..31 = index of the first one in `xforms` (outermost wrapper first, like an
 * item's chain).  The ray is taken into the primitive's frame before its test and the hit point / normal back after it
 * (traslate.rs:18-24, rotate.rs:85-113), inside whatever frame the item's own chain has established.  In a BVH the node
 * boxes above it are the reference's (Rotate::bounding_box is the whole space, rotate.rs:36-37). */
#define RTMI_PRIMFLAG_XF_COUNT_SHIFT 4
#define RTMI_PRIMFLAG_XF_FIRST_SHIFT 12
#define RTMI_PRIM_XF_MAX 15u
typedef struct {
    int32_t material;
    uint32_t flags;
    float inv_dt;
    int32_t type;
} rtmi_prim_meta; /* 16 B */

/* ---- BVH: BVHNode (src/bvh.rs:9-14) flattened; one record holds BOTH children's boxes.
 * child >= 0: index of an internal node; child < 0: leaf = 0x80000000 | type<<28 | prim.
 * The tree topology is the reference's (median split on a random axis, bvh.rs:17-66):
 * the traversal result (closest hit, ties -> right child, bvh.rs:75-81) depends on it
 * only through ties and through which boxes prune, both of which are preserved. */
typedef struct {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2]; /* reserved: ignored on input (the library's device copy keeps derived child references here) */
} rtmi_bvh_node; /* 64 B */
/* Node of an ALTERNATIVE tree (see prim_gate): four children, SoA boxes.  child: node index (into alt_nodes),
 * RTMI_LEAF(type, prim), or RTMI_NO_CHILD with an empty box (min > max). */
typedef struct {
    float minx[4], miny[4], minz[4], maxx[4], maxy[4], maxz[4];
    int32_t child[4];
    int32_t pad[4]; /* reserved: ignored on input */
} rtmi_bvh4_node; /* 128 B */
#define RTMI_NO_CHILD ((int32_t)0x7fffffff)
#define RTMI_LEAF(type, prim) ((int32_t)(0x80000000u | ((uint32_t)(type) << 28) | (uint32_t)(prim)))

/* ---- instance transforms: Traslate (src/traslate.rs), Rotate (src/rotate.rs) ---- */
enum { RTMI_XF_TRANSLATE = 0, RTMI_XF_ROTATE_X = 1, RTMI_XF_ROTATE_Y = 2, RTMI_XF_ROTATE_Z = 3,
       /* not transforms: the two records BEHIND the chain of a DEFERRED BVH item hold its gate box (x, y, z = min / max) */
       RTMI_XF_GATE_MIN = 4, RTMI_XF_GATE_MAX = 5,
       /* not a transform either: x = -(1/density) of the INNER medium of a nested pair (RTMI_ITEMFLAG_NESTED_MEDIUM), the
        * record behind the chain (and behind the two gate records of a DEFERRED BVH item) */
       RTMI_XF_INNER_MEDIUM = 6 };
typedef struct {
    int32_t kind;
    float x, y, z; /* TRANSLATE: offset | ROTATE_*: x = sin(theta), y = cos(theta) */
} rtmi_xform; /* 16 B */

/* ---- top-level list entries: HittableList (src/hittable.rs:24-47) of
 *      [FlipNormals][ConstantMedium][Traslate/Rotate chain] geometry
 * geometry is a run of primitives scanned in order (a nested HittableList, a Cube, a
 * single primitive) or a BVH.  xforms are listed outermost first. */
enum { RTMI_ITEM_LIST = 0, RTMI_ITEM_BVH = 1 };
#define RTMI_ITEMFLAG_FLIP 1u
#define RTMI_ITEMFLAG_MEDIUM 2u /* ConstantMedium (src/medium.rs): geometry is the boundary */
/* MEDIUM items: bits 8..11 = how many of the item's transforms (its FIRST ones) wrap the ConstantMedium itself instead of its
 * boundary — Traslate(ConstantMedium(..)) / Rotate(ConstantMedium(..)), which the generic wrappers allow (traslate.rs:6-9).
 * The boundary is still queried through the whole chain; the medium's own arithmetic (medium.rs:38-47: the norm of the
 * ray direction, the scattering point) runs on the ray as those outer wrappers hand it down, and the point and normal go
 * back through them (traslate.rs:21-22, rotate.rs:94-105).  0 = the medium is outside all transforms. */
#define RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT 8
/* A ConstantMedium that is a CHILD OF A BVHNode (bvh.rs:11-12 takes any Hittable).  BVHNode::hit hands both children the
 * query's own (t_min, t_max) and keeps the closer hit (bvh.rs:70-89), so such a medium (i) is evaluated with the t_max the
 * BVH was entered with, not with what its siblings have found, (ii) draws its random number whenever its clamped
 * boundary interval is not empty (medium.rs:30-40), at its in-order position, (iii) is reached iff every ancestor's box
 * passes AABB::hit — nested boxes and a monotone slab test: iff its PARENT's box passes.  The trees of the description
 * hold primitives only; such a medium is a DEFERRED item that follows its BVH item in the list (several: in traversal
 * order; a BVHNode over one element, bvh.rs:44-45, evaluates — and draws — twice: two items):
 *   RTMI_ITEMFLAG_SAVE_T0   on the BVH item (or, when the BVH holds nothing but media and there is no BVH item, on the
 *                           first deferred one): remember the closest hit so far as it stands BEFORE this item (T0);
 *   RTMI_ITEMFLAG_DEFERRED  a MEDIUM item: its boundary is queried only for rays whose gate passes
 *                           AABB::hit(t_min, T0) — the gate is the box of the BVHNode the medium was a child of: prim_gate of
 *                           the item's first primitive (kind LIST), the two records behind the chain as below (a boundary
 *                           that is itself a BVHNode: kind BVH) —, the ray is the one the FIRST G transforms of the item hand
 *                           down (G in bits 12..15: the transforms of the enclosing BVH item, copied in front of the
 *                           medium's own) —, its interval is clamped to T0, and its hit is accepted when closer than the
 *                           closest hit so far (an exact tie with a primitive, of probability zero, goes to the primitive).
 * The same for a BVHNode inside Traslate / Rotate that is a child of a BVHNode (an instanced subtree; traslate.rs:6-9 and
 * rotate.rs:21-28 wrap any Hittable): a DEFERRED item of kind BVH without the MEDIUM flag — its chain = the enclosing item's
 * transforms, then its own; its gate box travels in the two xform records behind that chain (RTMI_XF_GATE_MIN / _MAX);
 * its query runs with t_max = T0; `count` = the number of primitive leaves of the enclosing tree that precede it in
 * traversal order (an exact tie with one of those goes to the subtree, with a later one to the leaf — bvh.rs:75-81: the
 * later child wins; ties between deferred items of one group go to the later item). */
#define RTMI_ITEMFLAG_SAVE_T0 4u
#define RTMI_ITEMFLAG_DEFERRED 8u
/* A ConstantMedium whose boundary is a ConstantMedium (medium.rs:11-15 is generic over any Hittable): the outer medium's
 * two boundary queries (medium.rs:30-31) are two evaluations of the inner medium — each of which queries the geometry
 * twice and draws its own random number when its interval is not empty — and their two random distances bound the
 * interval the outer medium then samples (third draw).  MEDIUM items with this flag carry the inner medium's -(1/density)
 * in an RTMI_XF_INNER_MEDIUM record behind their chain; material and density of the item are the OUTER medium's (the hit
 * record is its own, medium.rs:47-54).  One level, no wrappers between the two media. */
#define RTMI_ITEMFLAG_NESTED_MEDIUM 16u
/* A HittableList with media among its members as a child of a BVHNode.  The list's scan (hittable.rs:37-47) starts from the
 * t_max the BVH was entered with (T0) and hands every member the closest hit of the members before it; a medium member draws
 * whenever its boundary interval clamped to THAT is not empty; the BVH then folds the list's result like any child's
 * (bvh.rs:75-81).  Lowered as a group of DEFERRED items behind the BVH item, in member order: every member (a primitive,
 * also inside its own Traslate / Rotate, or a medium) is an item with LISTSCAN_MEMBER — evaluated with t_max = the scan's
 * closest hit so far, which LISTSCAN_BEGIN on the first member resets to T0; its gate is the box of the node that holds the
 * list, like every deferred item's — and the group ends with a terminator: kind LIST, count 0, LISTSCAN_END, `first` = the
 * number of primitive leaves of the enclosing tree that precede the list in traversal order; there the scan's result meets
 * the closest hit so far (closer wins; an exact tie of a primitive goes to the later child by that number; of a medium, to
 * the other side). */
#define RTMI_ITEMFLAG_LISTSCAN_BEGIN 32u
#define RTMI_ITEMFLAG_LISTSCAN_MEMBER 64u
#define RTMI_ITEMFLAG_LISTSCAN_END 128u
#define RTMI_ITEMFLAG_GATE_OUTER_SHIFT 12
typedef struct {
    int32_t kind;
    int32_t first;           /* LIST: first primitive | BVH: root node */
    int32_t count;           /* LIST: number of primitives */
    uint32_t flags;
    int32_t xform_first, xform_count;
    int32_t medium_material; /* MEDIUM: index of the Isotropic phase material (medium.rs:19-24) */
    float neg_inv_density;   /* MEDIUM: -(1/density) (medium.rs:40) */
    float root_min[3], root_max[3]; /* BVH: bbox of the root node (bvh.rs:60-64) */
    float scale;             /* BVH: largest |coordinate| of the root box (fast-cull margins only); 1e30 = never prune
                              * (a BVH whose boxes do not contain their primitives, e.g. Rect::bounding_box of a
                              * YZ/ZX rect, rect.rs:71-75: pruned traversal then visits what BVHNode::hit visits) */
    int32_t alt_first;       /* BVH: root (index into alt_nodes) of the alternative tree over the same primitives, or -1 */
} rtmi_item; /* 64 B */

typedef struct {
    uint32_t abi_version; /* RTMI_ABI_VERSION */
    uint32_t n_items;
    const rtmi_item *items;
    uint32_t n_prims;
    const float *prim_a; /* n_prims * 4 */
    const float *prim_b; /* n_prims * 4 */
    const rtmi_prim_meta *prim_meta;
    /* Optional (NULL = none): per primitive the box of its PARENT BVHNode in the reference tree, 8 floats
     * {min.xyz, 0, max.xyz, 0}.  BVHNode::hit (bvh.rs:70-73) reaches a leaf iff every ancestor's box passes
     * AABB::hit with the query's (t_min, t_max); the slab test is monotone in the box, so that is equivalent
     * to the parent's box passing.  With this "gate" any conservative tree over the same primitives (an item's
     * alt_first: a 4-wide SAH tree on the primitives' true extents, alt_nodes) returns the reference's result bit for bit:
     * accept a primitive iff its own test AND its gate pass, keep the minimum t, ties -> larger primitive
     * index (= rightmost leaf of the reference tree).  Used by the cooperative kernel; the exact and per-lane
     * kernels walk the reference tree (items' first). */
    const float *prim_gate;
    uint32_t alt_max_depth; /* deepest alternative tree (sizes the traversal stack's global part) */
    uint32_t n_alt_nodes;
    const rtmi_bvh4_node *alt_nodes;
    uint32_t n_nodes;
    const rtmi_bvh_node *nodes;
    uint32_t n_xforms;
    const rtmi_xform *xforms;
    uint32_t n_materials;
    const rtmi_material *materials;
    uint32_t n_textures;
    const rtmi_texture *textures;
    uint32_t n_perlin;
    const rtmi_perlin *perlin;
    uint32_t n_images;
    const rtmi_image *images;
    const uint8_t *image_data;
    uint64_t image_bytes;
    uint32_t max_bvh_depth; /* must be <= RTMI_MAX_BVH_DEPTH */
    /* Ray times for which the BVH boxes contain their moving spheres (the intersection of the [time0, time1] of
     * every MovingSphere inside a BVH; -FLT_MAX..FLT_MAX when there is none).  A camera whose shutter interval
     * leaves this range is rendered with exact instead of pruned traversal (same image, slower). */
    float bvh_time_lo, bvh_time_hi;
} rtmi_scene_desc;

/* Camera state (src/camera.rs:8-18), already derived by Camera::new (camera.rs:21-51) */
typedef struct {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    float time0, time1, lens_radius;
} rtmi_camera;

#define RTMI_FLAG_FAST_CULL 1u /* prune BVH subtrees behind the closest hit (same results) */
#define RTMI_FLAG_PATH_SIG 2u  /* also accumulate the per-pixel path signature into path_sig */
#define RTMI_FLAG_PROFILE 4u   /* diagnostics build: lane-activity counters into prof (64 uint64); slow */
#define RTMI_FLAG_SYNC 8u      /* per-lane BVH traversal instead of the wave-cooperative one (exact mode always is) */
#define RTMI_FLAG_ASYNC 16u    /* per-lane state-machine kernel (experimental, kept for comparison) */
#define RTMI_FLAG_REF_TREE 64u /* cooperative kernel: walk the reference-topology tree, not the alternative one */
#define RTMI_FLAG_BLOCK_COOP 32768u /* experimental: the four wavefronts of a workgroup share one traversal stack
                                * (csrc/rtmi_bvh_block.hpp); scenes whose BVH items all carry alternative trees, else ignored.
                                * Its stack (2432 entries in LDS) has NO global-memory part: a round takes only as many
                                * entries as its pushes fit, and a stack that cannot take one visit makes the call fail
                                * with RTMI_ERR_DEVICE (overflow) — possible for trees whose visits keep all four children
                                * level after level, which the default kernel (stack continued in global memory) renders.
                                * An independent implementation for the parity tests, 23 % slower than the default. */
/* Diagnostic knobs in the upper flag bits (results never depend on them): bits 8..10 = wavefronts per SIMD the
 * cooperative kernel is compiled for (3 or 5; default 4); bit 11 = a 256-entry LDS part of the traversal stack,
 * so that it spills to global memory all the time (tests/test_gpu_parity.py). */
#define RTMI_FLAG_SKY 32u      /* opt-in extension, off by default: a ray that misses the world returns the gradient
                                * the reference keeps commented out at src/color.rs:18-20 instead of black (:21) */
/* Two more opt-in extensions, off by default; the default path reproduces the reference's quirks. */
#define RTMI_FLAG_FACE_FORWARD 128u /* Lambertian / Metal / Isotropic scatter and Metal's absorption test use the
                                * normal turned against the incoming ray (n := -n when d.n > 0).  The reference never
                                * turns it (src/sphere.rs:50 outward, src/rect.rs:58-59 always +e_k), so e.g. the three
                                * min-side faces of a Cube (src/cube.rs:21-74) scatter into the box.  Dielectric keeps
                                * the geometric normal: it resolves the side itself (src/material.rs:106-114). */
#define RTMI_FLAG_UV_BOOK 4096u /* get_sphere_uv with v = (theta + pi/2) / pi (the book's formula) instead of the
                                * reference's FRAC_2_PI = 2/pi (src/sphere.rs:13), i.e. v in [0,1] */
#define RTMI_FLAG_PROGRESSIVE 16384u /* opt-in (progressive output; the reference's stand-in is the
                                * sleeping bar of src/progressbar.rs:6-58): after EVERY pass of the sample range the framebuffer
                                * holds the image of the samples rendered so far — their mean in sample order, quantised like
                                * the final image, i.e. exactly the image of a render with ns = samples so far — instead of only
                                * after the last one.  A render runs in passes when sample_buffer_bytes is smaller than
                                * RTMI_SAMPLE_SLOT_BYTES x pixels x ns (a budget of K samples' worth gives passes of K samples).  With a progress
                                * callback on rtmi_render, rtmi_partial_image() fetches that image from inside the callback. */
#define RTMI_FLAG_TEST_OVERFLOW 8192u /* test knob: the cooperative kernel reports a traversal-pool overflow although
                                * none happened, to exercise the error path (results of that call are poisoned) */
typedef struct {
    uint32_t nx, ny, ns; /* create_image(ny, nx, ns, ..) */
    uint32_t max_depth;  /* 50 (color.rs:9) */
    float t_min;         /* 0.001 (color.rs:7) */
    uint32_t flags;
    uint64_t seed;
    uint32_t tile_rank, tile_world; /* this call renders tiles t with t % tile_world == tile_rank */
    uint32_t spp_chunks;            /* sample chunks per tile (one wavefront each); 0 = choose automatically */
    uint32_t shade_threshold;       /* two-phase kernel: lanes holding a hit before shading starts (0 = default 40) */
    uint64_t path_sig;              /* RTMI_FLAG_PATH_SIG: DEVICE address of rtmi_local_tiles()*64 uint64 (else 0) */
    uint64_t prof;                  /* RTMI_FLAG_PROFILE: DEVICE address of 64 uint64 counters (else 0) */
    uint64_t sample_buffer_bytes;   /* budget of the per-sample radiance buffer (RTMI_SAMPLE_SLOT_BYTES per pixel sample of this rank);
                                     * 0 = default (all ns samples when they fit 45 GiB and 3/4 of the free HBM);
                                     * a smaller budget renders the sample range in passes — same result */
    /* Progress callback (replaces the reference's stand-alone bar, src/progressbar.rs:6-58, which only sleeps): the
     * blocking entry points (rtmi_render, rtmi_render_multi) call it from the calling thread about every 50 ms while
     * the kernels run, and once with done == total at the end.  `done`/`total` count work units (8x8 tile x sample
     * chunk) handed out by the device's unit queue.  A non-zero return value is remembered and the call returns
     * RTMI_ERR_CANCELLED after the running launch (launches are not pre-empted).  0 = no callback. */
    uint64_t progress_fn;           /* rtmi_progress_fn cast to an integer, or 0 */
    uint64_t progress_user;         /* passed back as `user` */
} rtmi_render_params;
typedef int (*rtmi_progress_fn)(uint64_t done, uint64_t total, void *user);

/* Path signature (test/validation aid): for every hit query of every sample that finds a hit,
 * mix(bits of the fp32 hit distance t, bounce index) is added (wrapping uint64) to the pixel's
 * signature.  It pins the whole geometric path sequence, so two implementations can be compared
 * bit-for-bit even on scenes whose radiance is identically zero.
 *   mix(x,k): x ^= (k+1)*0x9E3779B9; x ^= x>>16; x *= 0x7FEB352D; x ^= x>>15; x *= 0x846CA68B; x ^= x>>16 */

/* One framebuffer texel: mean linear radiance (before gamma) and the quantised
 * ir,ig,ib of tests/test.rs:71-78 packed as r | g<<8 | b<<16.  Bit 31 of rgb8 (RTMI_TEXEL_POISON) marks a texel of
 * a launch whose cooperative traversal pool overflowed (results invalid; r,g,b are NaN): rtmi_untile refuses it. */
#define RTMI_TEXEL_POISON 0x80000000u
typedef struct {
    float r, g, b;
    uint32_t rgb8;
} rtmi_texel; /* 16 B */

typedef struct {
    double kernel_ms;  /* render + resolve kernels, from HIP events on the launch stream */
    double render_ms;  /* the render kernel alone */
    uint64_t samples;  /* camera paths traced by this call */
    uint32_t tiles, chunks, blocks;
    uint32_t kernel;   /* which render kernel ran (diagnostics; results never depend on it): RTMI_KERNEL_* */
} rtmi_stats;
enum { RTMI_KERNEL_PERLANE = 0, RTMI_KERNEL_WAVE_COOP = 1, RTMI_KERNEL_ASYNC = 2, RTMI_KERNEL_BLOCK_COOP = 3 };

typedef struct rtmi_scene rtmi_scene;

int rtmi_device_count(void);
const char *rtmi_last_error(void);
/* 16 hex digits identifying the device code of this library: a hash of its kernel sources, this header and the compile
 * flags, taken by the in-tree build (raytracing_rust_amd/build.py).  Profiles committed under profiles/ carry the hash of
 * the build they were taken on; a reader can tell whether counters and library belong together.  "unknown" for a library
 * built some other way. */
const char *rtmi_build_hash(void);

/* Copies the description to `device` (hipMemcpy). */
int rtmi_scene_create(const rtmi_scene_desc *desc, int device, rtmi_scene **out);
/* Frees the handle.  Its per-sample radiance buffer (the one large allocation, up to 45 GiB) is PARKED instead of freed:
 * one buffer per device outlives its handle, and the next handle on that device (also inside rtmi_multi_* and the
 * one-shot rtmi_render_multi) takes it over when it is large enough — hosts that render one image per handle, the
 * reference's usage model (tests/test.rs:802-838), then do not pay a multi-GB hipMalloc per image, which on this
 * stack now and then takes seconds right after a hipFree of the same size.  rtmi_release_cached()
 * returns the parked memory of all devices to the system. */
void rtmi_scene_destroy(rtmi_scene *scene);
void rtmi_release_cached(void);

/* Number of tiles / texels in the tile-packed local framebuffer of this call:
 * texel index = local_tile * 64 + (ly * 8 + lx), local_tile = tile / tile_world,
 * tile = ty * tiles_x + tx counted from the TOP-left tile. */
uint32_t rtmi_local_tiles(const rtmi_render_params *p);

/* Optional: allocate everything a later rtmi_render_device/rtmi_render call with the same params needs (the
 * per-sample radiance buffer, RTMI_SAMPLE_SLOT_BYTES x local pixels x samples per pass, and the f64 sums), so that the first
 * render call does not pay for the allocation.  Idempotent; render calls allocate on demand anyway. */
int rtmi_render_prepare(rtmi_scene *scene, const rtmi_render_params *params);

/* Enqueues the render on `stream` (a hipStream_t, may be NULL) and writes
 * rtmi_local_tiles()*64 texels to the DEVICE buffer d_texels.  Does not synchronise
 * unless `stats` is non-NULL (then it waits for the kernels to fill kernel_ms, and reports a traversal-pool
 * overflow as RTMI_ERR_DEVICE).  Without `stats` the call is asynchronous: an overflow poisons the texels of
 * that launch (RTMI_TEXEL_POISON) and is reported by the next rtmi_scene_status(). */
int rtmi_render_device(rtmi_scene *scene, const rtmi_camera *cam, const rtmi_render_params *p, void *d_texels,
                       void *stream, rtmi_stats *stats);

/* Waits for the scene's device and returns RTMI_ERR_DEVICE if any render launch since the last call (or since
 * a render call that reported it) overflowed its cooperative traversal pool; the condition is cleared by the
 * report.  *overflows (optional) receives the number of wavefronts that saw one.  Cannot happen by the
 * depth-first bound of the pool; this is the loud end of that argument for the asynchronous entry point. */
int rtmi_scene_status(rtmi_scene *scene, uint32_t *overflows);

/* ---- several GPUs of this process behind one handle ("rtmi_scene_create copies ... to each selected
 * device", multi-GPU internal to the render call) --------------------------------------------------------------------
 * rtmi_multi_create uploads the description to every listed device (one host thread per device, so the uploads and
 * the derived-record building overlap), makes the RCCL communicators when the listed devices are distinct, and
 * returns a handle that keeps everything between calls: the device scenes, the per-device tile-packed framebuffers,
 * the gathered framebuffer on devices[0] with its pinned host mirror, and — after the first render (or
 * rtmi_multi_prepare) of a given size — the per-sample radiance buffers.  rtmi_multi_render then costs what the
 * kernels, ONE gather and the un-tiling cost; it is what a host's Camera::render binds for create_image
 * (tests/test.rs:55-85) with a device list.  devices[i] renders the tiles t with t % n_devices == i on its own stream,
 * all devices concurrently; the gather onto devices[0] is one ncclGather over xGMI (rccl.h) when the listed devices
 * are distinct, plain device-to-device copies when a device is listed more than once (single-GPU rehearsal: RCCL
 * cannot put two ranks on one device).  The image is bit-identical to rtmi_render's for any device list.
 * A communicator set belongs to ONE handle while that handle lives (RCCL lets one thread at a time enqueue on a
 * communicator): it is checked out of a per-device-list pool at create — made with ncclCommInitAll when the pool has
 * none free — and handed back at destroy, so two live handles on the same device list render concurrently on their
 * own communicators and one-shot calls reuse a set instead of initialising one per image.
 * A one-entry device list needs no exchange and loads no RCCL; with RTMI_FORCE_RCCL=1 in the environment at create
 * it gets a one-rank communicator as well and the render runs the same grouped ncclGather (the way to execute the
 * whole collective path on a single GPU).  rtmi_multi_collective() says which exchange a handle uses.
 * stats (optional): kernel_ms / render_ms = the slowest device's, samples = all devices'.
 * params: tile_rank / tile_world must be 0 / 1; PATH_SIG and PROFILE are single-device diagnostics (rejected). */
typedef struct rtmi_multi rtmi_multi;
enum { RTMI_COLLECTIVE_NONE = 0,      /* one device, framebuffer copied on the device */
       RTMI_COLLECTIVE_PEER_COPY = 1, /* a device listed more than once: hipMemcpyPeerAsync per rank */
       RTMI_COLLECTIVE_RCCL = 2 };    /* one grouped ncclGather onto devices[0] */
/* which exchange rtmi_multi_render of this handle performs (RTMI_COLLECTIVE_*; -1 for NULL) */
int rtmi_multi_collective(const rtmi_multi *m);
int rtmi_multi_create(const rtmi_scene_desc *desc, const int *devices, uint32_t n_devices, rtmi_multi **out);
/* optional: allocate the per-sample buffers for renders with these params on every device now (idempotent) */
int rtmi_multi_prepare(rtmi_multi *m, const rtmi_render_params *params);
int rtmi_multi_render(rtmi_multi *m, const rtmi_camera *cam, const rtmi_render_params *params, float *out_linear_rgb,
                      uint8_t *out_rgb8, rtmi_stats *stats);
void rtmi_multi_destroy(rtmi_multi *m);

/* One-shot form: rtmi_multi_create + rtmi_multi_render + rtmi_multi_destroy (pays the uploads and the allocations on
 * every call; hosts that render more than once keep an rtmi_multi).
 * Whole image on several GPUs of this process — the triple loop of create_image (tests/test.rs:62-79), which the
 * reference runs on one thread, split over devices: the scene description is uploaded to every
 * listed device, device i renders the tiles t with t % n_devices == i on its own stream (all devices run
 * concurrently), the tile-packed framebuffers are gathered on devices[0] — one ncclGather over xGMI
 * (rccl.h ncclGather) when the listed devices are distinct, plain device-to-device copies when a device is
 * listed more than once (single-GPU rehearsal: RCCL cannot put two ranks on one device) — and un-tiled into the
 * host buffers exactly like rtmi_render.  The image is bit-identical to rtmi_render's for any device list.
 * stats (optional): kernel_ms / render_ms = the slowest device's, samples = all devices'. */
int rtmi_render_multi(const rtmi_scene_desc *desc, const int *devices, uint32_t n_devices, const rtmi_camera *cam,
                      const rtmi_render_params *p, float *out_linear_rgb, uint8_t *out_rgb8, rtmi_stats *stats);

/* Blocking whole-image render into host buffers (tile_world must be 1):
 * out_linear_rgb: ny*nx*3 floats, row 0 = top row (reference j = ny-1); may be NULL
 * out_rgb8:       ny*nx*3 bytes, same order; may be NULL */
int rtmi_render(rtmi_scene *scene, const rtmi_camera *cam, const rtmi_render_params *p, float *out_linear_rgb,
                uint8_t *out_rgb8, uint64_t *out_path_sig /* ny*nx, optional: sets RTMI_FLAG_PATH_SIG */,
                rtmi_stats *stats);

/* RTMI_FLAG_PROGRESSIVE: the image of the passes finished so far of the rtmi_render call running on `scene`.  To be
 * called ONLY from inside that call's progress callback (same thread; it takes no lock and reads the handle's
 * framebuffer through its copy stream).  *spp_done = samples per pixel the image holds (0: no pass has finished yet,
 * the outputs are left untouched); pixels may come from two consecutive passes when a pass ends during the copy.
 * `p` = the parameters of the running call (image size).  out_linear_rgb / out_rgb8 as for rtmi_render; may be NULL. */
int rtmi_partial_image(rtmi_scene *scene, const rtmi_render_params *p, float *out_linear_rgb, uint8_t *out_rgb8,
                       uint32_t *spp_done);

/* Host-side un-tiling of `tile_world` gathered local buffers (rank-major, each
 * rtmi_local_tiles(rank 0)*64 texels, i.e. padded to the largest rank) into raster order. */
int rtmi_untile(const rtmi_render_params *p, const rtmi_texel *gathered, float *out_linear_rgb, uint8_t *out_rgb8);

/* The P3 text of create_image (tests/test.rs:59,79): "P3\n{nx} {ny}\n255\n" then one
 * "{ir} {ig} {ib}\n" line per pixel.  Returns the bytes written, or the capacity needed
 * when buf is NULL or cap is too small. */
size_t rtmi_ppm_p3(uint32_t nx, uint32_t ny, const uint8_t *rgb8, char *buf, size_t cap);

/* The same image written straight to a file in 1 MiB pieces, without the whole-image string of create_image
 * (tests/test.rs:58 builds ~12 B per pixel in memory, :560 writes it): format 3 = the P3 text above, byte for byte;
 * format 6 = binary PPM ("P6\n{nx} {ny}\n255\n" + ny*nx*3 bytes, 4x smaller).  Returns RTMI_OK or an error code. */
int rtmi_write_ppm(const char *path, uint32_t nx, uint32_t ny, const uint8_t *rgb8, int format);

/* Test hooks: evaluate pieces of the arithmetic contract ON THE DEVICE so that parity
 * tests can compare them bit-for-bit with a host evaluation of rtmi_math.h / Philox.
 * op: 0 rtmi_sinf(x), 1 rtmi_logf(x), 2 rtmi_atan2f(x,y), 3 rtmi_asinf(x), 4 x/y,
 * 5 sqrt(x), 6 rtmi_u01(bits of x).  ctr: n*4 words, key: n*2 words, out: n*4 words. */
int rtmi_probe_math(int op, const float *x, const float *y, float *out, uint32_t n);
int rtmi_probe_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out, uint32_t n);
/* Instance transforms as the render kernels apply them (src/traslate.rs:18-24, src/rotate.rs:85-113): for each of
 * the n inputs (a, b: 3 floats each) out[13*i..] = world->object ray (origin a, direction b: 6 floats), object->world
 * hit record (point a, normal b: 6 floats), and 1.0 when a rotation was applied.  xforms: outermost wrapper first. */
int rtmi_probe_xform(const rtmi_xform *xforms, uint32_t count, const float *a, const float *b, float *out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
