// rtmi_types.hpp — vector type and the device-side views of scene, camera and render parameters.
// Part of the single translation unit rtmi_device.hip (device code is header-only so that every
// kernel instantiation inlines the whole path); arithmetic contract as stated there.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

#include <cstdint>

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

// Device copy of one world-list entry (built by rtmi_scene_create): the item record followed by the first two
// transforms of its chain, so that neither the item loop (one scalar load sequence) nor the shading of a hit (one
// dependent fetch) waits for `xforms` after the item record has arrived — final_scene's sphere BVH sits under
// Traslate(Rotate(..)) and every one of its hits used to pay four serialised transform fetches.
struct DevItem {
    rtmi_item it;     // 64 B
    rtmi_xform x0, x1; // xforms[it.xform_first], [.. + 1] (zeros when the chain is shorter)
}; // 96 B
// shade_hit (rtmi_shade.hpp) fetches the four shading words of the item record and its two embedded transforms by
// byte offset, and the kernels load {x0, x1} as one pair from &items[i].x0: pin the layout those reads assume
static_assert(sizeof(rtmi_item) == 64 && sizeof(rtmi_xform) == 16 && sizeof(DevItem) == 96, "DevItem layout");
static_assert(offsetof(rtmi_item, flags) == 12 && offsetof(rtmi_item, medium_material) == 24, "item shading words at bytes 12..27");
static_assert(offsetof(DevItem, x0) == 64 && offsetof(DevItem, x1) == 80, "embedded transforms at bytes 64 and 80");

struct PrimRec { // head of a leaf record (DevScene::leaf_rec, 80 B apart)
    float4 A, B;
    rtmi_prim_meta M;
};
static_assert(sizeof(PrimRec) == 48, "leaf record head");

struct DevScene {
    const DevItem *items;
    const float4 *prim_a;
    const float4 *prim_b;
    const rtmi_prim_meta *meta;
    const float4 *gate;  // 2 x float4 per primitive: box of its parent BVHNode in the reference tree (or NULL); the
                         // kernels read it through leaf_rec
    const float4 *nodes; // 4 x float4 per rtmi_bvh_node
    const float4 *nodes4; // 8 x float4 per rtmi_bvh4_node (alternative trees)
    // shading records (device-side layout, built by rtmi_scene_create): 4 x float4 = {plane A, material record,
    // its first-level texture record (32 B)} per primitive / per material — the closest hit's normal, material and
    // texture arrive with ONE dependent fetch instead of the chain prim -> material -> texture
    const float4 *shade_prim;
    const float4 *shade_mat;
    // leaf records (device-side layout, every scene): 5 x float4 = {plane A, plane B, meta, gate min, gate max} per
    // primitive (gate: zeros without prim_gate), so a leaf visit of a gated tree computes one address and touches one or
    // two cache lines instead of five, and list scans and shading read a primitive through the same base pointer
    const float4 *leaf_rec;
    const rtmi_xform *xforms;
    const rtmi_material *mats;
    const rtmi_texture *texs;
    const rtmi_perlin *perlin;
    const rtmi_image *images;
    const uint8_t *image_data;
    uint32_t n_items;
    uint32_t has_prim_xf; // some primitive carries its own transform chain (instanced primitive, rtmi.h)
    uint32_t has_medium_outer; // some medium sits inside transforms of its item (RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT)
};

// one slot of the per-sample radiance buffer: three fp32, 12 bytes, written by one global_store_dwordx3 (a float4 slot
// carried a padding word: a quarter of the buffer and of the resolve kernel's reads)
struct Rad3 { float r, g, b; };
static_assert(sizeof(Rad3) == RTMI_SAMPLE_SLOT_BYTES, "per-sample slot size is part of rtmi.h");

struct DevCamera {
    F3 origin, llc, horizontal, vertical, u, v;
    float time0, time1, lens_radius;
};

struct DevParams {
    uint32_t nx, ny, ns, max_depth;
    float t_min;
    uint32_t key0, key1;
    uint32_t tile_rank, tile_world, tiles_x, ntiles_local;
    uint32_t nchunks, chunk_spp;               // sample chunks of this pass: chunk c = [pass_s0 + c*chunk_spp, ..)
    uint32_t pass_s0, pass_cnt, pass_stride;   // this launch renders samples [pass_s0, pass_s0 + pass_cnt)
    Rad3 *samples;                             // [local tile][pass_stride][64] radiance of every finished path (12 B)
    unsigned long long *path_sig;
    unsigned long long *prof;
    uint32_t stack_depth;
    uint32_t shade_threshold;
    uint32_t coop_cap;   // entries of the cooperative traversal's LDS pool
    uint32_t spill_cap;  // entries per wavefront of its global-memory extension
    uint2 *spill;
    unsigned int *status;
    unsigned int *queue; // next unit of the persistent wavefronts (zeroed before every launch)
    uint32_t sky; // RTMI_FLAG_SKY
    uint32_t use_alt; // cooperative kernel: walk the items' alternative trees, leaves accepted through their gate
    uint32_t ext;     // opt-in extensions / test knobs: RTMI_EXT_*
};
// device-only item flag (set by rtmi_scene_create, never part of the ABI): MEDIUM item whose boundary is one static
// sphere; root_min = its centre, root_max[0] = its radius
#define RTMI_ITEMFLAG_DEV_MEDIUM_SPHERE (1u << 16)
#define RTMI_COOP_PARK_WORDS 256u   /* cooperative kernel, INSD instantiations: 4 words per lane of group state in LDS */
#define RTMI_PARK_NONE 0x7fffffffu  /* ... no member of the list scan holds a hit */
#define RTMI_EXT_FACE_FORWARD 1u  // RTMI_FLAG_FACE_FORWARD
#define RTMI_EXT_UV_BOOK 2u       // RTMI_FLAG_UV_BOOK
#define RTMI_EXT_TEST_OVERFLOW 4u // RTMI_FLAG_TEST_OVERFLOW
// status words of a scene (device memory): [0] wavefronts of the CURRENT render call that overflowed their traversal
// pool (cleared at the start of every call), [1] unit counter of the persistent wavefronts (zero between passes),
// [2] overflows accumulated until rtmi_scene_status() reports them, [3] units of the finished passes of this call,
// [4] samples per pixel of the finished passes of this call (RTMI_FLAG_PROGRESSIVE: what the framebuffer holds)
#define RTMI_STATUS_WORDS 5
