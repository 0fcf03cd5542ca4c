//! Raw FFI of include/rtmi.h (ABI version 5): one declaration per entry point, one `#[repr(C)]` struct per
//! C struct, same field order.  UNVERIFIED SOURCE: the build image has no Rust toolchain; the layouts are
//! kept in sync with the tested ctypes binding (raytracing_rust_amd/abi.py) by tests/test_rust_binding_source.py.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RTMI_ABI_VERSION: u32 = 7;
pub const RTMI_FLAG_FAST_CULL: u32 = 1;
pub const RTMI_FLAG_PATH_SIG: u32 = 2;
pub const RTMI_FLAG_PROFILE: u32 = 4;
pub const RTMI_FLAG_SYNC: u32 = 8;
pub const RTMI_FLAG_ASYNC: u32 = 16;
pub const RTMI_FLAG_SKY: u32 = 32;
pub const RTMI_FLAG_REF_TREE: u32 = 64;
pub const RTMI_FLAG_BLOCK_COOP: u32 = 32768;
pub const RTMI_SAMPLE_SLOT_BYTES: u32 = 12;
pub const RTMI_FLAG_FACE_FORWARD: u32 = 128;
pub const RTMI_FLAG_UV_BOOK: u32 = 4096;
/// opt-in: after every pass the framebuffer holds the image of the samples so far (rtmi_partial_image)
pub const RTMI_FLAG_PROGRESSIVE: u32 = 16384;
pub const RTMI_OK: i32 = 0;
pub const RTMI_ERR_INVALID: i32 = 1;
pub const RTMI_ERR_UNSUPPORTED: i32 = 2;
pub const RTMI_ERR_DEVICE: i32 = 3;
pub const RTMI_ERR_NOMEM: i32 = 4;
pub const RTMI_ERR_CANCELLED: i32 = 5;
pub const RTMI_TEXEL_POISON: u32 = 0x8000_0000;
pub const RTMI_MAX_BVH_DEPTH: u32 = 24;
pub const RTMI_TILE: u32 = 8;
pub const RTMI_TEX_SOLID: i32 = 0;
pub const RTMI_TEX_CHECKER: i32 = 1;
pub const RTMI_TEX_NOISE: i32 = 2;
pub const RTMI_TEX_IMAGE: i32 = 3;
pub const RTMI_MAT_LAMBERTIAN: i32 = 0;
pub const RTMI_MAT_METAL: i32 = 1;
pub const RTMI_MAT_DIELECTRIC: i32 = 2;
pub const RTMI_MAT_DIFFUSE_LIGHT: i32 = 3;
pub const RTMI_MAT_ISOTROPIC: i32 = 4;
pub const RTMI_MATFLAG_NEEDS_UV: u32 = 1;
pub const RTMI_PRIM_SPHERE: i32 = 0;
pub const RTMI_PRIM_MSPHERE: i32 = 1;
pub const RTMI_PRIM_RECT: i32 = 2;
pub const RTMI_PRIM_CUBE: i32 = 3;
pub const RTMI_PRIMFLAG_FLIP: u32 = 1;
pub const RTMI_PRIMFLAG_PLANE_SHIFT: u32 = 8;
/// instanced primitive (rtmi.h): bits 4..7 = number of its own transforms, bits 12..31 = index of the first in xforms
pub const RTMI_PRIMFLAG_XF_COUNT_SHIFT: u32 = 4;
pub const RTMI_PRIMFLAG_XF_FIRST_SHIFT: u32 = 12;
pub const RTMI_PRIM_XF_MAX: u32 = 15;
pub const RTMI_XF_TRANSLATE: i32 = 0;
pub const RTMI_XF_ROTATE_X: i32 = 1;
pub const RTMI_XF_ROTATE_Y: i32 = 2;
pub const RTMI_XF_ROTATE_Z: i32 = 3;
/// not transforms: the two records behind the chain of a DEFERRED BVH item hold its gate box
pub const RTMI_XF_GATE_MIN: i32 = 4;
pub const RTMI_XF_GATE_MAX: i32 = 5;
/// not a transform: x = -(1/density) of the inner medium of a nested pair, behind the chain (and the gate records)
pub const RTMI_XF_INNER_MEDIUM: i32 = 6;
pub const RTMI_ITEM_LIST: i32 = 0;
pub const RTMI_ITEM_BVH: i32 = 1;
pub const RTMI_ITEMFLAG_FLIP: u32 = 1;
pub const RTMI_ITEMFLAG_MEDIUM: u32 = 2;
/// MEDIUM items: bits 8..11 = number of the item's first transforms that wrap the ConstantMedium itself
pub const RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT: u32 = 8;
/// a ConstantMedium that was a child of a BVHNode follows its BVH item as a DEFERRED item (rtmi.h): SAVE_T0 on the BVH item
/// (or on the first deferred one when the BVH holds nothing but media) remembers the closest hit before it
pub const RTMI_ITEMFLAG_SAVE_T0: u32 = 4;
pub const RTMI_ITEMFLAG_DEFERRED: u32 = 8;
/// a ConstantMedium whose boundary is a ConstantMedium: the inner density travels in an RTMI_XF_INNER_MEDIUM record
pub const RTMI_ITEMFLAG_NESTED_MEDIUM: u32 = 16;
/// a HittableList with media among its members as a BVH child: a group of DEFERRED member items and a terminator (rtmi.h)
pub const RTMI_ITEMFLAG_LISTSCAN_BEGIN: u32 = 32;
pub const RTMI_ITEMFLAG_LISTSCAN_MEMBER: u32 = 64;
pub const RTMI_ITEMFLAG_LISTSCAN_END: u32 = 128;
/// DEFERRED items: bits 12..15 = number of leading transforms that belong to the enclosing BVH item
pub const RTMI_ITEMFLAG_GATE_OUTER_SHIFT: u32 = 12;
pub const RTMI_NO_CHILD: i32 = 0x7fff_ffff;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiTexture {
    pub kind: i32,
    pub i0: i32,
    pub i1: i32,
    pub pad: i32,
    pub f0: f32,
    pub f1: f32,
    pub f2: f32,
    pub f3: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiPerlin {
    pub ranvec: [f32; 1024],
    pub perm: [i32; 768],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiImage {
    pub offset: u64,
    pub nx: u32,
    pub ny: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiMaterial {
    pub kind: i32,
    pub tex: i32,
    pub param: f32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiPrimMeta {
    pub material: i32,
    pub flags: u32,
    pub inv_dt: f32,
    pub r#type: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiBvhNode {
    pub lmin: [f32; 3],
    pub lmax: [f32; 3],
    pub rmin: [f32; 3],
    pub rmax: [f32; 3],
    pub left: i32,
    pub right: i32,
    pub pad: [i32; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiBvh4Node {
    pub minx: [f32; 4],
    pub miny: [f32; 4],
    pub minz: [f32; 4],
    pub maxx: [f32; 4],
    pub maxy: [f32; 4],
    pub maxz: [f32; 4],
    pub child: [i32; 4],
    pub pad: [i32; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiXform {
    pub kind: i32,
    pub x: f32,
    pub y: f32,
    pub z: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiItem {
    pub kind: i32,
    pub first: i32,
    pub count: i32,
    pub flags: u32,
    pub xform_first: i32,
    pub xform_count: i32,
    pub medium_material: i32,
    pub neg_inv_density: f32,
    pub root_min: [f32; 3],
    pub root_max: [f32; 3],
    pub scale: f32,
    pub alt_first: i32,
}

#[repr(C)]
pub struct RtmiSceneDesc {
    pub abi_version: u32,
    pub n_items: u32,
    pub items: *const RtmiItem,
    pub n_prims: u32,
    pub prim_a: *const f32,
    pub prim_b: *const f32,
    pub prim_meta: *const RtmiPrimMeta,
    pub prim_gate: *const f32,
    pub alt_max_depth: u32,
    pub n_alt_nodes: u32,
    pub alt_nodes: *const RtmiBvh4Node,
    pub n_nodes: u32,
    pub nodes: *const RtmiBvhNode,
    pub n_xforms: u32,
    pub xforms: *const RtmiXform,
    pub n_materials: u32,
    pub materials: *const RtmiMaterial,
    pub n_textures: u32,
    pub textures: *const RtmiTexture,
    pub n_perlin: u32,
    pub perlin: *const RtmiPerlin,
    pub n_images: u32,
    pub images: *const RtmiImage,
    pub image_data: *const u8,
    pub image_bytes: u64,
    pub max_bvh_depth: u32,
    pub bvh_time_lo: f32,
    pub bvh_time_hi: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiCamera {
    pub origin: [f32; 3],
    pub lower_left_corner: [f32; 3],
    pub horizontal: [f32; 3],
    pub vertical: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub time0: f32,
    pub time1: f32,
    pub lens_radius: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiRenderParams {
    pub nx: u32,
    pub ny: u32,
    pub ns: u32,
    pub max_depth: u32,
    pub t_min: f32,
    pub flags: u32,
    pub seed: u64,
    pub tile_rank: u32,
    pub tile_world: u32,
    pub spp_chunks: u32,
    pub shade_threshold: u32,
    pub path_sig: u64,
    pub prof: u64,
    pub sample_buffer_bytes: u64,
    /// `RtmiProgressFn` cast to an integer, or 0 (called by the blocking entry points about every 50 ms)
    pub progress_fn: u64,
    pub progress_user: u64,
}
/// `int (*)(uint64_t done, uint64_t total, void *user)`; non-zero return = cancel (RTMI_ERR_CANCELLED)
pub type RtmiProgressFn = unsafe extern "C" fn(done: u64, total: u64, user: *mut c_void) -> c_int;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtmiTexel {
    pub r: f32,
    pub g: f32,
    pub b: f32,
    pub rgb8: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtmiStats {
    pub kernel_ms: f64,
    pub render_ms: f64,
    pub samples: u64,
    pub tiles: u32,
    pub chunks: u32,
    pub blocks: u32,
    pub kernel: u32, // RTMI_KERNEL_*: which render kernel ran (diagnostics)
}

#[repr(C)]
pub struct RtmiScene {
    _private: [u8; 0],
}

/// opaque persistent multi-device handle (rtmi.h: rtmi_multi)
#[repr(C)]
pub struct RtmiMulti {
    _private: [u8; 0],
}

extern "C" {
    pub fn rtmi_device_count() -> c_int;
    pub fn rtmi_last_error() -> *const c_char;
    /// 16 hex digits: hash of the kernel sources, rtmi.h and the compile flags this library was built from
    pub fn rtmi_build_hash() -> *const c_char;
    pub fn rtmi_scene_create(desc: *const RtmiSceneDesc, device: c_int, out: *mut *mut RtmiScene) -> c_int;
    pub fn rtmi_scene_destroy(scene: *mut RtmiScene);
    /// frees the per-sample buffers that destroyed handles left parked (one per device) for their successors
    pub fn rtmi_release_cached();
    pub fn rtmi_local_tiles(p: *const RtmiRenderParams) -> u32;
    pub fn rtmi_render_prepare(scene: *mut RtmiScene, p: *const RtmiRenderParams) -> c_int;
    pub fn rtmi_render_device(
        scene: *mut RtmiScene,
        cam: *const RtmiCamera,
        p: *const RtmiRenderParams,
        d_texels: *mut c_void,
        stream: *mut c_void,
        stats: *mut RtmiStats,
    ) -> c_int;
    pub fn rtmi_scene_status(scene: *mut RtmiScene, overflows: *mut u32) -> c_int;
    pub fn rtmi_render_multi(
        desc: *const RtmiSceneDesc,
        devices: *const c_int,
        n_devices: u32,
        cam: *const RtmiCamera,
        p: *const RtmiRenderParams,
        out_linear_rgb: *mut f32,
        out_rgb8: *mut u8,
        stats: *mut RtmiStats,
    ) -> c_int;
    pub fn rtmi_multi_create(desc: *const RtmiSceneDesc, devices: *const c_int, n_devices: u32, out: *mut *mut RtmiMulti) -> c_int;
    pub fn rtmi_multi_prepare(m: *mut RtmiMulti, p: *const RtmiRenderParams) -> c_int;
    pub fn rtmi_multi_render(
        m: *mut RtmiMulti,
        cam: *const RtmiCamera,
        p: *const RtmiRenderParams,
        out_linear_rgb: *mut f32,
        out_rgb8: *mut u8,
        stats: *mut RtmiStats,
    ) -> c_int;
    pub fn rtmi_multi_destroy(m: *mut RtmiMulti);
    /// RTMI_COLLECTIVE_*: 0 none (one device), 1 peer copies (a device listed twice), 2 one grouped ncclGather
    pub fn rtmi_multi_collective(m: *const RtmiMulti) -> c_int;
    pub fn rtmi_render(
        scene: *mut RtmiScene,
        cam: *const RtmiCamera,
        p: *const RtmiRenderParams,
        out_linear_rgb: *mut f32,
        out_rgb8: *mut u8,
        out_path_sig: *mut u64,
        stats: *mut RtmiStats,
    ) -> c_int;
    /// RTMI_FLAG_PROGRESSIVE: only from inside the progress callback of the rtmi_render call running on `scene`
    pub fn rtmi_partial_image(
        scene: *mut RtmiScene,
        p: *const RtmiRenderParams,
        out_linear_rgb: *mut f32,
        out_rgb8: *mut u8,
        spp_done: *mut u32,
    ) -> c_int;
    pub fn rtmi_untile(
        p: *const RtmiRenderParams,
        gathered: *const RtmiTexel,
        out_linear_rgb: *mut f32,
        out_rgb8: *mut u8,
    ) -> c_int;
    pub fn rtmi_ppm_p3(nx: u32, ny: u32, rgb8: *const u8, buf: *mut c_char, cap: usize) -> usize;
    /// streaming writer: format 3 = the P3 text of `create_image`, 6 = binary P6
    pub fn rtmi_write_ppm(path: *const c_char, nx: u32, ny: u32, rgb8: *const u8, format: c_int) -> c_int;
    pub fn rtmi_probe_math(op: c_int, x: *const f32, y: *const f32, out: *mut f32, n: u32) -> c_int;
    pub fn rtmi_probe_philox(ctr: *const u32, key: *const u32, out: *mut u32, n: u32) -> c_int;
    pub fn rtmi_probe_xform(
        xforms: *const RtmiXform,
        count: u32,
        a: *const f32,
        b: *const f32,
        out: *mut f32,
        n: u32,
    ) -> c_int;
}
