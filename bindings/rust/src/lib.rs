//! Safe wrapper over librtmi.so for the reference crate: `Scene` owns the device copy of a lowered world,
//! `Scene::render` is the triple loop of `create_image` (tests/test.rs:62-79) + `color` (src/color.rs:6-23)
//! on the MI355X, `ppm_p3` is the text `create_image` returns.  The lowering of the reference's object graph
//! (`Hittable::lower`) fills a `FlatScene`; it lives in the reference crate because it needs
//! the concrete types.  UNVERIFIED SOURCE (no Rust toolchain in the build image) — the C++ host mirror
//! (raytracing_rust_amd/host/rt_host.cpp) is the built and tested implementation of the same calls.
pub mod sys;

use std::ffi::CStr;
use std::os::raw::c_char;
use sys::*;

#[derive(Debug)]
pub struct RtmiError {
    pub code: i32,
    pub message: String,
}

fn check(rc: i32) -> Result<(), RtmiError> {
    if rc == 0 {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(rtmi_last_error()) }.to_string_lossy().into_owned();
    Err(RtmiError { code: rc, message })
}

/// The flat arrays of include/rtmi.h, owned on the host until `Scene::upload`.
#[derive(Default)]
pub struct FlatScene {
    pub items: Vec<RtmiItem>,
    pub prim_a: Vec<f32>,
    pub prim_b: Vec<f32>,
    pub prim_meta: Vec<RtmiPrimMeta>,
    /// optional: 8 floats per primitive (include/rtmi.h, prim_gate); empty = none
    pub prim_gate: Vec<f32>,
    pub alt_max_depth: u32,
    pub alt_nodes: Vec<RtmiBvh4Node>,
    pub nodes: Vec<RtmiBvhNode>,
    pub xforms: Vec<RtmiXform>,
    pub materials: Vec<RtmiMaterial>,
    pub textures: Vec<RtmiTexture>,
    pub perlin: Vec<RtmiPerlin>,
    pub images: Vec<RtmiImage>,
    pub image_data: Vec<u8>,
    pub max_bvh_depth: u32,
    pub bvh_time_lo: f32,
    pub bvh_time_hi: f32,
}

impl FlatScene {
    pub fn new() -> Self {
        FlatScene { bvh_time_lo: f32::MIN, bvh_time_hi: f32::MAX, ..Default::default() }
    }
    pub fn desc(&self) -> RtmiSceneDesc {
        RtmiSceneDesc {
            abi_version: RTMI_ABI_VERSION,
            n_items: self.items.len() as u32,
            items: self.items.as_ptr(),
            n_prims: self.prim_meta.len() as u32,
            prim_a: self.prim_a.as_ptr(),
            prim_b: self.prim_b.as_ptr(),
            prim_meta: self.prim_meta.as_ptr(),
            prim_gate: if self.prim_gate.is_empty() { std::ptr::null() } else { self.prim_gate.as_ptr() },
            alt_max_depth: self.alt_max_depth,
            n_alt_nodes: self.alt_nodes.len() as u32,
            alt_nodes: self.alt_nodes.as_ptr(),
            n_nodes: self.nodes.len() as u32,
            nodes: self.nodes.as_ptr(),
            n_xforms: self.xforms.len() as u32,
            xforms: self.xforms.as_ptr(),
            n_materials: self.materials.len() as u32,
            materials: self.materials.as_ptr(),
            n_textures: self.textures.len() as u32,
            textures: self.textures.as_ptr(),
            n_perlin: self.perlin.len() as u32,
            perlin: self.perlin.as_ptr(),
            n_images: self.images.len() as u32,
            images: self.images.as_ptr(),
            image_data: self.image_data.as_ptr(),
            image_bytes: self.image_data.len() as u64,
            max_bvh_depth: self.max_bvh_depth,
            bvh_time_lo: self.bvh_time_lo,
            bvh_time_hi: self.bvh_time_hi,
        }
    }
}

pub struct Image {
    pub nx: usize,
    pub ny: usize,
    /// mean linear radiance, row 0 = top row (reference j = ny-1), 3 floats per pixel
    pub linear: Vec<f32>,
    /// the ir, ig, ib of tests/test.rs:76-78
    pub rgb8: Vec<u8>,
    pub stats: RtmiStats,
}

impl Image {
    /// The P3 text of create_image (tests/test.rs:59,79).
    pub fn to_ppm(&self) -> String {
        let need = unsafe { rtmi_ppm_p3(self.nx as u32, self.ny as u32, self.rgb8.as_ptr(), std::ptr::null_mut(), 0) };
        let mut buf = vec![0u8; need];
        let n = unsafe {
            rtmi_ppm_p3(self.nx as u32, self.ny as u32, self.rgb8.as_ptr(), buf.as_mut_ptr() as *mut c_char, buf.len())
        };
        buf.truncate(n);
        String::from_utf8(buf).expect("P3 text is ASCII")
    }
}

pub fn default_params(nx: usize, ny: usize, ns: usize, seed: u64) -> RtmiRenderParams {
    RtmiRenderParams {
        nx: nx as u32,
        ny: ny as u32,
        ns: ns as u32,
        max_depth: 50, // color.rs:9
        t_min: 0.001,  // color.rs:7
        flags: RTMI_FLAG_FAST_CULL,
        seed,
        tile_rank: 0,
        tile_world: 1,
        spp_chunks: 0,
        shade_threshold: 0,
        path_sig: 0,
        prof: 0,
        sample_buffer_bytes: 0,
        progress_fn: 0,
        progress_user: 0,
    }
}

/// A world resident in HBM.
pub struct Scene {
    raw: *mut RtmiScene,
}

impl Scene {
    pub fn upload(flat: &FlatScene, device: i32) -> Result<Scene, RtmiError> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { rtmi_scene_create(&flat.desc(), device, &mut raw) })?;
        Ok(Scene { raw })
    }
    /// Blocking whole-image render (tile_world = 1).
    pub fn render(&mut self, cam: &RtmiCamera, p: &RtmiRenderParams) -> Result<Image, RtmiError> {
        let (nx, ny) = (p.nx as usize, p.ny as usize);
        let mut img = Image { nx, ny, linear: vec![0.0; nx * ny * 3], rgb8: vec![0; nx * ny * 3], stats: RtmiStats::default() };
        check(unsafe {
            rtmi_render(self.raw, cam, p, img.linear.as_mut_ptr(), img.rgb8.as_mut_ptr(), std::ptr::null_mut(), &mut img.stats)
        })?;
        Ok(img)
    }
}

impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { rtmi_scene_destroy(self.raw) }
    }
}
