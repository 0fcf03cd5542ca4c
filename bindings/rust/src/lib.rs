//! The Rust side of the drop-in boundary for the reference crate (DrStiev/raytracing_rust):
//!   * `sys`    raw FFI of include/rtmi.h (librtmi.so);
//!   * `desc`   a plain description of the reference's object graph — what the five-line `describe()` methods that
//!              INTEGRATION.md adds to the reference's types return (their fields are private to their modules);
//!   * `lower`  description -> flat scene, the function-by-function counterpart of the C++ SceneBuilder
//!              (raytracing_rust_amd/host/rt_host.cpp); golden dumps of that lowering: tests/golden/flat_*.bin.gz;
//!   * `Camera::render`, `create_image` — the triple loop of tests/test.rs:55-85 + `color` (src/color.rs:6-23) on the
//!     MI355X, one or several GPUs; `Image::to_ppm` is the text `create_image` returns;
//!   * `philox`, `scenes`, `dump` — seeded scene construction and the byte dump used to diff against the goldens.
//! UNVERIFIED SOURCE (no Rust toolchain in the build image) — the C++ host mirror is the built and tested
//! implementation of the same calls.
pub mod desc;
pub mod dump;
pub mod lower;
pub mod philox;
pub mod scenes;
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

    /// Streams the image to `path` without building the string: `binary` = false writes the P3 text of `to_ppm`
    /// (what tests/test.rs:560 writes), true the 4x smaller P6 form.
    pub fn write_ppm(&self, path: &str, binary: bool) -> Result<(), RtmiError> {
        let c = std::ffi::CString::new(path)
            .map_err(|_| RtmiError { code: RTMI_ERR_INVALID, message: "path contains a NUL byte".into() })?;
        let rc = unsafe { rtmi_write_ppm(c.as_ptr(), self.nx as u32, self.ny as u32, self.rgb8.as_ptr(), if binary { 6 } else { 3 }) };
        check(rc)
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

impl Scene {
    /// RTMI_FLAG_PROGRESSIVE: the image of the passes finished so far of the `render` call running on this scene — only
    /// from inside that call's progress callback.  Returns the samples per pixel the image holds (0 = nothing yet).
    pub fn partial_image(&mut self, p: &RtmiRenderParams, img: &mut Image) -> Result<u32, RtmiError> {
        let mut spp = 0u32;
        check(unsafe { rtmi_partial_image(self.raw, p, img.linear.as_mut_ptr(), img.rgb8.as_mut_ptr(), &mut spp) })?;
        Ok(spp)
    }
}

impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { rtmi_scene_destroy(self.raw) }
    }
}


/// A world resident on a list of GPUs of this process (rtmi_multi_create): lowered and uploaded once, then any number
/// of camera views rendered from it — a render costs the kernels, ONE gather and the un-tiling, not the uploads.
/// `Camera::render` with `RenderOptions::devices` is `DeviceScene::new(..)?.render(..)` for a single image.
pub struct DeviceScene {
    raw: *mut RtmiMulti,
}

impl DeviceScene {
    pub fn new(world: &std::rc::Rc<desc::HittableDesc>, devices: &[i32]) -> Result<DeviceScene, RtmiError> {
        let flat = lower::lower_world(world).map_err(|e| RtmiError { code: RTMI_ERR_UNSUPPORTED, message: format!("{:?}", e) })?;
        DeviceScene::upload(&flat, devices)
    }
    pub fn upload(flat: &FlatScene, devices: &[i32]) -> Result<DeviceScene, RtmiError> {
        let mut raw = std::ptr::null_mut();
        // the description is only borrowed for the call: rtmi_multi_create copies it to every listed device
        check(unsafe { rtmi_multi_create(&flat.desc(), devices.as_ptr(), devices.len() as u32, &mut raw) })?;
        Ok(DeviceScene { raw })
    }
    /// allocate the per-sample buffers of renders with these parameters now instead of in the first render (optional)
    pub fn prepare(&mut self, p: &RtmiRenderParams) -> Result<(), RtmiError> {
        check(unsafe { rtmi_multi_prepare(self.raw, p) })
    }
    /// What brings the tiles of a render together: 0 nothing to exchange (one device), 1 device-to-device copies
    /// (a device listed twice), 2 one grouped ncclGather over xGMI (distinct devices; RTMI_FORCE_RCCL=1 for one).
    pub fn collective(&self) -> i32 {
        unsafe { rtmi_multi_collective(self.raw) }
    }
    /// Blocking whole-image render over all devices of the handle (tile_rank / tile_world = 0 / 1).
    pub fn render(&mut self, cam: &RtmiCamera, p: &RtmiRenderParams) -> Result<Image, RtmiError> {
        let (nx, ny) = (p.nx as usize, p.ny as usize);
        let mut img = Image { nx, ny, linear: vec![0.0; nx * ny * 3], rgb8: vec![0; nx * ny * 3], stats: RtmiStats::default() };
        check(unsafe { rtmi_multi_render(self.raw, cam, p, img.linear.as_mut_ptr(), img.rgb8.as_mut_ptr(), &mut img.stats) })?;
        Ok(img)
    }
}

impl Drop for DeviceScene {
    fn drop(&mut self) {
        unsafe { rtmi_multi_destroy(self.raw) }
    }
}

/// `Camera` of src/camera.rs:8-18, with the state `Camera::new` derives (camera.rs:21-51) in f64.
pub struct Camera {
    pub origin: [f64; 3],
    pub lower_left_corner: [f64; 3],
    pub horizontal: [f64; 3],
    pub vertical: [f64; 3],
    pub u: [f64; 3],
    pub v: [f64; 3],
    pub time0: f64,
    pub time1: f64,
    pub lens_radius: f64,
}

/// What the reference fixes in code (color.rs:7,9; tests/test.rs:56) plus the device choices.
pub struct RenderOptions {
    pub seed: u64,
    pub max_depth: u32,
    pub t_min: f64,
    pub flags: u32,
    /// empty = `device`; otherwise the GPUs of this process to split the image over (a `DeviceScene` for one image)
    pub devices: Vec<i32>,
    pub device: i32,
    /// called about every 50 ms with (work units done, total); return false to cancel (replaces progressbar.rs)
    pub progress: Option<Box<dyn FnMut(u64, u64) -> bool>>,
}

impl Default for RenderOptions {
    fn default() -> Self {
        RenderOptions { seed: 42, max_depth: 50, t_min: 0.001, flags: RTMI_FLAG_FAST_CULL, devices: Vec::new(), device: 0, progress: None }
    }
}

fn sub(a: [f64; 3], b: [f64; 3]) -> [f64; 3] { [a[0] - b[0], a[1] - b[1], a[2] - b[2]] }
fn scale(a: [f64; 3], s: f64) -> [f64; 3] { [a[0] * s, a[1] * s, a[2] * s] }
fn cross(a: [f64; 3], b: [f64; 3]) -> [f64; 3] { [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]] }
fn normalize(a: [f64; 3]) -> [f64; 3] { let n = (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]).sqrt(); [a[0] / n, a[1] / n, a[2] / n] }

unsafe extern "C" fn progress_trampoline(done: u64, total: u64, user: *mut std::os::raw::c_void) -> std::os::raw::c_int {
    let f = &mut *(user as *mut Box<dyn FnMut(u64, u64) -> bool>);
    // nothing may unwind through the C ABI
    match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| f(done, total))) {
        Ok(true) => 0,
        _ => 1,
    }
}

impl Camera {
    /// camera.rs:21-51 — note the parameter order (…, vfov, aspect, aperture, focus_dist, t0, t1)
    #[allow(clippy::too_many_arguments)]
    pub fn new(look_from: [f64; 3], look_at: [f64; 3], view_up: [f64; 3], vertical_fov: f64, aspect: f64, aperture: f64,
               focus_dist: f64, time0: f64, time1: f64) -> Camera {
        let theta = vertical_fov * std::f64::consts::PI / 180.0;
        let half_height = focus_dist * (theta / 2.0).tan();
        let half_width = aspect * half_height;
        let w = normalize(sub(look_from, look_at));
        let u = normalize(cross(view_up, w));
        let v = cross(w, u);
        let llc = sub(sub(sub(look_from, scale(u, half_width)), scale(v, half_height)), scale(w, focus_dist));
        Camera {
            origin: look_from, lower_left_corner: llc, horizontal: scale(u, 2.0 * half_width), vertical: scale(v, 2.0 * half_height),
            u, v, time0, time1, lens_radius: aperture / 2.0,
        }
    }
    /// f64 state rounded ONCE to the device's f32 (arithmetic contract item 4)
    pub fn lower(&self) -> RtmiCamera {
        let f = |a: [f64; 3]| [a[0] as f32, a[1] as f32, a[2] as f32];
        RtmiCamera {
            origin: f(self.origin), lower_left_corner: f(self.lower_left_corner), horizontal: f(self.horizontal), vertical: f(self.vertical),
            u: f(self.u), v: f(self.v), time0: self.time0 as f32, time1: self.time1 as f32, lens_radius: self.lens_radius as f32,
        }
    }
    /// The addition the north star asks for: `cam.render(&world, nx, ny, ns)` = lower + upload + the triple loop of
    /// create_image on the GPU(s) + free.  `world` is what `Hittable::describe()` returns for the reference's world.
    pub fn render(&self, world: &std::rc::Rc<desc::HittableDesc>, nx: usize, ny: usize, ns: usize, opt: &mut RenderOptions) -> Result<Image, RtmiError> {
        let flat = lower::lower_world(world).map_err(|e| RtmiError { code: RTMI_ERR_UNSUPPORTED, message: format!("{:?}", e) })?;
        let mut p = default_params(nx, ny, ns, opt.seed);
        p.max_depth = opt.max_depth;
        p.t_min = opt.t_min as f32;
        p.flags = opt.flags;
        if let Some(cb) = opt.progress.as_mut() {
            p.progress_fn = progress_trampoline as usize as u64;
            p.progress_user = cb as *mut Box<dyn FnMut(u64, u64) -> bool> as usize as u64;
        }
        let cam = self.lower();
        if !opt.devices.is_empty() {
            // several GPUs of this process: tiles t % n, one gather.  Hosts that render more than one image keep the
            // DeviceScene instead (uploads and buffers are then paid once, not per image).
            return DeviceScene::upload(&flat, &opt.devices)?.render(&cam, &p);
        }
        Scene::upload(&flat, opt.device)?.render(&cam, &p)
    }
}

/// tests/test.rs:55-85 — `create_image(ny, nx, ns, cam, world) -> String`; note the argument order (ny, nx, …)
pub fn create_image(ny: usize, nx: usize, ns: usize, cam: &Camera, world: &std::rc::Rc<desc::HittableDesc>) -> Result<String, RtmiError> {
    Ok(cam.render(world, nx, ny, ns, &mut RenderOptions::default())?.to_ppm())
}
