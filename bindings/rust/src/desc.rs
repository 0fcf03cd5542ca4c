//! A plain description of the reference's object graph — what `Hittable::describe()` (the five-line method
//! INTEGRATION.md adds to each of the reference's types, where their private fields are visible) returns, and what
//! `lower::lower_world` turns into the flat scene of include/rtmi.h.  The variants are exactly the closed set the
//! device supports (src/sphere.rs, rect.rs, cube.rs, traslate.rs, rotate.rs, hittable.rs, medium.rs, bvh.rs,
//! material.rs, texture.rs); anything else cannot be described and `Camera::render` fails loudly, as
//! `rt::Unsupported` does in the C++ mirror (raytracing_rust_amd/host/rt_host.cpp).
//! Shared ownership (`Rc`) carries identity: one `Rc<MaterialDesc>` used by many objects is lowered once, and a
//! `Bvh` whose two sides are the same `Rc` is the reference's one-element node (bvh.rs:44-45).
//! UNVERIFIED SOURCE (no Rust toolchain in the build image).
use crate::philox::Stream;
use std::rc::Rc;

pub type V3 = [f64; 3];

pub const PLANE_YZ: u8 = 0; // rect.rs:8-12, axes (k, a, b) = (0,1,2)
pub const PLANE_ZX: u8 = 1; // (1,2,0)
pub const PLANE_XY: u8 = 2; // (2,0,1)
pub const AXIS_X: u8 = 0; // rotate.rs:7-19, the same triples
pub const AXIS_Y: u8 = 1;
pub const AXIS_Z: u8 = 2;

pub fn plane_axes(plane: u8) -> (usize, usize, usize) {
    match plane {
        0 => (0, 1, 2),
        1 => (1, 2, 0),
        _ => (2, 0, 1),
    }
}

#[derive(Clone, Copy, Debug, PartialEq)]
pub struct Aabb {
    pub min: V3,
    pub max: V3,
}

/// aabb.rs:12-19 (`f64::min` / `f64::max`: NaN-ignoring)
pub fn surrounding_box(a: &Aabb, b: &Aabb) -> Aabb {
    let mut r = *a;
    for k in 0..3 {
        r.min[k] = a.min[k].min(b.min[k]);
        r.max[k] = a.max[k].max(b.max[k]);
    }
    r
}

/// Perlin tables (perlin.rs:12-36, 59-74)
pub struct PerlinDesc {
    pub ran_vec: Vec<V3>,
    pub perm_x: Vec<usize>,
    pub perm_y: Vec<usize>,
    pub perm_z: Vec<usize>,
}

impl PerlinDesc {
    /// `Perlin::new()` on the scene stream: 256 unit vectors from normalize(-1 + 2U)^3, then three Fisher-Yates
    /// permutations `for i in (0..256).rev() { swap(i, gen_range(0..i+1)) }` — the draw order of
    /// rt::Perlin::Perlin (rt_host.cpp).
    pub fn new(rng: &mut Stream) -> Self {
        let mut ran_vec = Vec::with_capacity(256);
        for _ in 0..256 {
            let x = -1.0 + 2.0 * rng.gen();
            let y = -1.0 + 2.0 * rng.gen();
            let z = -1.0 + 2.0 * rng.gen();
            let n = (x * x + y * y + z * z).sqrt();
            ran_vec.push([x / n, y / n, z / n]);
        }
        let mut perm = |rng: &mut Stream| {
            let mut p: Vec<usize> = (0..256).collect();
            for i in (0..256usize).rev() {
                let j = rng.gen_range(i as u32 + 1) as usize;
                p.swap(i, j);
            }
            p
        };
        let perm_x = perm(rng);
        let perm_y = perm(rng);
        let perm_z = perm(rng);
        PerlinDesc { ran_vec, perm_x, perm_y, perm_z }
    }
}

pub enum TextureDesc {
    Solid { color: V3 },                                        // texture.rs:9-25
    Checker { odd: Rc<TextureDesc>, even: Rc<TextureDesc> },    // :28-48
    Noise { noise: PerlinDesc, scale: f64 },                    // :51-71
    Image { data: Vec<u8>, nx: u32, ny: u32 },                  // :74-108, row-major RGB8
}

pub enum MaterialDesc {
    Lambertian { albedo: Rc<TextureDesc> },          // material.rs:36-58
    Metal { albedo: Rc<TextureDesc>, fuzz: f64 },    // :61-92, fuzz already clamped to <= 1 (:70)
    Dielectric { ref_idx: f64 },                     // :95-131
    DiffuseLight { emit: Rc<TextureDesc> },          // :133-151
    Isotropic { albedo: Rc<TextureDesc> },           // :154-173
}

pub enum HittableDesc {
    Sphere { center: V3, radius: f64, material: Rc<MaterialDesc> },
    MovingSphere { center0: V3, center1: V3, time0: f64, time1: f64, radius: f64, material: Rc<MaterialDesc> },
    Rect { plane: u8, x0: f64, y0: f64, x1: f64, y1: f64, k: f64, material: Rc<MaterialDesc> },
    Cube { p_min: V3, p_max: V3, material: Rc<MaterialDesc> },
    Traslate { inner: Rc<HittableDesc>, offset: V3 },
    /// `bbox_is_some`: Rotate::new keeps `hittable.bounding_box(0,1)` only as Some/None — its extent is the whole
    /// space because of the inverted initialisation at rotate.rs:36-37
    Rotate { axis: u8, inner: Rc<HittableDesc>, sin_theta: f64, cos_theta: f64 },
    FlipNormals { inner: Rc<HittableDesc> },
    /// phase = Isotropic(texture) (medium.rs:19-24)
    ConstantMedium { boundary: Rc<HittableDesc>, density: f64, phase: Rc<MaterialDesc> },
    List { list: Vec<Rc<HittableDesc>> },
    Bvh { left: Rc<HittableDesc>, right: Rc<HittableDesc>, bbox: Aabb },
}

// ---- constructors with the reference's `new` signatures ---------------------------------------------------------
pub fn solid_texture(r: f64, g: f64, b: f64) -> Rc<TextureDesc> {
    Rc::new(TextureDesc::Solid { color: [r, g, b] })
}
pub fn checker_texture(odd: Rc<TextureDesc>, even: Rc<TextureDesc>) -> Rc<TextureDesc> {
    Rc::new(TextureDesc::Checker { odd, even })
}
pub fn noise_texture(scale: f64, rng: &mut Stream) -> Rc<TextureDesc> {
    Rc::new(TextureDesc::Noise { noise: PerlinDesc::new(rng), scale })
}
pub fn image_texture(data: Vec<u8>, nx: u32, ny: u32) -> Rc<TextureDesc> {
    Rc::new(TextureDesc::Image { data, nx, ny })
}
pub fn lambertian(albedo: Rc<TextureDesc>) -> Rc<MaterialDesc> {
    Rc::new(MaterialDesc::Lambertian { albedo })
}
pub fn metal(albedo: Rc<TextureDesc>, fuzz: f64) -> Rc<MaterialDesc> {
    Rc::new(MaterialDesc::Metal { albedo, fuzz: if fuzz < 1.0 { fuzz } else { 1.0 } }) // material.rs:70
}
pub fn dielectric(ref_idx: f64) -> Rc<MaterialDesc> {
    Rc::new(MaterialDesc::Dielectric { ref_idx })
}
pub fn diffuse_light(emit: Rc<TextureDesc>) -> Rc<MaterialDesc> {
    Rc::new(MaterialDesc::DiffuseLight { emit })
}
pub fn isotropic(albedo: Rc<TextureDesc>) -> Rc<MaterialDesc> {
    Rc::new(MaterialDesc::Isotropic { albedo })
}
pub fn sphere(center: V3, radius: f64, material: Rc<MaterialDesc>) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::Sphere { center, radius, material })
}
pub fn moving_sphere(center0: V3, center1: V3, time0: f64, time1: f64, radius: f64, material: Rc<MaterialDesc>) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::MovingSphere { center0, center1, time0, time1, radius, material })
}
pub fn rect(plane: u8, x0: f64, y0: f64, x1: f64, y1: f64, k: f64, material: Rc<MaterialDesc>) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::Rect { plane, x0, y0, x1, y1, k, material })
}
pub fn cube(p_min: V3, p_max: V3, material: Rc<MaterialDesc>) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::Cube { p_min, p_max, material })
}
pub fn traslate(inner: Rc<HittableDesc>, offset: V3) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::Traslate { inner, offset })
}
/// Rotate::new(axis, hittable, angle in degrees) — rotate.rs:30-34
pub fn rotate(axis: u8, inner: Rc<HittableDesc>, angle: f64) -> Rc<HittableDesc> {
    let radians = (std::f64::consts::PI / 180.0) * angle;
    Rc::new(HittableDesc::Rotate { axis, inner, sin_theta: radians.sin(), cos_theta: radians.cos() })
}
pub fn flip_normals(inner: Rc<HittableDesc>) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::FlipNormals { inner })
}
pub fn constant_medium(boundary: Rc<HittableDesc>, density: f64, texture: Rc<TextureDesc>) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::ConstantMedium { boundary, density, phase: isotropic(texture) })
}
pub fn hittable_list(list: Vec<Rc<HittableDesc>>) -> Rc<HittableDesc> {
    Rc::new(HittableDesc::List { list })
}

fn moving_center(c0: &V3, c1: &V3, t0: f64, t1: f64, time: f64) -> V3 {
    let f = (time - t0) / (t1 - t0); // sphere.rs:115-118
    [c0[0] + f * (c1[0] - c0[0]), c0[1] + f * (c1[1] - c0[1]), c0[2] + f * (c1[2] - c0[2])]
}

impl HittableDesc {
    /// `Hittable::bounding_box(t0, t1)` with the reference's semantics, slips included: Rect ignores its plane
    /// (rect.rs:71-75), Rotate covers the whole space (rotate.rs:36-37), Sphere with a negative radius gives an
    /// inverted box (sphere.rs:79-84).
    pub fn bounding_box(&self, t0: f64, t1: f64) -> Option<Aabb> {
        match self {
            HittableDesc::Sphere { center, radius, .. } => {
                let r = *radius;
                Some(Aabb { min: [center[0] - r, center[1] - r, center[2] - r], max: [center[0] + r, center[1] + r, center[2] + r] })
            }
            HittableDesc::MovingSphere { center0, center1, time0, time1, radius, .. } => {
                let r = *radius;
                let a = moving_center(center0, center1, *time0, *time1, t0);
                let b = moving_center(center0, center1, *time0, *time1, t1);
                let ba = Aabb { min: [a[0] - r, a[1] - r, a[2] - r], max: [a[0] + r, a[1] + r, a[2] + r] };
                let bb = Aabb { min: [b[0] - r, b[1] - r, b[2] - r], max: [b[0] + r, b[1] + r, b[2] + r] };
                Some(surrounding_box(&ba, &bb))
            }
            HittableDesc::Rect { x0, y0, x1, y1, k, .. } => Some(Aabb { min: [*x0, *y0, *k - 0.0001], max: [*x1, *y1, *k + 0.0001] }),
            HittableDesc::Cube { p_min, p_max, .. } => Some(Aabb { min: *p_min, max: *p_max }),
            HittableDesc::Traslate { inner, offset } => inner.bounding_box(t0, t1).map(|b| Aabb {
                min: [b.min[0] + offset[0], b.min[1] + offset[1], b.min[2] + offset[2]],
                max: [b.max[0] + offset[0], b.max[1] + offset[1], b.max[2] + offset[2]],
            }),
            HittableDesc::Rotate { inner, .. } => inner.bounding_box(0.0, 1.0).map(|_| Aabb { min: [f64::MIN; 3], max: [f64::MAX; 3] }),
            HittableDesc::FlipNormals { inner } => inner.bounding_box(t0, t1),
            HittableDesc::ConstantMedium { boundary, .. } => boundary.bounding_box(t0, t1),
            HittableDesc::List { list } => {
                // hittable.rs:49-64
                let mut acc = list.first()?.bounding_box(t0, t1)?;
                for h in list.iter().skip(1) {
                    acc = surrounding_box(&acc, &h.bounding_box(t0, t1)?);
                }
                Some(acc)
            }
            HittableDesc::Bvh { bbox, .. } => Some(*bbox),
        }
    }
}

/// `BVHNode::new(&mut list, time0, time1)` — bvh.rs:17-66 — with the split axis drawn from the scene stream.
/// The reference sorts with `sort_unstable_by` and a comparator that only answers Less/Greater (:32-36), which leaves
/// the order of equal keys unspecified; like the C++ mirror this sorts STABLY on `a.min[axis] - b.min[axis] < 0`,
/// one valid outcome.  Panics where the reference panics (no bounding box, :30, :58).
pub fn bvh_new(list: &mut [Rc<HittableDesc>], time0: f64, time1: f64, rng: &mut Stream) -> Rc<HittableDesc> {
    assert!(!list.is_empty(), "BVHNode::new on an empty slice");
    let axis = rng.gen_range(3) as usize;
    let mut keyed: Vec<(f64, Rc<HittableDesc>)> = list
        .iter()
        .map(|h| (h.bounding_box(time0, time1).expect("No bounding box in BVHNode").min[axis], h.clone()))
        .collect();
    // stable insertion by the reference's predicate (a - b < 0): std's stable sort_by needs a total order, so the
    // comparator answers Equal when neither side is Less
    keyed.sort_by(|a, b| {
        if a.0 - b.0 < 0.0 {
            std::cmp::Ordering::Less
        } else if b.0 - a.0 < 0.0 {
            std::cmp::Ordering::Greater
        } else {
            std::cmp::Ordering::Equal
        }
    });
    for (slot, (_, h)) in list.iter_mut().zip(keyed.into_iter()) {
        *slot = h;
    }
    let len = list.len();
    let (left, right) = if len == 1 {
        (list[0].clone(), list[0].clone())
    } else if len == 2 {
        (list[0].clone(), list[1].clone())
    } else {
        let (lo, hi) = list.split_at_mut(len / 2);
        let l = bvh_new(lo, time0, time1, rng);
        let r = bvh_new(hi, time0, time1, rng);
        (l, r)
    };
    let lb = left.bounding_box(time0, time1).expect("No bounding box in BVHNode");
    let rb = right.bounding_box(time0, time1).expect("No bounding box in BVHNode");
    Rc::new(HittableDesc::Bvh { left, right, bbox: surrounding_box(&lb, &rb) })
}
