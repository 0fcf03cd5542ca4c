//! The reference's lit / heavy scenes (tests/test.rs:242-523) as described worlds, built with the seeded scene
//! streams in exactly the draw order of raytracing_rust_amd/scenes.py (builder stream) and of the C++ mirror
//! (backend stream: BVH axes, Perlin tables) — so `lower::lower_world(&scenes::final_scene(1, false, earth))` is
//! byte-identical to tests/golden/flat_final_scene.bin.gz.  Scenes are reproduced as written, slips included;
//! `corrected = true` repairs exactly the three slips listed in scenes.py (opt-in, never the default).
//! UNVERIFIED SOURCE (no Rust toolchain in the build image).
use crate::desc::*;
use crate::philox::SceneStreams;
use std::rc::Rc;

/// tests/test.rs:242-323 — two coincident floors at y = 0 and no ceiling, as written (:268-285)
pub fn cornell_box(seed: u64, corrected: bool) -> Rc<HittableDesc> {
    let _streams = SceneStreams::new(seed); // no random draws in this scene
    let red = lambertian(solid_texture(0.65, 0.05, 0.05));
    let white = lambertian(solid_texture(0.73, 0.73, 0.73));
    let green = lambertian(solid_texture(0.12, 0.45, 0.15));
    let light = diffuse_light(solid_texture(15.0, 15.0, 15.0));
    let mut world = Vec::new();
    world.push(flip_normals(rect(PLANE_YZ, 0.0, 0.0, 555.0, 555.0, 555.0, green)));
    world.push(rect(PLANE_YZ, 0.0, 0.0, 555.0, 555.0, 0.0, red));
    world.push(rect(PLANE_ZX, 227.0, 213.0, 332.0, 343.0, 554.0, light));
    world.push(flip_normals(rect(PLANE_ZX, 0.0, 0.0, 555.0, 555.0, if corrected { 555.0 } else { 0.0 }, white.clone())));
    world.push(rect(PLANE_ZX, 0.0, 0.0, 555.0, 555.0, 0.0, white.clone()));
    world.push(flip_normals(rect(PLANE_XY, 0.0, 0.0, 555.0, 555.0, 555.0, white.clone())));
    world.push(traslate(rotate(AXIS_Y, cube([0.0, 0.0, 0.0], [165.0, 165.0, 165.0], white.clone()), -18.0), [130.0, 0.0, 65.0]));
    world.push(traslate(rotate(AXIS_Y, cube([0.0, 0.0, 0.0], [165.0, 330.0, 165.0], white), 15.0), [265.0, 0.0, 295.0]));
    hittable_list(world)
}

/// tests/test.rs:325-417 — the flipped XY wall sits at k = 0, in front of the camera (:369-377)
pub fn cornell_smoke(seed: u64, corrected: bool) -> Rc<HittableDesc> {
    let _streams = SceneStreams::new(seed);
    let red = lambertian(solid_texture(0.65, 0.05, 0.05));
    let white = lambertian(solid_texture(0.73, 0.73, 0.73));
    let green = lambertian(solid_texture(0.12, 0.45, 0.15));
    let light = diffuse_light(solid_texture(7.0, 7.0, 7.0));
    let mut world = Vec::new();
    world.push(flip_normals(rect(PLANE_YZ, 0.0, 0.0, 555.0, 555.0, 555.0, green)));
    world.push(rect(PLANE_YZ, 0.0, 0.0, 555.0, 555.0, 0.0, red));
    world.push(rect(PLANE_ZX, 127.0, 113.0, 432.0, 443.0, 554.0, light));
    world.push(flip_normals(rect(PLANE_ZX, 0.0, 0.0, 555.0, 555.0, 0.0, white.clone())));
    world.push(rect(PLANE_ZX, 0.0, 0.0, 555.0, 555.0, 555.0, white.clone()));
    world.push(flip_normals(rect(PLANE_XY, 0.0, 0.0, 555.0, 555.0, if corrected { 555.0 } else { 0.0 }, white.clone())));
    let box1 = traslate(rotate(AXIS_Y, cube([0.0, 0.0, 0.0], [165.0, 165.0, 165.0], white.clone()), -18.0), [130.0, 0.0, 65.0]);
    let box2 = traslate(rotate(AXIS_Y, cube([0.0, 0.0, 0.0], [165.0, 330.0, 165.0], white), 15.0), [265.0, 0.0, 295.0]);
    world.push(constant_medium(box1, 0.01, solid_texture(1.0, 1.0, 1.0)));
    world.push(constant_medium(box2, 0.01, solid_texture(0.0, 0.0, 0.0)));
    hittable_list(world)
}

/// tests/test.rs:419-523 — the light rect has x0 = 147 > x1 = 123 and is never hit (:444-452).
/// `earth`: texture/earthmap.jpg decoded to row-major RGB8 (tests/test.rs:489-493), (data, nx, ny).
pub fn final_scene(seed: u64, corrected: bool, earth: (Vec<u8>, u32, u32)) -> Rc<HittableDesc> {
    let mut s = SceneStreams::new(seed);
    let white = lambertian(solid_texture(0.73, 0.73, 0.73));
    let ground = lambertian(solid_texture(0.48, 0.83, 0.53));
    let mut world = Vec::new();
    let mut box_list1 = Vec::new();
    for i in 0..20 {
        for j in 0..20 {
            let w = 100.0;
            let x0 = -1000.0 + i as f64 * w;
            let z0 = -1000.0 + j as f64 * w;
            let y0 = 0.0;
            let x1 = x0 + w;
            let y1 = 100.0 * (s.builder.gen() + 0.01);
            let z1 = z0 + w;
            box_list1.push(cube([x0, y0, z0], [x1, y1, z1], ground.clone()));
        }
    }
    world.push(bvh_new(&mut box_list1, 0.0, 1.0, &mut s.backend));
    let light = diffuse_light(solid_texture(7.0, 7.0, 7.0));
    if corrected {
        // the book's xz_rect(123, 423, 147, 412, 554) in the reference's ZX order (z0, x0, z1, x1)
        world.push(rect(PLANE_ZX, 147.0, 123.0, 412.0, 423.0, 554.0, light));
    } else {
        world.push(rect(PLANE_ZX, 147.0, 412.0, 123.0, 423.0, 554.0, light));
    }
    let center = [400.0, 400.0, 200.0];
    world.push(moving_sphere(center, [center[0] + 30.0, center[1], center[2]], 0.0, 1.0, 50.0, lambertian(solid_texture(0.7, 0.3, 0.1))));
    world.push(sphere([260.0, 150.0, 45.0], 50.0, dielectric(1.5)));
    world.push(sphere([0.0, 150.0, 145.0], 50.0, metal(solid_texture(0.8, 0.8, 0.9), 10.0)));
    world.push(sphere([360.0, 150.0, 145.0], 70.0, dielectric(1.5)));
    let boundary_clone = sphere([360.0, 150.0, 145.0], 70.0, dielectric(1.5));
    world.push(constant_medium(boundary_clone, 0.2, solid_texture(0.2, 0.4, 0.9)));
    let fog = sphere([0.0, 0.0, 0.0], 5000.0, dielectric(1.5));
    world.push(constant_medium(fog, 0.0001, solid_texture(1.0, 1.0, 1.0)));
    world.push(sphere([400.0, 200.0, 400.0], 100.0, lambertian(image_texture(earth.0, earth.1, earth.2))));
    world.push(sphere([220.0, 280.0, 300.0], 80.0, lambertian(noise_texture(0.1, &mut s.backend))));
    let mut box_list2 = Vec::new();
    for _ in 0..1000 {
        let x = 165.0 * s.builder.gen();
        let y = 165.0 * s.builder.gen();
        let z = 165.0 * s.builder.gen();
        box_list2.push(sphere([x, y, z], 10.0, white.clone()));
    }
    let bvh2 = bvh_new(&mut box_list2, 0.0, 0.1, &mut s.backend);
    world.push(traslate(rotate(AXIS_Y, bvh2, 15.0), [-100.0, 270.0, 395.0]));
    hittable_list(world)
}

/// The compositions the lowering accepts beyond the reference's own scenes, in one world (twin of `compositions` in
/// tools/dump_flat_scene.py; golden: tests/golden/flat_compositions.bin.gz): a HittableList as a BVH child with exact ties
/// inside (coincident spheres, coincident rects, a nested list), instanced primitives as BVH leaves and list members,
/// FlipNormals around an inner BVHNode, a ConstantMedium inside Traslate(Rotate(..)).
pub fn compositions(seed: u64) -> Rc<HittableDesc> {
    let mut s = SceneStreams::new(seed);
    let red = diffuse_light(solid_texture(4.0, 0.2, 0.2));
    let green = diffuse_light(solid_texture(0.2, 4.0, 0.2));
    let blue = diffuse_light(solid_texture(0.2, 0.2, 4.0));
    let grey = lambertian(solid_texture(0.7, 0.7, 0.7));
    let glass = dielectric(1.5);
    let inner = hittable_list(vec![
        traslate(cube([0.0, 0.0, -0.5], [1.0, 1.0, 0.5], grey.clone()), [1.0, 0.0, 0.0]),
        flip_normals(sphere([1.5, 1.4, 0.0], 0.4, red.clone())),
    ]);
    let lst = hittable_list(vec![
        sphere([0.0, 0.5, 0.0], 0.5, red),
        sphere([0.0, 0.5, 0.0], 0.5, green.clone()),
        rect(PLANE_XY, -2.0, 0.0, -1.0, 1.0, 0.0, blue),
        rect(PLANE_XY, -2.0, 0.0, -1.0, 1.0, 0.0, green),
        inner,
    ]);
    let mut sub_list = vec![
        sphere([-3.0, 0.5, 1.0], 0.5, grey.clone()),
        rotate(AXIS_Y, cube([-0.3, 0.0, -0.3], [0.3, 0.8, 0.3], grey.clone()), 30.0),
        sphere([-3.0, 0.5, -1.0], 0.5, glass.clone()),
    ];
    let sub = bvh_new(&mut sub_list, 0.0, 1.0, &mut s.backend);
    let mut objs = vec![
        lst,
        sphere([0.0, -100.0, 0.0], 100.0, grey.clone()),
        sphere([3.0, 0.5, 0.0], 0.5, grey.clone()),
        flip_normals(sub),
        flip_normals(rect(PLANE_XY, -4.0, 0.0, 4.0, 3.0, -2.0, grey.clone())),
        traslate(rotate(AXIS_Z, moving_sphere([0.0, 0.0, 0.0], [0.0, 0.3, 0.0], 0.0, 1.0, 0.3, grey), 20.0), [2.0, 2.0, 1.0]),
    ];
    let bvh = bvh_new(&mut objs, 0.0, 1.0, &mut s.backend);
    hittable_list(vec![
        bvh,
        traslate(rotate(AXIS_Y, constant_medium(sphere([0.0, 0.0, 0.0], 1.0, glass), 0.5, solid_texture(0.9, 0.9, 0.9)), 25.0), [-1.5, 1.5, 2.0]),
        sphere([0.0, 6.0, 0.0], 1.5, diffuse_light(solid_texture(4.0, 4.0, 4.0))),
    ])
}

/// ConstantMedium as a child of a BVHNode (twin of `media_in_bvh` in tools/dump_flat_scene.py; golden:
/// tests/golden/flat_media_in_bvh.bin.gz): a BVH of primitives, media and an instanced subtree inside Traslate(Rotate(..)), a BVHNode over ONE
/// object that is a BVH with a medium in it (evaluated on both sides), and a BVHNode over one medium (no primitives at all).
pub fn media_in_bvh(seed: u64) -> Rc<HittableDesc> {
    let mut s = SceneStreams::new(seed);
    let grey = lambertian(solid_texture(0.7, 0.7, 0.7));
    let glass = dielectric(1.5);
    let mut objs = vec![
        sphere([-3.0, 0.0, 0.0], 0.9, grey.clone()),
        constant_medium(sphere([-1.0, 0.2, 0.5], 1.0, glass.clone()), 1.5, solid_texture(0.9, 0.2, 0.2)),
        cube([0.3, -1.0, -0.8], [1.5, 0.4, 0.6], grey.clone()),
        traslate(constant_medium(cube([0.0, 0.0, 0.0], [1.2, 1.2, 1.2], glass.clone()), 2.5, solid_texture(0.2, 0.9, 0.2)), [1.8, -0.9, 0.8]),
        sphere([3.3, 0.1, -0.3], 0.8, grey.clone()),
    ];
    // an instanced subtree (Traslate / Rotate around a BVHNode as a child of a BVHNode): a deferred BVH item with its gate records
    let mut sub_list = vec![sphere([0.0, 0.0, 0.0], 0.4, grey.clone()), cube([0.5, -0.3, -0.3], [1.1, 0.3, 0.3], grey.clone())];
    let sub = bvh_new(&mut sub_list, 0.0, 1.0, &mut s.backend);
    objs.push(traslate(rotate(AXIS_Z, sub, 20.0), [4.5, 1.0, 0.5]));
    let mut inner_list = vec![
        sphere([-4.5, 1.2, 1.5], 0.5, grey.clone()),
        constant_medium(sphere([-4.2, 1.3, 1.4], 1.0, glass.clone()), 1.0, solid_texture(0.4, 0.9, 0.6)),
        cube([-5.6, 0.2, 0.8], [-5.0, 0.9, 1.6], grey.clone()),
    ];
    // (construction order = the order of the scene stream's draws in the twin: sub, inner, then the rest)
    let inner = bvh_new(&mut inner_list, 0.0, 1.0, &mut s.backend);
    let bvh = bvh_new(&mut objs, 0.0, 1.0, &mut s.backend);
    let mut one_bvh = vec![inner];
    let over_inner = bvh_new(&mut one_bvh, 0.0, 1.0, &mut s.backend);
    let mut one_medium = vec![constant_medium(sphere([0.0, 2.6, -1.5], 0.8, glass.clone()), 1.2, solid_texture(0.9, 0.8, 0.2))];
    let over_medium = bvh_new(&mut one_medium, 0.0, 1.0, &mut s.backend);
    // a nested medium (one item, the inner density behind its chain) ...
    // (the same two material objects as above: materials are numbered by identity, as in the C++ lowering)
    let nested = constant_medium(
        constant_medium(sphere([5.0, 3.0, -2.0], 0.7, glass.clone()), 0.8, solid_texture(0.1, 0.1, 0.1)),
        2.0,
        solid_texture(0.9, 0.5, 0.2),
    );
    // ... and a HittableList WITH MEDIA as a child of a BVHNode: a group of LISTSCAN members and a terminator behind the BVH item
    let inner_l = hittable_list(vec![
        traslate(cube([-0.3, -0.3, -0.3], [0.3, 0.3, 0.3], grey.clone()), [6.9, -0.2, 2.1]),
        constant_medium(cube([6.0, 0.3, 1.2], [7.0, 0.9, 2.0], glass.clone()), 3.0, solid_texture(0.9, 0.9, 0.3)),
    ]);
    let lst = hittable_list(vec![
        sphere([6.0, 0.0, 2.0], 0.4, grey.clone()),
        constant_medium(sphere([6.3, 0.1, 1.8], 0.8, glass), 2.0, solid_texture(0.3, 0.9, 0.4)),
        flip_normals(inner_l),
    ]);
    let mut over_list_items = vec![lst, sphere([8.0, 0.0, 2.0], 0.5, grey)];
    let over_list = bvh_new(&mut over_list_items, 0.0, 1.0, &mut s.backend);
    hittable_list(vec![
        traslate(rotate(AXIS_Y, bvh, -30.0), [-1.0, 1.2, 3.0]),
        over_inner,
        over_medium,
        nested,
        over_list,
        sphere([0.0, 9.0, 0.0], 2.0, diffuse_light(solid_texture(4.0, 4.0, 4.0))),
    ])
}

/// Camera literals of the #[test] drivers (tests/test.rs:741-752, 780-791, 819-830): (look_from, look_at, vfov);
/// every driver uses view_up = (0,1,0), focus_dist = 10, aperture = 0.1, shutter [0,1], aspect = nx/ny (:47).
pub fn camera_of(scene: &str) -> ([f64; 3], [f64; 3], f64) {
    match scene {
        "final_scene" => ([478.0, 278.0, -600.0], [278.0, 278.0, 0.0], 40.0),
        _ => ([278.0, 278.0, -800.0], [278.0, 278.0, 0.0], 40.0), // cornell_box, cornell_smoke
    }
}
