"""Test-only scenes (NOT reference scenes): lit variants that exercise the BVH, instancing and
media with non-zero radiance, so that parity is visible in the image as well as in the path
signature.  Built with the same backend-agnostic `api` as raytracing_rust_amd.scenes."""
import numpy as np

from raytracing_rust_amd import scenes
from raytracing_rust_amd.philox import SceneRng


def lit_random_spheres(api, seed=7, n=6):
    """A BVH over mixed Sphere/MovingSphere with Lambertian/Metal/Dielectric plus emitters."""
    rng = SceneRng(seed)
    api.seed_scene_rng(seed)
    objs = []
    checker = api.CheckerTexture(api.SolidTexture(0.2, 0.3, 0.1), api.SolidTexture(0.9, 0.9, 0.9))
    objs.append(api.Sphere((0.0, -1000.0, 0.0), 1000.0, api.Lambertian(checker)))
    for a in range(-n, n):
        for b in range(-n, n):
            m = rng.gen()
            c = np.array([a + 0.9 * rng.gen(), 0.2, b + 0.9 * rng.gen()])
            if m < 0.5:
                alb = api.SolidTexture(rng.gen(), rng.gen(), rng.gen())
                objs.append(api.MovingSphere(c, c + np.array([0.0, 0.5 * rng.gen(), 0.0]), 0.0, 1.0, 0.2,
                                             api.Lambertian(alb)))
            elif m < 0.7:
                alb = api.SolidTexture(0.5 * (1 + rng.gen()), 0.5 * (1 + rng.gen()), 0.5 * (1 + rng.gen()))
                objs.append(api.Sphere(c, 0.2, api.Metal(alb, 0.5 * rng.gen())))
            elif m < 0.85:
                objs.append(api.Sphere(c, 0.2, api.Dielectric(1.5)))
            else:
                e = api.SolidTexture(4.0 * rng.gen(), 4.0 * rng.gen(), 4.0 * rng.gen())
                objs.append(api.Sphere(c, 0.2, api.DiffuseLight(e)))
    objs.append(api.Sphere((0.0, 1.0, 0.0), 1.0, api.Dielectric(1.5)))
    objs.append(api.Sphere((-4.0, 1.0, 0.0), 1.0, api.Lambertian(api.SolidTexture(0.4, 0.2, 0.1))))
    objs.append(api.Sphere((4.0, 1.0, 0.0), 1.0, api.Metal(api.SolidTexture(0.7, 0.6, 0.5), 0.0)))
    objs.append(api.Sphere((0.0, 6.0, 0.0), 2.0, api.DiffuseLight(api.SolidTexture(5.0, 5.0, 5.0))))
    return api.BVHNode(objs, 0.0, 1.0)


def lit_final_scene(api, seed=1):
    """final_scene with the light rect's x range the right way round (tests/test.rs:444-452 has
    x0=147 > x1=123); everything else as the reference builds it."""
    rng = SceneRng(seed)
    api.seed_scene_rng(seed)
    white = api.Lambertian(api.SolidTexture(0.73, 0.73, 0.73))
    ground = api.Lambertian(api.SolidTexture(0.48, 0.83, 0.53))
    world = api.HittableList()
    boxes = []
    for i in range(20):
        for j in range(20):
            x0, z0 = -1000.0 + i * 100.0, -1000.0 + j * 100.0
            boxes.append(api.Cube((x0, 0.0, z0), (x0 + 100.0, 100.0 * (rng.gen() + 0.01), z0 + 100.0), ground))
    world.push(api.BVHNode(boxes, 0.0, 1.0))
    world.push(api.Rect(api.PLANE_ZX, 147.0, 123.0, 412.0, 423.0, 554.0, api.DiffuseLight(api.SolidTexture(7.0, 7.0, 7.0))))
    c = np.array([400.0, 400.0, 200.0])
    world.push(api.MovingSphere(c, c + np.array([30.0, 0.0, 0.0]), 0.0, 1.0, 50.0,
                                api.Lambertian(api.SolidTexture(0.7, 0.3, 0.1))))
    world.push(api.Sphere((260.0, 150.0, 45.0), 50.0, api.Dielectric(1.5)))
    world.push(api.Sphere((0.0, 150.0, 145.0), 50.0, api.Metal(api.SolidTexture(0.8, 0.8, 0.9), 10.0)))
    world.push(api.Sphere((360.0, 150.0, 145.0), 70.0, api.Dielectric(1.5)))
    world.push(api.ConstantMedium(api.Sphere((360.0, 150.0, 145.0), 70.0, api.Dielectric(1.5)), 0.2,
                                  api.SolidTexture(0.2, 0.4, 0.9)))
    world.push(api.ConstantMedium(api.Sphere((0.0, 0.0, 0.0), 5000.0, api.Dielectric(1.5)), 0.0001,
                                  api.SolidTexture(1.0, 1.0, 1.0)))
    data, nx, ny = scenes.earthmap_rgb8()
    world.push(api.Sphere((400.0, 200.0, 400.0), 100.0, api.Lambertian(api.ImageTexture(data, nx, ny))))
    world.push(api.Sphere((220.0, 280.0, 300.0), 80.0, api.Lambertian(api.NoiseTexture(0.1))))
    balls = [api.Sphere((165.0 * rng.gen(), 165.0 * rng.gen(), 165.0 * rng.gen()), 10.0, white) for _ in range(1000)]
    world.push(api.Traslate(api.Rotate(api.AXIS_Y, api.BVHNode(balls, 0.0, 0.1), 15.0), (-100.0, 270.0, 395.0)))
    return world


def lit_smoke(api, seed=1):
    """cornell_smoke with the back wall at z=555 (the reference puts it at k=0, tests/test.rs:369-377)."""
    api.seed_scene_rng(seed)
    red = api.Lambertian(api.SolidTexture(0.65, 0.05, 0.05))
    white = api.Lambertian(api.SolidTexture(0.73, 0.73, 0.73))
    green = api.Lambertian(api.SolidTexture(0.12, 0.45, 0.15))
    light = api.DiffuseLight(api.SolidTexture(7.0, 7.0, 7.0))
    world = api.HittableList()
    world.push(api.FlipNormals(api.Rect(api.PLANE_YZ, 0.0, 0.0, 555.0, 555.0, 555.0, green)))
    world.push(api.Rect(api.PLANE_YZ, 0.0, 0.0, 555.0, 555.0, 0.0, red))
    world.push(api.Rect(api.PLANE_ZX, 127.0, 113.0, 432.0, 443.0, 554.0, light))
    world.push(api.FlipNormals(api.Rect(api.PLANE_ZX, 0.0, 0.0, 555.0, 555.0, 555.0, white)))
    world.push(api.Rect(api.PLANE_ZX, 0.0, 0.0, 555.0, 555.0, 0.0, white))
    world.push(api.FlipNormals(api.Rect(api.PLANE_XY, 0.0, 0.0, 555.0, 555.0, 555.0, white)))
    b1 = api.Traslate(api.Rotate(api.AXIS_Y, api.Cube((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white), -18.0), (130.0, 0.0, 65.0))
    b2 = api.Traslate(api.Rotate(api.AXIS_Y, api.Cube((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), white), 15.0), (265.0, 0.0, 295.0))
    world.push(api.ConstantMedium(b1, 0.01, api.SolidTexture(1.0, 1.0, 1.0)))
    world.push(api.ConstantMedium(b2, 0.01, api.SolidTexture(0.0, 0.0, 0.0)))
    return world


def hollow_glass(api, seed=1):
    """The hollow-glass idiom: concentric Sphere(r) and Sphere(-0.9 r) — inside BVHs of 2 and of 5 leaves, next to
    plain neighbours.  Sphere::bounding_box of the negative radius is an inverted box (sphere.rs:79-84); the
    reference never tests a leaf's box (bvh.rs:72-73), so the inner sphere must stay reachable in every kernel."""
    api.seed_scene_rng(seed)
    glass = api.Dielectric(1.5)
    world = api.HittableList()
    world.push(api.Sphere((0.0, -1000.0, 0.0), 1000.0, api.Lambertian(api.CheckerTexture(api.SolidTexture(0.2, 0.3, 0.1), api.SolidTexture(0.9, 0.9, 0.9)))))
    world.push(api.BVHNode([api.Sphere((0.0, 1.0, 0.0), 1.0, glass), api.Sphere((0.0, 1.0, 0.0), -0.9, glass)], 0.0, 1.0))
    many = [api.Sphere((4.0, 1.0, 0.5), 1.0, glass), api.Sphere((4.0, 1.0, 0.5), -0.95, glass),
            api.Sphere((-4.0, 1.0, 0.0), 1.0, api.Lambertian(api.SolidTexture(0.4, 0.2, 0.1))),
            api.Sphere((2.0, 0.5, 2.5), 0.5, api.Metal(api.SolidTexture(0.7, 0.6, 0.5), 0.1)),
            api.Sphere((2.0, 0.5, 2.5), -0.4, glass)]
    world.push(api.BVHNode(many, 0.0, 1.0))
    world.push(api.Sphere((0.0, 9.0, 0.0), 3.0, api.DiffuseLight(api.SolidTexture(6.0, 6.0, 6.0))))
    return world


EXTRA = {
    "hollow_glass": (hollow_glass, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0),
    "lit_random_spheres": (lit_random_spheres, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0),
    "lit_final_scene": (lit_final_scene, (478.0, 278.0, -600.0), (278.0, 278.0, 0.0), 40.0),
    "lit_smoke": (lit_smoke, (278.0, 278.0, -800.0), (278.0, 278.0, 0.0), 40.0),
}


def build(api, name, nx, ny, seed=1):
    if name in scenes.SCENES or name.endswith("_corrected"):
        return scenes.build(api, name, nx, ny, seed)
    fn, look_from, look_at, vfov = EXTRA[name]
    world = fn(api, seed)
    cam = scenes.set_camera(api, nx, ny, look_from, look_at, vertical_fov=vfov)
    return cam, world
