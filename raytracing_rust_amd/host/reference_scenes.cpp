// reference_scenes.cpp — the reference's scene builders (tests/test.rs:89-523) and its #[test]-style
// drivers (tests/test.rs:525-838) written against the C++ host mirror, as a stand-alone program:
//
//     rt_reference_tests <scene> <nx> <ny> <ns> <out.ppm> [seed] [scene_seed]
//
// does what `test_<scene>` does in the reference — set_camera, build the world, create_image, write the
// P3 file — with the render loop running on the MI355X through Camera::render.  Random scene parameters
// come from a seeded Philox stream (stream_id 2) instead of rand::thread_rng(), drawing in the same
// program order as raytracing_rust_amd/scenes.py, so both hosts build identical scenes.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>

#include "rt_host.hpp"

using namespace rt;

namespace {

template <typename T, typename... A>
std::shared_ptr<const T> mk(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
TexturePtr solid(double r, double g, double b) { return mk<SolidTexture>(r, g, b); }
MaterialPtr lambert(TexturePtr t) { return mk<Lambertian>(std::move(t)); }

struct SceneRng { // thread_rng() of the builders
    Rng r;
    explicit SceneRng(uint64_t seed) { r.seed(seed, 0, 0, 2); }
    double gen() { return r.gen(); }
};

// the texture file the reference opens with the `image` crate (tests/test.rs:201, 489): decoding JPEG is
// out of scope here, the caller provides decoded RGB8 (scenes.earthmap_rgb8() writes it) or gets a flat grey
std::vector<uint8_t> g_earth;
uint32_t g_earth_nx = 0, g_earth_ny = 0;
TexturePtr earth_texture() {
    if (g_earth.empty()) { g_earth.assign(3 * 4, 128); g_earth_nx = 2; g_earth_ny = 2; }
    return mk<ImageTexture>(g_earth, g_earth_nx, g_earth_ny);
}

HittablePtr random_scene(uint64_t seed) { // tests/test.rs:89-163
    SceneRng rng(seed);
    scene_rng().seed(seed, 0, 0, 1);
    const Vec3 origin(4.0, 0.2, 0.0);
    std::vector<HittablePtr> world;
    auto checker = mk<CheckerTexture>(solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9));
    world.push_back(mk<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, lambert(checker)));
    for (int a = -10; a < 10; a++)
        for (int b = -10; b < 10; b++) {
            const double choose_material = rng.gen();
            const double cx = a + 0.9 * rng.gen();
            const double cz = b + 0.9 * rng.gen();
            const Vec3 center(cx, 0.2, cz);
            if ((center - origin).magnitude() > 0.9) {
                if (choose_material < 0.8) {
                    const Vec3 c1 = center + Vec3(0.0, 0.5 * rng.gen(), 0.0);
                    const double r1 = rng.gen(), r2 = rng.gen(), g1 = rng.gen(), g2 = rng.gen(), b1 = rng.gen(), b2 = rng.gen();
                    world.push_back(mk<MovingSphere>(center, c1, 0.0, 1.0, 0.2, lambert(solid(r1 * r2, g1 * g2, b1 * b2))));
                } else if (choose_material < 0.95) {
                    const double r = 0.5 * (1.0 + rng.gen()), g = 0.5 * (1.0 + rng.gen()), bb = 0.5 * (1.0 + rng.gen());
                    const double fuzz = 0.5 * rng.gen();
                    world.push_back(mk<Sphere>(center, 0.2, mk<Metal>(solid(r, g, bb), fuzz)));
                } else {
                    world.push_back(mk<Sphere>(center, 0.2, mk<Dielectric>(1.5)));
                }
            }
        }
    world.push_back(mk<Sphere>(Vec3(0.0, 1.0, 0.0), 1.0, mk<Dielectric>(1.5)));
    world.push_back(mk<Sphere>(Vec3(-4.0, 1.0, 0.0), 1.0, lambert(solid(0.4, 0.2, 0.1))));
    world.push_back(mk<Sphere>(Vec3(4.0, 1.0, 0.0), 1.0, mk<Metal>(solid(0.7, 0.6, 0.5), 0.0)));
    return std::make_shared<BVHNode>(world, 0.0, 1.0);
}

HittablePtr two_spheres(uint64_t seed) { // tests/test.rs:165-182
    scene_rng().seed(seed, 0, 0, 1);
    auto checker = mk<CheckerTexture>(solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9));
    auto world = std::make_shared<HittableList>();
    world->push(mk<Sphere>(Vec3(0.0, -10.0, 0.0), 10.0, lambert(checker)));
    world->push(mk<Sphere>(Vec3(0.0, 10.0, 0.0), 10.0, lambert(checker)));
    return world;
}

HittablePtr two_perlin_spheres(uint64_t seed) { // tests/test.rs:184-198
    scene_rng().seed(seed, 0, 0, 1);
    auto noise = mk<NoiseTexture>(4.0);
    auto world = std::make_shared<HittableList>();
    world->push(mk<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, lambert(noise)));
    world->push(mk<Sphere>(Vec3(0.0, 2.0, 0.0), 2.0, lambert(noise)));
    return world;
}

HittablePtr earth(uint64_t seed) { // tests/test.rs:200-209
    scene_rng().seed(seed, 0, 0, 1);
    return mk<Sphere>(Vec3(0.0, 0.0, 0.0), 2.0, lambert(earth_texture()));
}

HittablePtr simple_light(uint64_t seed) { // tests/test.rs:211-240
    scene_rng().seed(seed, 0, 0, 1);
    auto noise = mk<NoiseTexture>(4.0);
    auto world = std::make_shared<HittableList>();
    world->push(mk<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, lambert(noise)));
    world->push(mk<Sphere>(Vec3(0.0, 2.0, 0.0), 2.0, lambert(noise)));
    world->push(mk<Sphere>(Vec3(0.0, 7.0, 0.0), 2.0, mk<DiffuseLight>(solid(4.0, 4.0, 4.0))));
    world->push(mk<Rect>(Plane::XY, 3.0, 1.0, 5.0, 3.0, -2.0, mk<DiffuseLight>(solid(4.0, 4.0, 4.0))));
    return world;
}

HittablePtr rotated_box(const Vec3 &size, MaterialPtr m, double angle, const Vec3 &offset) {
    return mk<Traslate>(mk<Rotate>(Axis::Y, mk<Cube>(Vec3(0.0, 0.0, 0.0), size, std::move(m)), angle), offset);
}

HittablePtr cornell_box(uint64_t seed) { // tests/test.rs:242-323
    scene_rng().seed(seed, 0, 0, 1);
    auto red = lambert(solid(0.65, 0.05, 0.05)), white = lambert(solid(0.73, 0.73, 0.73)), green = lambert(solid(0.12, 0.45, 0.15));
    MaterialPtr light = mk<DiffuseLight>(solid(15.0, 15.0, 15.0));
    auto world = std::make_shared<HittableList>();
    world->push(mk<FlipNormals>(mk<Rect>(Plane::YZ, 0.0, 0.0, 555.0, 555.0, 555.0, green)));
    world->push(mk<Rect>(Plane::YZ, 0.0, 0.0, 555.0, 555.0, 0.0, red));
    world->push(mk<Rect>(Plane::ZX, 227.0, 213.0, 332.0, 343.0, 554.0, light));
    world->push(mk<FlipNormals>(mk<Rect>(Plane::ZX, 0.0, 0.0, 555.0, 555.0, 0.0, white)));
    world->push(mk<Rect>(Plane::ZX, 0.0, 0.0, 555.0, 555.0, 0.0, white));
    world->push(mk<FlipNormals>(mk<Rect>(Plane::XY, 0.0, 0.0, 555.0, 555.0, 555.0, white)));
    world->push(rotated_box(Vec3(165.0, 165.0, 165.0), white, -18.0, Vec3(130.0, 0.0, 65.0)));
    world->push(rotated_box(Vec3(165.0, 330.0, 165.0), white, 15.0, Vec3(265.0, 0.0, 295.0)));
    return world;
}

HittablePtr cornell_smoke(uint64_t seed) { // tests/test.rs:325-417
    scene_rng().seed(seed, 0, 0, 1);
    auto red = lambert(solid(0.65, 0.05, 0.05)), white = lambert(solid(0.73, 0.73, 0.73)), green = lambert(solid(0.12, 0.45, 0.15));
    MaterialPtr light = mk<DiffuseLight>(solid(7.0, 7.0, 7.0));
    auto world = std::make_shared<HittableList>();
    world->push(mk<FlipNormals>(mk<Rect>(Plane::YZ, 0.0, 0.0, 555.0, 555.0, 555.0, green)));
    world->push(mk<Rect>(Plane::YZ, 0.0, 0.0, 555.0, 555.0, 0.0, red));
    world->push(mk<Rect>(Plane::ZX, 127.0, 113.0, 432.0, 443.0, 554.0, light));
    world->push(mk<FlipNormals>(mk<Rect>(Plane::ZX, 0.0, 0.0, 555.0, 555.0, 0.0, white)));
    world->push(mk<Rect>(Plane::ZX, 0.0, 0.0, 555.0, 555.0, 555.0, white));
    world->push(mk<FlipNormals>(mk<Rect>(Plane::XY, 0.0, 0.0, 555.0, 555.0, 0.0, white)));
    auto box1 = rotated_box(Vec3(165.0, 165.0, 165.0), white, -18.0, Vec3(130.0, 0.0, 65.0));
    auto box2 = rotated_box(Vec3(165.0, 330.0, 165.0), white, 15.0, Vec3(265.0, 0.0, 295.0));
    world->push(mk<ConstantMedium>(box1, 0.01, solid(1.0, 1.0, 1.0)));
    world->push(mk<ConstantMedium>(box2, 0.01, solid(0.0, 0.0, 0.0)));
    return world;
}

HittablePtr final_scene(uint64_t seed) { // tests/test.rs:419-523
    SceneRng rng(seed);
    scene_rng().seed(seed, 0, 0, 1);
    auto white = lambert(solid(0.73, 0.73, 0.73)), ground = lambert(solid(0.48, 0.83, 0.53));
    auto world = std::make_shared<HittableList>();
    std::vector<HittablePtr> box_list1;
    for (int i = 0; i < 20; i++)
        for (int j = 0; j < 20; j++) {
            const double w = 100.0, x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
            const double x1 = x0 + w, y1 = 100.0 * (rng.gen() + 0.01), z1 = z0 + w;
            box_list1.push_back(mk<Cube>(Vec3(x0, y0, z0), Vec3(x1, y1, z1), ground));
        }
    world->push(std::make_shared<BVHNode>(box_list1, 0.0, 1.0));
    world->push(mk<Rect>(Plane::ZX, 147.0, 412.0, 123.0, 423.0, 554.0, mk<DiffuseLight>(solid(7.0, 7.0, 7.0))));
    const Vec3 center(400.0, 400.0, 200.0);
    world->push(mk<MovingSphere>(center, center + Vec3(30.0, 0.0, 0.0), 0.0, 1.0, 50.0, lambert(solid(0.7, 0.3, 0.1))));
    world->push(mk<Sphere>(Vec3(260.0, 150.0, 45.0), 50.0, mk<Dielectric>(1.5)));
    world->push(mk<Sphere>(Vec3(0.0, 150.0, 145.0), 50.0, mk<Metal>(solid(0.8, 0.8, 0.9), 10.0)));
    world->push(mk<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, mk<Dielectric>(1.5)));
    world->push(mk<ConstantMedium>(mk<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, mk<Dielectric>(1.5)), 0.2, solid(0.2, 0.4, 0.9)));
    world->push(mk<ConstantMedium>(mk<Sphere>(Vec3(0.0, 0.0, 0.0), 5000.0, mk<Dielectric>(1.5)), 0.0001, solid(1.0, 1.0, 1.0)));
    world->push(mk<Sphere>(Vec3(400.0, 200.0, 400.0), 100.0, lambert(earth_texture())));
    world->push(mk<Sphere>(Vec3(220.0, 280.0, 300.0), 80.0, lambert(mk<NoiseTexture>(0.1))));
    std::vector<HittablePtr> box_list2;
    for (int k = 0; k < 1000; k++) {
        const double x = 165.0 * rng.gen(), y = 165.0 * rng.gen(), z = 165.0 * rng.gen();
        box_list2.push_back(mk<Sphere>(Vec3(x, y, z), 10.0, white));
    }
    world->push(mk<Traslate>(mk<Rotate>(Axis::Y, std::make_shared<BVHNode>(box_list2, 0.0, 0.1), 15.0), Vec3(-100.0, 270.0, 395.0)));
    return world;
}

struct Driver { // camera literals of the #[test] drivers (tests/test.rs:543-554 ... :819-830)
    std::function<HittablePtr(uint64_t)> build;
    Vec3 look_from, look_at;
    double vfov;
};
const std::map<std::string, Driver> &drivers() {
    static const std::map<std::string, Driver> d = {
        {"random_spheres", {random_scene, Vec3(13, 2, 3), Vec3(0, 0, 0), 20.0}},
        {"two_spheres", {two_spheres, Vec3(13, 2, 3), Vec3(0, 0, 0), 20.0}},
        {"two_perlin_spheres", {two_perlin_spheres, Vec3(13, 2, 3), Vec3(0, 0, 0), 20.0}},
        {"earth", {earth, Vec3(13, 2, 3), Vec3(0, 0, 0), 20.0}},
        {"simple_light", {simple_light, Vec3(13, 3, 3), Vec3(0, 0, 0), 50.0}},
        {"cornell_box", {cornell_box, Vec3(278, 278, -800), Vec3(278, 278, 0), 40.0}},
        {"cornell_smoke", {cornell_smoke, Vec3(278, 278, -800), Vec3(278, 278, 0), 40.0}},
        {"final_scene", {final_scene, Vec3(478, 278, -600), Vec3(278, 278, 0), 40.0}},
    };
    return d;
}

// tests/test.rs:30-53
Camera set_camera(size_t nx, size_t ny, const Vec3 &look_from, const Vec3 &look_at, const Vec3 &view_up, double vfov,
                  double focus_dist, double aperture, double time0, double time1) {
    return Camera(look_from, look_at, view_up, vfov, (double)nx / (double)ny, aperture, focus_dist, time0, time1);
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s <scene>[+sky] <nx> <ny> <ns> <out.ppm> [seed=42] [scene_seed=1] [earth.rgb8 nx ny]\n", argv[0]);
        return 2;
    }
    std::string name = argv[1];
    bool sky = false; // "<scene>+sky": opt-in background of color.rs:18-20 (RTMI_FLAG_SKY); default black (:21)
    if (name.size() > 4 && name.compare(name.size() - 4, 4, "+sky") == 0) { sky = true; name.resize(name.size() - 4); }
    const size_t nx = (size_t)atol(argv[2]), ny = (size_t)atol(argv[3]), ns = (size_t)atol(argv[4]);
    RenderOptions opt;
    opt.seed = argc > 6 ? strtoull(argv[6], nullptr, 10) : 42;
    opt.flags = RTMI_FLAG_FAST_CULL | (sky ? RTMI_FLAG_SKY : 0u);
    const uint64_t scene_seed = argc > 7 ? strtoull(argv[7], nullptr, 10) : 1;
    if (argc > 10) {
        std::ifstream f(argv[8], std::ios::binary);
        g_earth.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        g_earth_nx = (uint32_t)atol(argv[9]); g_earth_ny = (uint32_t)atol(argv[10]);
        if (g_earth.size() != (size_t)3 * g_earth_nx * g_earth_ny) { fprintf(stderr, "bad earth texture file\n"); return 2; }
    }
    auto it = drivers().find(name);
    if (it == drivers().end()) { fprintf(stderr, "unknown scene %s\n", name.c_str()); return 2; }
    try {
        const Camera cam = set_camera(nx, ny, it->second.look_from, it->second.look_at, Vec3(0.0, 1.0, 0.0), it->second.vfov,
                                      10.0, 0.1, 0.0, 1.0);
        const HittablePtr world = it->second.build(scene_seed);
        const std::string res = create_image(ny, nx, ns, cam, *world, opt); // tests/test.rs:55
        std::ofstream out(argv[5], std::ios::binary);
        out << res;                                                          // write!(file, "{}", res)
        printf("%s %zux%zux%zu -> %s (%zu bytes)\n", name.c_str(), nx, ny, ns, argv[5], res.size());
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
