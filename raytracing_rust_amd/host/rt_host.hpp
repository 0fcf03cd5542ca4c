// rt_host.hpp — C++ mirror of the reference's public surface for the render path.
//
// The reference is a Rust crate (`raytracing_in_rust`, src/lib.rs:1-18); there is no Rust
// toolchain in the build image, so the host side above the C ABI (include/rtmi.h) is this
// C++ library.  It keeps the reference's names, constructor argument order and error
// behaviour: traits Hittable (src/hittable.rs:18-21), Material (src/material.rs:30-33),
// Texture (src/texture.rs:4-6); types Sphere, MovingSphere, Rect, Plane, Cube, Traslate
// (sic), Rotate, Axis, FlipNormals, ConstantMedium, HittableList, BVHNode, Lambertian,
// Metal, Dielectric, DiffuseLight, Isotropic, SolidTexture, CheckerTexture, NoiseTexture,
// ImageTexture, Perlin, Camera, Ray, AABB, HitRecord; free functions color (src/color.rs:6),
// create_image (tests/test.rs:55) and the addition the north star asks for, Camera::render.
//
// Every object can (i) be evaluated on the CPU in f64 exactly like the reference
// (hit / bounding_box / scatter / emitted / value) and (ii) lower() itself into the flat
// scene description the device consumes.  lower() is defined for the closed set of types
// above; user-defined trait implementations cannot run on the GPU and make lowering fail
// with rt::Unsupported (never a silent CPU fallback).
//
// Where the reference panics (bvh.rs:30,58 "No bounding box in BVHNode"), this mirror
// throws rt::Panic; the C bindings (rt_host_c.cpp) turn exceptions into error codes.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rtmi.h"

namespace rt {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- nalgebra::Vector3<f64> (the operations the path uses) ----------------------------
struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Vec3 operator+(const Vec3 &o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(const Vec3 &o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator-() const { return {-x, -y, -z}; }
    Vec3 operator*(double s) const { return {x * s, y * s, z * s}; }
    Vec3 operator/(double s) const { return {x / s, y / s, z / s}; }
    Vec3 zip_mul(const Vec3 &o) const { return {x * o.x, y * o.y, z * o.z}; }
    double dot(const Vec3 &o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3 cross(const Vec3 &o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    double magnitude_squared() const { return dot(*this); }
    double magnitude() const { return std::sqrt(dot(*this)); }
    double norm() const { return magnitude(); }
    Vec3 normalize() const { return *this / norm(); }
};
inline Vec3 operator*(double s, const Vec3 &v) { return v * s; }

// ---- host RNG: the reference's rand::thread_rng(), made seedable ----------------------
// Philox4x32-10 stream; stream_id 1 = scene construction, 0 = render path (see philox.py).
class Rng {
  public:
    void seed(uint64_t seed, uint32_t sample, uint32_t pixel, uint32_t stream_id);
    uint32_t next_u32();
    double gen() { return (double)(next_u32() >> 8) * (1.0 / 16777216.0); } // rng.gen::<f64>()
    uint32_t gen_range(uint32_t n) { return (uint32_t)(((uint64_t)next_u32() * n) >> 32); } // gen_range(0..n)
  private:
    uint32_t key_[2] = {0, 0}, ctr_[4] = {0, 0, 0, 0}, buf_[4] = {0, 0, 0, 0};
    int pos_ = 4;
};
Rng &scene_rng();  // BVHNode::new (bvh.rs:40), Perlin::new (perlin.rs:7,18-20)
Rng &render_rng(); // the per-sample stream when evaluating on the CPU
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

// ---- src/ray.rs ---------------------------------------------------------------------------
class Ray {
  public:
    Ray() = default;
    Ray(const Vec3 &origin, const Vec3 &direction, double time) : origin_(origin), direction_(direction), time_(time) {}
    Vec3 origin() const { return origin_; }
    Vec3 direction() const { return direction_; }
    Vec3 pointing_at(double t) const { return origin_ + t * direction_; }
    double time() const { return time_; }
  private:
    Vec3 origin_, direction_;
    double time_ = 0;
};

// ---- src/aabb.rs --------------------------------------------------------------------------
struct AABB {
    Vec3 min, max;
    AABB() = default;
    AABB(const Vec3 &mn, const Vec3 &mx) : min(mn), max(mx) {}
    bool hit(const Ray &ray, double t_min, double t_max) const;
};
AABB surrounding_box(const AABB &box0, const AABB &box1);

class Material;
class SceneBuilder;

// ---- src/hittable.rs ----------------------------------------------------------------------
struct HitRecord {
    double t = 0, u = 0, v = 0;
    Vec3 p, normal;
    const Material *material = nullptr;
};

class Texture {
  public:
    virtual ~Texture() = default;
    virtual Vec3 value(double u, double v, const Vec3 &p) const = 0;
    virtual int lower(SceneBuilder &b) const; // default: Unsupported
};

class Material {
  public:
    virtual ~Material() = default;
    virtual std::optional<std::pair<Ray, Vec3>> scatter(const Ray &ray, const HitRecord &hit) const = 0;
    virtual Vec3 emitted(double u, double v, const Vec3 &p) const = 0;
    virtual int lower(SceneBuilder &b) const; // default: Unsupported
};

class Hittable {
  public:
    virtual ~Hittable() = default;
    virtual std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const = 0;
    virtual std::optional<AABB> bounding_box(double t0, double t1) const = 0;
};

using HittablePtr = std::shared_ptr<const Hittable>;
using MaterialPtr = std::shared_ptr<const Material>;
using TexturePtr = std::shared_ptr<const Texture>;

class HittableList : public Hittable {
  public:
    void push(HittablePtr h) { list_.push_back(std::move(h)); }
    const std::vector<HittablePtr> &items() const { return list_; }
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
  private:
    std::vector<HittablePtr> list_;
};

class FlipNormals : public Hittable {
  public:
    explicit FlipNormals(HittablePtr h) : hittable_(std::move(h)) {}
    const HittablePtr &inner() const { return hittable_; }
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
  private:
    HittablePtr hittable_;
};

// ---- src/sphere.rs ------------------------------------------------------------------------
class Sphere : public Hittable {
  public:
    Sphere(const Vec3 &center, double radius, MaterialPtr material)
        : center_(center), radius_(radius), material_(std::move(material)) {}
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    Vec3 center_; double radius_; MaterialPtr material_;
};
class MovingSphere : public Hittable {
  public:
    MovingSphere(const Vec3 &center0, const Vec3 &center1, double time0, double time1, double radius, MaterialPtr material)
        : center0_(center0), center1_(center1), time0_(time0), time1_(time1), radius_(radius), material_(std::move(material)) {}
    Vec3 center(double time) const;
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    Vec3 center0_, center1_; double time0_, time1_, radius_; MaterialPtr material_;
};

// ---- src/rect.rs, src/cube.rs -------------------------------------------------------------
enum class Plane { YZ = 0, ZX = 1, XY = 2 };
class Rect : public Hittable {
  public:
    Rect(Plane plane, double x0, double y0, double x1, double y1, double k, MaterialPtr material)
        : plane_(plane), x0_(x0), y0_(y0), x1_(x1), y1_(y1), k_(k), material_(std::move(material)) {}
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    Plane plane_; double x0_, y0_, x1_, y1_, k_; MaterialPtr material_;
};
class Cube : public Hittable {
  public:
    Cube(const Vec3 &p_min, const Vec3 &p_max, MaterialPtr material);
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    Vec3 p_min_, p_max_; MaterialPtr material_;
  private:
    HittableList sides_;
};

// ---- src/traslate.rs, src/rotate.rs, src/medium.rs ----------------------------------------
class Traslate : public Hittable {
  public:
    Traslate(HittablePtr hitable, const Vec3 &offset) : hitable_(std::move(hitable)), offset_(offset) {}
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    HittablePtr hitable_; Vec3 offset_;
};
enum class Axis { X = 0, Y = 1, Z = 2 };
class Rotate : public Hittable {
  public:
    Rotate(Axis axis, HittablePtr hittable, double angle);
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    Axis axis_; double sin_theta_, cos_theta_; HittablePtr hittable_; std::optional<AABB> bbox_;
};
class ConstantMedium : public Hittable {
  public:
    ConstantMedium(HittablePtr boundary, double density, TexturePtr texture);
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    HittablePtr boundary_; double density_; MaterialPtr phase_function_;
};

// ---- src/bvh.rs ---------------------------------------------------------------------------
class BVHNode : public Hittable {
  public:
    // BVHNode::new(&mut [Rc<dyn Hittable>], time0, time1) — sorts `hittable` in place like the reference
    BVHNode(std::vector<HittablePtr> &hittable, size_t begin, size_t end, double time0, double time1);
    BVHNode(std::vector<HittablePtr> &hittable, double time0, double time1)
        : BVHNode(hittable, 0, hittable.size(), time0, time1) {}
    std::optional<HitRecord> hit(const Ray &ray, double t_min, double t_max) const override;
    std::optional<AABB> bounding_box(double t0, double t1) const override;
    HittablePtr left_, right_; AABB bbox_;
};

// ---- src/perlin.rs, src/texture.rs --------------------------------------------------------
class Perlin {
  public:
    Perlin(); // draws from scene_rng()
    double noise(const Vec3 &p) const;
    double turb(const Vec3 &p, size_t depth) const;
    std::vector<Vec3> ran_vec_; std::vector<size_t> perm_x_, perm_y_, perm_z_;
};
class SolidTexture : public Texture {
  public:
    SolidTexture(double r, double g, double b) : color_(r, g, b) {}
    Vec3 value(double u, double v, const Vec3 &p) const override;
    int lower(SceneBuilder &b) const override;
    Vec3 color_;
};
class CheckerTexture : public Texture {
  public:
    CheckerTexture(TexturePtr odd, TexturePtr even) : odd_(std::move(odd)), even_(std::move(even)) {}
    Vec3 value(double u, double v, const Vec3 &p) const override;
    int lower(SceneBuilder &b) const override;
    TexturePtr odd_, even_;
};
class NoiseTexture : public Texture {
  public:
    explicit NoiseTexture(double scale) : scale_(scale) {}
    Vec3 value(double u, double v, const Vec3 &p) const override;
    int lower(SceneBuilder &b) const override;
    Perlin noise_; double scale_;
};
class ImageTexture : public Texture {
  public:
    ImageTexture(std::vector<uint8_t> data, uint32_t nx, uint32_t ny) : data_(std::move(data)), nx_(nx), ny_(ny) {}
    Vec3 value(double u, double v, const Vec3 &p) const override;
    int lower(SceneBuilder &b) const override;
    std::vector<uint8_t> data_; uint32_t nx_, ny_;
};

// ---- src/material.rs ----------------------------------------------------------------------
class Lambertian : public Material {
  public:
    explicit Lambertian(TexturePtr albedo) : albedo_(std::move(albedo)) {}
    std::optional<std::pair<Ray, Vec3>> scatter(const Ray &ray, const HitRecord &hit) const override;
    Vec3 emitted(double, double, const Vec3 &) const override { return {}; }
    int lower(SceneBuilder &b) const override;
    TexturePtr albedo_;
};
class Metal : public Material {
  public:
    Metal(TexturePtr albedo, double fuzz) : albedo_(std::move(albedo)), fuzz_(fuzz < 1.0 ? fuzz : 1.0) {}
    std::optional<std::pair<Ray, Vec3>> scatter(const Ray &ray, const HitRecord &hit) const override;
    Vec3 emitted(double, double, const Vec3 &) const override { return {}; }
    int lower(SceneBuilder &b) const override;
    TexturePtr albedo_; double fuzz_;
};
class Dielectric : public Material {
  public:
    explicit Dielectric(double ref_idx) : ref_idx_(ref_idx) {}
    std::optional<std::pair<Ray, Vec3>> scatter(const Ray &ray, const HitRecord &hit) const override;
    Vec3 emitted(double, double, const Vec3 &) const override { return {}; }
    int lower(SceneBuilder &b) const override;
    double ref_idx_;
};
class DiffuseLight : public Material {
  public:
    explicit DiffuseLight(TexturePtr emit) : emit_(std::move(emit)) {}
    std::optional<std::pair<Ray, Vec3>> scatter(const Ray &, const HitRecord &) const override { return std::nullopt; }
    Vec3 emitted(double u, double v, const Vec3 &p) const override { return emit_->value(u, v, p); }
    int lower(SceneBuilder &b) const override;
    TexturePtr emit_;
};
class Isotropic : public Material {
  public:
    explicit Isotropic(TexturePtr albedo) : albedo_(std::move(albedo)) {}
    std::optional<std::pair<Ray, Vec3>> scatter(const Ray &ray, const HitRecord &hit) const override;
    Vec3 emitted(double, double, const Vec3 &) const override { return {}; }
    int lower(SceneBuilder &b) const override;
    TexturePtr albedo_;
};

// ---- src/util.rs --------------------------------------------------------------------------
Vec3 random_in_unit_sphere();
Vec3 random_in_unit_disk();

// ---- src/color.rs -------------------------------------------------------------------------
Vec3 color(const Ray &ray, const Hittable &world, size_t depth);
// Opt-in extension, off by default: a missing ray returns the gradient the reference keeps commented out at
// color.rs:18-20 instead of black (:21).  Thread-local switch for the CPU evaluation; the device takes
// RTMI_FLAG_SKY in RenderOptions::flags.
void set_sky_background(bool on);
bool sky_background();
// Two more opt-in extensions for the CPU evaluation, off by default (device: RTMI_FLAG_FACE_FORWARD, RTMI_FLAG_UV_BOOK):
// opaque materials scatter about the normal turned against the ray (the reference never turns it, sphere.rs:50,
// rect.rs:58-59); get_sphere_uv with the book's pi/2 instead of FRAC_2_PI (sphere.rs:13).
void set_face_forward(bool on);
void set_uv_book(bool on);

// ---- lowering to the flat device scene ----------------------------------------------------
struct LoweredScene {
    std::vector<rtmi_item> items;
    std::vector<float> prim_a, prim_b;
    std::vector<rtmi_prim_meta> prim_meta;
    std::vector<float> prim_gate;  // 8 floats per primitive: box of its parent BVHNode in the reference tree (rtmi.h)
    std::vector<AABB> prim_box;    // host only: true extent of a BVH primitive (its alternative tree is built on it)
    std::vector<char> prim_has_box;
    uint32_t alt_max_depth = 0;    // deepest alternative tree (in 4-wide nodes)
    std::vector<rtmi_bvh4_node> alt_nodes; // the alternative trees
    std::vector<rtmi_bvh_node> nodes;
    std::vector<rtmi_xform> xforms;
    std::vector<rtmi_material> materials;
    std::vector<rtmi_texture> textures;
    std::vector<rtmi_perlin> perlin;
    std::vector<rtmi_image> images;
    std::vector<uint8_t> image_data;
    uint32_t max_bvh_depth = 0;
    // ray times for which every MovingSphere inside a BVH stays inside the boxes built for it (its own
    // [time0, time1]); outside, librtmi falls back from pruned to exact traversal
    float bvh_time_lo = -3.40282346638528859811704183484516925e+38f, bvh_time_hi = 3.40282346638528859811704183484516925e+38f;
    rtmi_scene_desc desc() const;
};

#define RTMI_NO_SUBTREE (-1) /* lower_bvh: nothing but media below (they become deferred items).  Not a child reference:
                             * -1 would be the leaf of type 7, primitive 2^28 - 1 (INT32_MIN IS one: the sphere leaf of primitive 0) */
class SceneBuilder {
  public:
    int texture_index(const Texture *t);   // lowers on first use
    int material_index(const Material *m); // lowers on first use
    int add_texture(const rtmi_texture &t) { out.textures.push_back(t); return (int)out.textures.size() - 1; }
    int add_material(const rtmi_material &m) { out.materials.push_back(m); return (int)out.materials.size() - 1; }
    bool texture_needs_uv(int tex) const;
    void lower_world(const Hittable &world);
    LoweredScene out;
  private:
    // a ConstantMedium that was a child of a BVHNode, lowered as an item of its own behind the BVH item (rt_host.cpp)
    struct PendingMedium { const Hittable *obj; AABB gate; int32_t rank; bool flip; }; // rank: primitives pushed before it (in-order position)
    struct DeferredMedium { AABB gate; int32_t chain_first; int32_t chain_count; bool save_t0; int32_t rank; bool flip; uint32_t scan = 0u; }; // scan: LISTSCAN flags of a member
    void lower_scan_group(const Hittable &top, const DeferredMedium &deferred);
    std::vector<PendingMedium> pending_media_;
    void collect_media(const Hittable *h, const BVHNode &parent, bool flip_all);
    void lower_item(const Hittable &h, const DeferredMedium *deferred = nullptr);
    int push_prim(const Hittable &h, bool flip, bool force_moving);
    int32_t lower_bvh(const BVHNode &n, uint32_t depth, bool force_moving, double pad, bool unbounded_leaves, bool flip_all);
    int32_t lower_leaf(const Hittable &h, const BVHNode &n, bool flip, bool force_moving, double pad, bool unbounded_leaves, AABB &lb);
    int32_t lower_list_leaf(const HittableList &l, const BVHNode &holder, uint32_t depth, bool flip, bool force_moving, double pad,
                            bool unbounded_leaves, AABB &box_out);
    int32_t build_alt_tree(std::vector<int> &prims, size_t lo, size_t hi, uint32_t depth, double pad, AABB *box_out);
#define RTMI_ALT_PLAN_H 16 /* height budgets the collapse plan is computed for */
    std::vector<double> alt_forest_; // [binary node][height level][slots 0..4]: cost of the cheapest collapse (plan_collapse)
    std::vector<int8_t> alt_split_;  // ... and the split that achieves it (slot 0: the node's own share-out)
    int alt_plan_levels_ = 1;        // 1 = planned without the height bound
    int alt_plan_height_ = 0;
    void plan_collapse(int32_t root);
    int32_t collapse_alt(int32_t ref, uint32_t depth, int height); // binary scratch tree -> 4-wide nodes in out.alt_nodes
    std::vector<rtmi_bvh_node> alt_scratch_;            // binary SAH tree being built for the current item
    int run_item_ = -1; // index of the item that collects the current run of plain top-level primitives
    std::map<const Texture *, int> tex_ids_;
    std::map<const Material *, int> mat_ids_;
};
LoweredScene lower_scene(const Hittable &world);

// ---- src/camera.rs + the render entry points -----------------------------------------------
struct RenderOptions {
    uint64_t seed = 42;
    uint32_t max_depth = 50; // color.rs:9
    double t_min = 0.001;    // color.rs:7
    uint32_t flags = 0;
    uint32_t spp_chunks = 0;
    int device = 0;
    std::vector<int> devices; // non-empty: render on these GPUs of this process (tiles t % n, one gather); `device` is ignored
    // Called from the rendering thread about every 50 ms with (work units done, total); return false to cancel.
    // Replaces the reference's stand-alone bar (src/progressbar.rs:6-58).
    std::function<bool(uint64_t, uint64_t)> progress;
};
struct Image {
    uint32_t nx = 0, ny = 0;
    std::vector<float> linear;  // ny*nx*3, row 0 = top
    std::vector<uint8_t> rgb8;  // ny*nx*3
    rtmi_stats stats{};
    std::string to_ppm() const; // the String create_image returns (tests/test.rs:58-84)
    // streamed to a file without that string: the same P3 text, or binary P6 (rtmi_write_ppm)
    void write_ppm(const std::string &path, bool binary = false) const;
};
class Camera {
  public:
    Camera(const Vec3 &look_from, const Vec3 &look_at, const Vec3 &view_up, double vertical_fov, double aspect,
           double aperture, double focus_dist, double time0, double time1);
    Ray get_ray(double s, double t) const;
    rtmi_camera lower() const;
    // The addition the north star asks for: the triple loop of create_image on the GPU.
    Image render(const Hittable &world, uint32_t nx, uint32_t ny, uint32_t ns, const RenderOptions &opt = {}) const;
    Vec3 origin_, lower_left_corner_, horizontal_, vertical_, u_, v_;
    double time0_, time1_, lens_radius_;
};
// A world kept resident on a list of GPUs of this process (rtmi_multi_create / _render / _destroy): lowered and
// uploaded once, then any number of Camera views rendered from it — every render costs the kernels, one gather and the
// un-tiling, not the uploads.  Camera::render with RenderOptions::devices is DeviceScene(world, devices).render(...)
// for a single image.  Not copyable; render() may be called from several threads (calls on one scene serialise).
class DeviceScene {
  public:
    DeviceScene(const Hittable &world, const std::vector<int> &devices);
    ~DeviceScene();
    DeviceScene(const DeviceScene &) = delete;
    DeviceScene &operator=(const DeviceScene &) = delete;
    // allocates the per-sample buffers of renders of this size now instead of in the first render() (optional)
    void prepare(uint32_t nx, uint32_t ny, uint32_t ns, const RenderOptions &opt = {});
    Image render(const Camera &cam, uint32_t nx, uint32_t ny, uint32_t ns, const RenderOptions &opt = {});
    const std::vector<int> &devices() const { return devices_; }
    // RTMI_COLLECTIVE_*: what brings the tiles of a render together (one ncclGather for distinct devices)
    int collective() const { return rtmi_multi_collective(handle_); }

  private:
    std::vector<int> devices_;
    rtmi_multi *handle_ = nullptr;
};
// tests/test.rs:55-85 — note the argument order (ny, nx, ns, cam, world)
std::string create_image(size_t ny, size_t nx, size_t ns, const Camera &cam, const Hittable &world,
                         const RenderOptions &opt = {});

} // namespace rt
