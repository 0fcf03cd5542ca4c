// rt_host.cpp — implementation of the C++ host mirror (see rt_host.hpp).
// CPU evaluation restates the reference in f64 (file:line cited per function); lowering
// produces the flat scene of include/rtmi.h; Camera::render / create_image call the C ABI.
#include "rt_host.hpp"

#include <algorithm>
#include <cstring>

namespace rt {

static constexpr double kPi = 3.14159265358979323846264338327950288;
static constexpr double kFrac2Pi = 0.636619772367581343075535053490057448; // std::f64::consts::FRAC_2_PI
static constexpr double kF64Max = 1.79769313486231570814527423731704357e+308;

// ---- Philox / host RNG -------------------------------------------------------------------
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void Rng::seed(uint64_t seed, uint32_t sample, uint32_t pixel, uint32_t stream_id) {
    key_[0] = (uint32_t)seed; key_[1] = (uint32_t)(seed >> 32);
    ctr_[0] = 0; ctr_[1] = sample; ctr_[2] = pixel; ctr_[3] = stream_id;
    pos_ = 4;
}
uint32_t Rng::next_u32() {
    if (pos_ == 4) {
        philox4x32_10(ctr_, key_, buf_);
        ctr_[0]++;
        pos_ = 0;
    }
    return buf_[pos_++];
}
Rng &scene_rng() {
    static thread_local Rng r = [] { Rng x; x.seed(1, 0, 0, 1); return x; }();
    return r;
}
Rng &render_rng() {
    static thread_local Rng r = [] { Rng x; x.seed(42, 0, 0, 0); return x; }();
    return r;
}

// ---- src/util.rs:4-24 --------------------------------------------------------------------
Vec3 random_in_unit_sphere() {
    Rng &rng = render_rng();
    const Vec3 unit(1.0, 1.0, 1.0);
    for (;;) {
        const double x = rng.gen(), y = rng.gen(), z = rng.gen();
        const Vec3 p = 2.0 * Vec3(x, y, z) - unit;
        if (p.magnitude_squared() < 1.0) return p;
    }
}
Vec3 random_in_unit_disk() {
    Rng &rng = render_rng();
    const Vec3 unit(1.0, 1.0, 0.0);
    for (;;) {
        const double x = rng.gen(), y = rng.gen();
        const Vec3 p = 2.0 * Vec3(x, y, 0.0) - unit;
        if (p.magnitude_squared() < 1.0) return p;
    }
}

// ---- src/aabb.rs:6-18, 31-44 -------------------------------------------------------------
AABB surrounding_box(const AABB &a, const AABB &b) {
    return AABB(Vec3(std::fmin(a.min.x, b.min.x), std::fmin(a.min.y, b.min.y), std::fmin(a.min.z, b.min.z)),
                Vec3(std::fmax(a.max.x, b.max.x), std::fmax(a.max.y, b.max.y), std::fmax(a.max.z, b.max.z)));
}
bool AABB::hit(const Ray &ray, double t_min, double t_max) const {
    for (int a = 0; a < 3; a++) {
        const double inv_d = 1.0 / ray.direction()[a];
        double t0 = (min[a] - ray.origin()[a]) * inv_d;
        double t1 = (max[a] - ray.origin()[a]) * inv_d;
        if (inv_d < 0.0) std::swap(t0, t1);
        t_min = std::fmax(t_min, t0);
        t_max = std::fmin(t_max, t1);
        if (t_max <= t_min) return false;
    }
    return true;
}

// ---- src/hittable.rs:37-64, 78-87 --------------------------------------------------------
std::optional<HitRecord> HittableList::hit(const Ray &ray, double t_min, double t_max) const {
    double closest_so_far = t_max;
    std::optional<HitRecord> hit_anything;
    for (const auto &h : list_) {
        if (auto hit = h->hit(ray, t_min, closest_so_far)) {
            closest_so_far = hit->t;
            hit_anything = hit;
        }
    }
    return hit_anything;
}
std::optional<AABB> HittableList::bounding_box(double t0, double t1) const {
    if (list_.empty()) return std::nullopt;
    auto acc = list_[0]->bounding_box(t0, t1);
    if (!acc) return std::nullopt;
    for (size_t i = 1; i < list_.size(); i++) {
        auto b = list_[i]->bounding_box(t0, t1);
        if (!b) return std::nullopt;
        acc = surrounding_box(*acc, *b);
    }
    return acc;
}
std::optional<HitRecord> FlipNormals::hit(const Ray &ray, double t_min, double t_max) const {
    auto hit = hittable_->hit(ray, t_min, t_max);
    if (hit) hit->normal = -hit->normal;
    return hit;
}
std::optional<AABB> FlipNormals::bounding_box(double t0, double t1) const { return hittable_->bounding_box(t0, t1); }

// ---- src/sphere.rs -----------------------------------------------------------------------
static thread_local bool g_uv_book = false; // opt-in RTMI_FLAG_UV_BOOK for the CPU evaluation
void set_uv_book(bool on) { g_uv_book = on; }
static std::pair<double, double> get_sphere_uv(const Vec3 &p) { // :9-15
    const double phi = std::atan2(p.z, p.x);
    const double theta = std::asin(p.y);
    const double u = 1.0 - (phi + kPi) / (2.0 * kPi);
    const double v = (theta + (g_uv_book ? kPi / 2.0 : kFrac2Pi)) / kPi; // FRAC_2_PI: sic (the book adds pi/2)
    return {u, v};
}
static std::optional<HitRecord> sphere_hit(const Vec3 &center, double radius, const Material *mat, const Ray &ray,
                                           double t_min, double t_max) { // :37-77 / :122-164
    const Vec3 oc = ray.origin() - center;
    const double a = ray.direction().dot(ray.direction());
    const double b = oc.dot(ray.direction());
    const double c = oc.dot(oc) - radius * radius;
    const double discriminant = b * b - a * c;
    if (discriminant > 0.0) {
        const double sq = std::sqrt(discriminant);
        for (int root = 0; root < 2; root++) {
            const double t = root == 0 ? (-b - sq) / a : (-b + sq) / a;
            if (t < t_max && t > t_min) {
                HitRecord r;
                r.t = t;
                r.p = ray.pointing_at(t);
                r.normal = (r.p - center) / radius;
                std::tie(r.u, r.v) = get_sphere_uv(r.normal);
                r.material = mat;
                return r;
            }
        }
    }
    return std::nullopt;
}
std::optional<HitRecord> Sphere::hit(const Ray &ray, double t_min, double t_max) const {
    return sphere_hit(center_, radius_, material_.get(), ray, t_min, t_max);
}
std::optional<AABB> Sphere::bounding_box(double, double) const { // :79-84
    const Vec3 r(radius_, radius_, radius_);
    return AABB(center_ - r, center_ + r);
}
Vec3 MovingSphere::center(double time) const { // :115-118
    return center0_ + ((time - time0_) / (time1_ - time0_)) * (center1_ - center0_);
}
std::optional<HitRecord> MovingSphere::hit(const Ray &ray, double t_min, double t_max) const {
    return sphere_hit(center(ray.time()), radius_, material_.get(), ray, t_min, t_max);
}
std::optional<AABB> MovingSphere::bounding_box(double t0, double t1) const { // :165-174
    const Vec3 r(radius_, radius_, radius_);
    return surrounding_box(AABB(center(t0) - r, center(t0) + r), AABB(center(t1) - r, center(t1) + r));
}

// ---- src/rect.rs -------------------------------------------------------------------------
static void plane_axes(int plane, int &k, int &a, int &b) { // :40-44 ; Axis uses the same triples (rotate.rs:13-19)
    if (plane == 0) { k = 0; a = 1; b = 2; }
    else if (plane == 1) { k = 1; a = 2; b = 0; }
    else { k = 2; a = 0; b = 1; }
}
std::optional<HitRecord> Rect::hit(const Ray &ray, double t_min, double t_max) const { // :39-69
    int ka, aa, ba;
    plane_axes((int)plane_, ka, aa, ba);
    const double t = (k_ - ray.origin()[ka]) / ray.direction()[ka];
    if (t < t_min || t > t_max) return std::nullopt;
    const double x = ray.origin()[aa] + t * ray.direction()[aa];
    const double y = ray.origin()[ba] + t * ray.direction()[ba];
    if (x < x0_ || x > x1_ || y < y0_ || y > y1_) return std::nullopt;
    HitRecord r;
    r.u = (x - x0_) / (x1_ - x0_);
    r.v = (y - y0_) / (y1_ - y0_);
    r.t = t;
    r.p = ray.pointing_at(t);
    r.normal = Vec3(0, 0, 0);
    r.normal[ka] = 1.0;
    r.material = material_.get();
    return r;
}
std::optional<AABB> Rect::bounding_box(double, double) const { // :71-75 (sic: ignores the plane)
    return AABB(Vec3(x0_, y0_, k_ - 0.0001), Vec3(x1_, y1_, k_ + 0.0001));
}

// ---- src/cube.rs:15-94 -------------------------------------------------------------------
Cube::Cube(const Vec3 &p_min, const Vec3 &p_max, MaterialPtr material)
    : p_min_(p_min), p_max_(p_max), material_(std::move(material)) {
    sides_.push(std::make_shared<Rect>(Plane::XY, p_min.x, p_min.y, p_max.x, p_max.y, p_max.z, material_));
    sides_.push(std::make_shared<Rect>(Plane::XY, p_min.x, p_min.y, p_max.x, p_max.y, p_min.z, material_));
    sides_.push(std::make_shared<Rect>(Plane::ZX, p_min.z, p_min.x, p_max.z, p_max.x, p_max.y, material_));
    sides_.push(std::make_shared<Rect>(Plane::ZX, p_min.z, p_min.x, p_max.z, p_max.x, p_min.y, material_));
    sides_.push(std::make_shared<Rect>(Plane::YZ, p_min.y, p_min.z, p_max.y, p_max.z, p_max.x, material_));
    sides_.push(std::make_shared<Rect>(Plane::YZ, p_min.y, p_min.z, p_max.y, p_max.z, p_min.x, material_));
}
std::optional<HitRecord> Cube::hit(const Ray &ray, double t_min, double t_max) const { return sides_.hit(ray, t_min, t_max); }
std::optional<AABB> Cube::bounding_box(double, double) const { return AABB(p_min_, p_max_); }

// ---- src/traslate.rs:18-32 ---------------------------------------------------------------
std::optional<HitRecord> Traslate::hit(const Ray &ray, double t_min, double t_max) const {
    const Ray moved(ray.origin() - offset_, ray.direction(), ray.time());
    auto hit = hitable_->hit(moved, t_min, t_max);
    if (hit) hit->p = hit->p + offset_;
    return hit;
}
std::optional<AABB> Traslate::bounding_box(double t0, double t1) const {
    auto b = hitable_->bounding_box(t0, t1);
    if (b) { b->min = b->min + offset_; b->max = b->max + offset_; }
    return b;
}

// ---- src/rotate.rs:30-118 ----------------------------------------------------------------
Rotate::Rotate(Axis axis, HittablePtr hittable, double angle) : axis_(axis), hittable_(std::move(hittable)) {
    const double radians = (kPi / 180.0) * angle;
    sin_theta_ = std::sin(radians);
    cos_theta_ = std::cos(radians);
    bbox_ = hittable_->bounding_box(0.0, 1.0);
    if (bbox_) {
        // :36-37 start max at f64::MAX and min at f64::MIN, so no update in :38-76 ever fires
        bbox_->min = Vec3(-kF64Max, -kF64Max, -kF64Max);
        bbox_->max = Vec3(kF64Max, kF64Max, kF64Max);
    }
}
std::optional<HitRecord> Rotate::hit(const Ray &ray, double t_min, double t_max) const {
    int ra, a, b;
    plane_axes((int)axis_, ra, a, b);
    Vec3 origin = ray.origin(), direction = ray.direction();
    origin[a] = cos_theta_ * ray.origin()[a] + sin_theta_ * ray.origin()[b];
    origin[b] = -sin_theta_ * ray.origin()[a] + cos_theta_ * ray.origin()[b];
    direction[a] = cos_theta_ * ray.direction()[a] + sin_theta_ * ray.direction()[b];
    direction[b] = -sin_theta_ * ray.direction()[a] + cos_theta_ * ray.direction()[b];
    const Ray rotated(origin, direction, ray.time());
    auto hit = hittable_->hit(rotated, t_min, t_max);
    if (hit) {
        Vec3 p = hit->p, n = hit->normal;
        p[a] = cos_theta_ * hit->p[a] - sin_theta_ * hit->p[b];
        p[b] = sin_theta_ * hit->p[a] + cos_theta_ * hit->p[b];
        n[a] = cos_theta_ * hit->normal[a] - sin_theta_ * hit->normal[b];
        n[b] = sin_theta_ * hit->normal[a] + cos_theta_ * hit->normal[b];
        hit->p = p;
        hit->normal = n;
    }
    return hit;
}
std::optional<AABB> Rotate::bounding_box(double, double) const { return bbox_; }

// ---- src/medium.rs:16-60 -----------------------------------------------------------------
ConstantMedium::ConstantMedium(HittablePtr boundary, double density, TexturePtr texture)
    : boundary_(std::move(boundary)), density_(density), phase_function_(std::make_shared<Isotropic>(std::move(texture))) {}
std::optional<HitRecord> ConstantMedium::hit(const Ray &ray, double t_min, double t_max) const {
    Rng &rng = render_rng();
    if (auto hit1 = boundary_->hit(ray, -kF64Max, kF64Max)) {
        if (auto hit2 = boundary_->hit(ray, hit1->t + 0.0001, kF64Max)) {
            if (hit1->t < t_min) hit1->t = t_min;
            if (hit2->t > t_max) hit2->t = t_max;
            if (hit1->t < hit2->t) {
                const double distance_inside_boundary = (hit2->t - hit1->t) * ray.direction().norm();
                const double hit_distance = -(1.0 / density_) * std::log(rng.gen());
                if (hit_distance < distance_inside_boundary) {
                    HitRecord r;
                    r.t = hit1->t + hit_distance / ray.direction().norm();
                    r.u = 0.0; r.v = 0.0;
                    r.p = ray.pointing_at(r.t);
                    r.normal = Vec3(1.0, 0.0, 0.0);
                    r.material = phase_function_.get();
                    return r;
                }
            }
        }
    }
    return std::nullopt;
}
std::optional<AABB> ConstantMedium::bounding_box(double t0, double t1) const { return boundary_->bounding_box(t0, t1); }

// ---- src/bvh.rs:17-93 --------------------------------------------------------------------
// sort_unstable_by with the reference's Less/Greater-only comparator (:32-36) leaves the
// order of equal keys unspecified; this mirror sorts stably on `a.min[axis] - b.min[axis] < 0`.
BVHNode::BVHNode(std::vector<HittablePtr> &hittable, size_t begin, size_t end, double time0, double time1) {
    const size_t len = end - begin;
    if (len == 0) throw Panic("BVHNode::new on an empty slice");
    const uint32_t axis = scene_rng().gen_range(3);
    std::vector<std::pair<double, HittablePtr>> keyed;
    keyed.reserve(len);
    for (size_t i = begin; i < end; i++) {
        auto b = hittable[i]->bounding_box(time0, time1);
        if (!b) throw Panic("No bounding box in BVHNode");
        keyed.emplace_back(b->min[(int)axis], hittable[i]);
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto &a, const auto &b) { return a.first - b.first < 0.0; });
    for (size_t i = 0; i < len; i++) hittable[begin + i] = keyed[i].second;
    if (len == 1) { left_ = hittable[begin]; right_ = hittable[begin]; }
    else if (len == 2) { left_ = hittable[begin]; right_ = hittable[begin + 1]; }
    else {
        left_ = std::make_shared<BVHNode>(hittable, begin, begin + len / 2, time0, time1);
        right_ = std::make_shared<BVHNode>(hittable, begin + len / 2, end, time0, time1);
    }
    auto lb = left_->bounding_box(time0, time1), rb = right_->bounding_box(time0, time1);
    if (!lb || !rb) throw Panic("No bounding box in BVHNode");
    bbox_ = surrounding_box(*lb, *rb);
}
std::optional<HitRecord> BVHNode::hit(const Ray &ray, double t_min, double t_max) const {
    if (bbox_.hit(ray, t_min, t_max)) {
        auto l = left_->hit(ray, t_min, t_max);
        auto r = right_->hit(ray, t_min, t_max);
        if (l && r) return (l->t < r->t) ? l : r;
        if (l) return l;
        if (r) return r;
    }
    return std::nullopt;
}
std::optional<AABB> BVHNode::bounding_box(double, double) const { return bbox_; }

// ---- src/perlin.rs -----------------------------------------------------------------------
Perlin::Perlin() {
    Rng &rng = scene_rng();
    ran_vec_.reserve(256);
    for (int i = 0; i < 256; i++) { // perlin_generate :12-26
        const double x = -1.0 + 2.0 * rng.gen(), y = -1.0 + 2.0 * rng.gen(), z = -1.0 + 2.0 * rng.gen();
        ran_vec_.push_back(Vec3(x, y, z).normalize());
    }
    auto perm = [&rng]() { // perlin_generate_perm :28-36 + permute :4-10
        std::vector<size_t> p(256);
        for (size_t i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i >= 0; i--) std::swap(p[(size_t)i], p[rng.gen_range((uint32_t)i + 1)]);
        return p;
    };
    perm_x_ = perm(); perm_y_ = perm(); perm_z_ = perm();
}
static uint64_t as_usize(double x) { // Rust `f64 as usize`
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}
double Perlin::noise(const Vec3 &p) const { // :76-97 + perlin_interpolation :38-56
    const double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
    const uint64_t i = as_usize(std::floor(p.x)), j = as_usize(std::floor(p.y)), k = as_usize(std::floor(p.z));
    const double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (uint64_t di = 0; di < 2; di++)
        for (uint64_t dj = 0; dj < 2; dj++)
            for (uint64_t dk = 0; dk < 2; dk++) {
                const Vec3 &c = ran_vec_[perm_x_[(i + di) & 255] ^ perm_y_[(j + dj) & 255] ^ perm_z_[(k + dk) & 255]];
                const Vec3 weight(u - (double)di, v - (double)dj, w - (double)dk);
                accum += ((double)di * uu + (double)(1 - di) * (1.0 - uu)) * ((double)dj * vv + (double)(1 - dj) * (1.0 - vv)) *
                         ((double)dk * ww + (double)(1 - dk) * (1.0 - ww)) * c.dot(weight);
            }
    return accum;
}
double Perlin::turb(const Vec3 &p, size_t depth) const { // :99-109
    double accum = 0.0, weight = 1.0;
    Vec3 temp_p = p;
    for (size_t i = 0; i < depth; i++) {
        accum += weight * noise(temp_p);
        weight *= 0.5;
        temp_p = temp_p * 2.0;
    }
    return std::fabs(accum);
}

// ---- src/texture.rs ----------------------------------------------------------------------
Vec3 SolidTexture::value(double, double, const Vec3 &) const { return color_; }
Vec3 CheckerTexture::value(double u, double v, const Vec3 &p) const { // :39-48
    const double s = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
    return s < 0.0 ? odd_->value(u, v, p) : even_->value(u, v, p);
}
Vec3 NoiseTexture::value(double, double, const Vec3 &p) const { // :65-71
    return Vec3(1.0, 1.0, 1.0) * 0.5 * (1.0 + std::sin(scale_ * p.x + 5.0 * noise_.turb(p, 7)));
}
Vec3 ImageTexture::value(double u, double v, const Vec3 &) const { // :86-108
    const uint64_t nx = nx_, ny = ny_;
    uint64_t i = as_usize(u * (double)nx), j = as_usize((1.0 - v) * (double)ny);
    if (i > nx - 1) i = nx - 1;
    if (j > ny - 1) j = ny - 1;
    const uint64_t idx = 3 * i + 3 * nx * j;
    return Vec3(data_[idx] / 255.0, data_[idx + 1] / 255.0, data_[idx + 2] / 255.0);
}

// ---- src/material.rs ---------------------------------------------------------------------
static Vec3 reflect(const Vec3 &v, const Vec3 &n) { return v - 2.0 * v.dot(n) * n; } // :9-11
static std::optional<Vec3> refract(const Vec3 &v, const Vec3 &n, double ni_over_nt) { // :13-23
    const Vec3 uv = v.normalize();
    const double dt = uv.dot(n);
    const double discriminant = 1.0 - ni_over_nt * ni_over_nt * (1.0 - dt * dt);
    if (discriminant > 0.0) return ni_over_nt * (uv - n * dt) - n * std::sqrt(discriminant);
    return std::nullopt;
}
static double schlick(double cosine, double ref_idx) { // :25-28
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    const double x = 1.0 - cosine, x2 = x * x;
    return r0 + (1.0 - r0) * (x * (x2 * x2));
}
std::optional<std::pair<Ray, Vec3>> Lambertian::scatter(const Ray &ray, const HitRecord &hit) const { // :49-53
    const Vec3 target = hit.p + hit.normal + random_in_unit_sphere();
    return std::make_pair(Ray(hit.p, target - hit.p, ray.time()), albedo_->value(hit.u, hit.v, hit.p));
}
std::optional<std::pair<Ray, Vec3>> Metal::scatter(const Ray &ray, const HitRecord &hit) const { // :75-87
    Vec3 reflected = reflect(ray.direction().normalize(), hit.normal);
    if (fuzz_ > 0.0) reflected = reflected + fuzz_ * random_in_unit_sphere();
    if (reflected.dot(hit.normal) > 0.0)
        return std::make_pair(Ray(hit.p, reflected, ray.time()), albedo_->value(hit.u, hit.v, hit.p));
    return std::nullopt;
}
std::optional<std::pair<Ray, Vec3>> Dielectric::scatter(const Ray &ray, const HitRecord &hit) const { // :106-126
    const Vec3 attenuation(1.0, 1.0, 1.0);
    Vec3 outward_normal;
    double ni_over_nt, cosine;
    if (ray.direction().dot(hit.normal) > 0.0) {
        cosine = ref_idx_ * ray.direction().dot(hit.normal) / ray.direction().magnitude();
        outward_normal = -hit.normal;
        ni_over_nt = ref_idx_;
    } else {
        cosine = -ray.direction().dot(hit.normal) / ray.direction().magnitude();
        outward_normal = hit.normal;
        ni_over_nt = 1.0 / ref_idx_;
    }
    if (auto refracted = refract(ray.direction(), outward_normal, ni_over_nt)) {
        const double reflect_prob = schlick(cosine, ref_idx_);
        if (render_rng().gen() >= reflect_prob) return std::make_pair(Ray(hit.p, *refracted, ray.time()), attenuation);
    }
    return std::make_pair(Ray(hit.p, reflect(ray.direction(), hit.normal), ray.time()), attenuation);
}
std::optional<std::pair<Ray, Vec3>> Isotropic::scatter(const Ray &ray, const HitRecord &hit) const { // :165-168
    return std::make_pair(Ray(hit.p, random_in_unit_sphere(), ray.time()), albedo_->value(hit.u, hit.v, hit.p));
}

// ---- src/color.rs:6-23 -------------------------------------------------------------------
static thread_local bool g_sky_background = false;
void set_sky_background(bool on) { g_sky_background = on; }
bool sky_background() { return g_sky_background; }
static thread_local bool g_face_forward = false; // opt-in RTMI_FLAG_FACE_FORWARD for the CPU evaluation
void set_face_forward(bool on) { g_face_forward = on; }

Vec3 color(const Ray &ray, const Hittable &world, size_t depth) {
    if (auto hit = world.hit(ray, 0.001, kF64Max)) {
        const Vec3 emitted = hit->material->emitted(hit->u, hit->v, hit->p);
        if (depth < 50) {
            // extension: opaque materials see the normal turned against the ray; Dielectric resolves the side itself
            if (g_face_forward && !dynamic_cast<const Dielectric *>(hit->material) && ray.direction().dot(hit->normal) > 0.0)
                hit->normal = -hit->normal;
            if (auto sc = hit->material->scatter(ray, *hit))
                return emitted + sc->second.zip_mul(color(sc->first, world, depth + 1));
        }
        return emitted;
    }
    if (g_sky_background) { // color.rs:18-20
        const Vec3 unit_direction = ray.direction().normalize();
        const double t = 0.5 * (unit_direction.y + 1.0);
        return Vec3(1.0, 1.0, 1.0) * (1.0 - t) + Vec3(0.5, 0.7, 1.0) * t;
    }
    return Vec3(0.0, 0.0, 0.0);
}

// ======================================================================================
// lowering
// ======================================================================================
int Texture::lower(SceneBuilder &) const { throw Unsupported("user-defined Texture cannot be lowered to the device"); }
int Material::lower(SceneBuilder &) const { throw Unsupported("user-defined Material cannot be lowered to the device"); }

int SceneBuilder::texture_index(const Texture *t) {
    auto it = tex_ids_.find(t);
    if (it != tex_ids_.end()) return it->second;
    const int id = t->lower(*this);
    tex_ids_[t] = id;
    return id;
}
int SceneBuilder::material_index(const Material *m) {
    auto it = mat_ids_.find(m);
    if (it != mat_ids_.end()) return it->second;
    const int id = m->lower(*this);
    mat_ids_[m] = id;
    return id;
}
bool SceneBuilder::texture_needs_uv(int tex) const {
    const rtmi_texture &t = out.textures[(size_t)tex];
    if (t.kind == RTMI_TEX_IMAGE) return true;
    if (t.kind == RTMI_TEX_CHECKER) return texture_needs_uv(t.i0) || texture_needs_uv(t.i1);
    return false;
}
int SolidTexture::lower(SceneBuilder &b) const {
    rtmi_texture t{};
    t.kind = RTMI_TEX_SOLID; t.f0 = (float)color_.x; t.f1 = (float)color_.y; t.f2 = (float)color_.z;
    return b.add_texture(t);
}
int CheckerTexture::lower(SceneBuilder &b) const {
    rtmi_texture t{};
    t.kind = RTMI_TEX_CHECKER;
    t.i0 = b.texture_index(odd_.get());
    t.i1 = b.texture_index(even_.get());
    return b.add_texture(t);
}
int NoiseTexture::lower(SceneBuilder &b) const {
    rtmi_perlin pn{};
    for (int i = 0; i < 256; i++) {
        pn.ranvec[4 * i] = (float)noise_.ran_vec_[(size_t)i].x;
        pn.ranvec[4 * i + 1] = (float)noise_.ran_vec_[(size_t)i].y;
        pn.ranvec[4 * i + 2] = (float)noise_.ran_vec_[(size_t)i].z;
        pn.perm[i] = (int32_t)noise_.perm_x_[(size_t)i];
        pn.perm[256 + i] = (int32_t)noise_.perm_y_[(size_t)i];
        pn.perm[512 + i] = (int32_t)noise_.perm_z_[(size_t)i];
    }
    b.out.perlin.push_back(pn);
    rtmi_texture t{};
    t.kind = RTMI_TEX_NOISE; t.i0 = (int32_t)b.out.perlin.size() - 1; t.f0 = (float)scale_;
    return b.add_texture(t);
}
int ImageTexture::lower(SceneBuilder &b) const {
    if ((size_t)nx_ * ny_ * 3 != data_.size() || nx_ == 0 || ny_ == 0) throw Panic("ImageTexture: data size != 3*nx*ny");
    rtmi_image im{};
    im.offset = b.out.image_data.size(); im.nx = nx_; im.ny = ny_;
    b.out.image_data.insert(b.out.image_data.end(), data_.begin(), data_.end());
    b.out.images.push_back(im);
    rtmi_texture t{};
    t.kind = RTMI_TEX_IMAGE; t.i0 = (int32_t)b.out.images.size() - 1;
    return b.add_texture(t);
}
static int lower_mat(SceneBuilder &b, int kind, const Texture *tex, double param) {
    rtmi_material m{};
    m.kind = kind; m.param = (float)param;
    m.tex = tex ? b.texture_index(tex) : 0;
    m.flags = (tex && b.texture_needs_uv(m.tex)) ? RTMI_MATFLAG_NEEDS_UV : 0u;
    return b.add_material(m);
}
int Lambertian::lower(SceneBuilder &b) const { return lower_mat(b, RTMI_MAT_LAMBERTIAN, albedo_.get(), 0.0); }
int Metal::lower(SceneBuilder &b) const { return lower_mat(b, RTMI_MAT_METAL, albedo_.get(), fuzz_); }
int Dielectric::lower(SceneBuilder &b) const { return lower_mat(b, RTMI_MAT_DIELECTRIC, nullptr, ref_idx_); }
int DiffuseLight::lower(SceneBuilder &b) const { return lower_mat(b, RTMI_MAT_DIFFUSE_LIGHT, emit_.get(), 0.0); }
int Isotropic::lower(SceneBuilder &b) const { return lower_mat(b, RTMI_MAT_ISOTROPIC, albedo_.get(), 0.0); }

// strips FlipNormals wrappers (negation commutes exactly with translation and rotation)
static const Hittable *strip_flips(const Hittable *h, bool &flip) {
    while (auto f = dynamic_cast<const FlipNormals *>(h)) { flip = !flip; h = f->inner().get(); }
    return h;
}
// Strips the wrappers a device PRIMITIVE may carry — FlipNormals (hittable.rs:67-88), Traslate (traslate.rs:6-9),
// Rotate (rotate.rs:21-28), in any order and number — and returns the innermost object.  `chain` (optional) receives the
// transforms outermost first, exactly as lower_item records an item's chain: Traslate<H> / Rotate<H> are generic over
// any Hittable, so the reference lets them sit anywhere, e.g. as the children of a BVHNode (bvh.rs:11-12).
static const Hittable *strip_wrappers(const Hittable *h, bool &flip, std::vector<rtmi_xform> *chain) {
    for (;;) {
        if (auto f = dynamic_cast<const FlipNormals *>(h)) { flip = !flip; h = f->inner().get(); continue; }
        if (auto t = dynamic_cast<const Traslate *>(h)) {
            if (chain) {
                rtmi_xform x{};
                x.kind = RTMI_XF_TRANSLATE; x.x = (float)t->offset_.x; x.y = (float)t->offset_.y; x.z = (float)t->offset_.z;
                chain->push_back(x);
            }
            h = t->hitable_.get();
            continue;
        }
        if (auto r = dynamic_cast<const Rotate *>(h)) {
            if (chain) {
                rtmi_xform x{};
                x.kind = RTMI_XF_ROTATE_X + (int)r->axis_; x.x = (float)r->sin_theta_; x.y = (float)r->cos_theta_;
                chain->push_back(x);
            }
            h = r->hittable_.get();
            continue;
        }
        return h;
    }
}
// A primitive that can never report a hit, whatever the ray: a Rect with x0 > x1 or y0 > y1 (rect.rs:51 rejects every
// x) — final_scene's light is one (tests/test.rs:444-452).  Its test has no side effect (no random draw), so the
// members of a list scan that can never be hit are left out of the flat scene: the scan returns what it returned.
static bool never_hit(const Hittable *h) {
    bool dummy = false;
    h = strip_wrappers(h, dummy, nullptr);
    if (auto r = dynamic_cast<const Rect *>(h)) return r->x0_ > r->x1_ || r->y0_ > r->y1_;
    return false;
}
static bool contains_moving(const Hittable *h) {
    bool dummy = false;
    h = strip_wrappers(h, dummy, nullptr);
    if (dynamic_cast<const MovingSphere *>(h)) return true;
    if (auto n = dynamic_cast<const BVHNode *>(h)) return contains_moving(n->left_.get()) || contains_moving(n->right_.get());
    if (auto l = dynamic_cast<const HittableList *>(h)) { // a list that sits as a BVH child (lower_list_leaf)
        for (const auto &m : l->items())
            if (contains_moving(m.get())) return true;
    }
    return false;
}

// one primitive -> planes A/B + meta; returns RTMI_LEAF-style type in the high bits of nothing: plain index
int SceneBuilder::push_prim(const Hittable &h0, bool flip, bool force_moving) {
    float A[4] = {0, 0, 0, 0}, B[4] = {0, 0, 0, 0};
    rtmi_prim_meta m{};
    // an instanced primitive: its own Traslate / Rotate chain (outermost first) goes to `xforms`, referenced from the
    // meta word; FlipNormals anywhere in the chain only toggles the flag (negation commutes with both)
    std::vector<rtmi_xform> chain;
    const Hittable &h = *strip_wrappers(&h0, flip, &chain);
    m.flags = flip ? RTMI_PRIMFLAG_FLIP : 0u;
    if (!chain.empty()) {
        if (chain.size() > RTMI_PRIM_XF_MAX) throw Unsupported("more than 15 Traslate/Rotate wrappers around one primitive");
        if (out.xforms.size() + chain.size() >= (1u << 20)) throw Unsupported("too many instance transforms");
        m.flags |= ((uint32_t)chain.size() << RTMI_PRIMFLAG_XF_COUNT_SHIFT) | ((uint32_t)out.xforms.size() << RTMI_PRIMFLAG_XF_FIRST_SHIFT);
        out.xforms.insert(out.xforms.end(), chain.begin(), chain.end());
    }
    m.inv_dt = 0.0f;
    if (auto s = dynamic_cast<const Sphere *>(&h)) {
        A[0] = (float)s->center_.x; A[1] = (float)s->center_.y; A[2] = (float)s->center_.z; A[3] = (float)s->radius_;
        m.material = material_index(s->material_.get());
        if (force_moving) { // c0 + (time - 0)*1 * 0 == c0 exactly: same bits, one code path in the BVH
            m.type = RTMI_PRIM_MSPHERE; m.inv_dt = 1.0f;
        } else {
            m.type = RTMI_PRIM_SPHERE;
        }
    } else if (auto ms = dynamic_cast<const MovingSphere *>(&h)) {
        const float c0[3] = {(float)ms->center0_.x, (float)ms->center0_.y, (float)ms->center0_.z};
        const float c1[3] = {(float)ms->center1_.x, (float)ms->center1_.y, (float)ms->center1_.z};
        const float t0 = (float)ms->time0_, t1 = (float)ms->time1_;
        A[0] = c0[0]; A[1] = c0[1]; A[2] = c0[2]; A[3] = (float)ms->radius_;
        B[0] = c1[0] - c0[0]; B[1] = c1[1] - c0[1]; B[2] = c1[2] - c0[2]; B[3] = t0;
        m.inv_dt = 1.0f / (t1 - t0);
        m.type = RTMI_PRIM_MSPHERE;
        m.material = material_index(ms->material_.get());
    } else if (auto r = dynamic_cast<const Rect *>(&h)) {
        A[0] = (float)r->x0_; A[1] = (float)r->y0_; A[2] = (float)r->x1_; A[3] = (float)r->y1_;
        B[0] = (float)r->k_;
        m.flags |= ((uint32_t)r->plane_) << RTMI_PRIMFLAG_PLANE_SHIFT;
        m.type = RTMI_PRIM_RECT;
        m.material = material_index(r->material_.get());
    } else if (auto c = dynamic_cast<const Cube *>(&h)) {
        A[0] = (float)c->p_min_.x; A[1] = (float)c->p_min_.y; A[2] = (float)c->p_min_.z; A[3] = (float)c->p_max_.x;
        B[0] = (float)c->p_max_.y; B[1] = (float)c->p_max_.z;
        m.type = RTMI_PRIM_CUBE;
        m.material = material_index(c->material_.get());
    } else {
        throw Unsupported("this Hittable cannot be a device primitive (supported: Sphere, MovingSphere, Rect, Cube)");
    }
    out.prim_a.insert(out.prim_a.end(), A, A + 4);
    out.prim_b.insert(out.prim_b.end(), B, B + 4);
    out.prim_meta.push_back(m);
    const float big = 3.40282346638528859811704183484516925e+38f;
    const float gate[8] = {-big, -big, -big, 0.0f, big, big, big, 0.0f}; // no gate unless a BVH sets one
    out.prim_gate.insert(out.prim_gate.end(), gate, gate + 8);
    out.prim_box.push_back(AABB(Vec3(0, 0, 0), Vec3(0, 0, 0)));
    out.prim_has_box.push_back(0);
    return (int)out.prim_meta.size() - 1;
}

// f64 box -> fp32, one rounding; coordinates beyond the fp32 range (Rotate::bounding_box is the whole f64 space,
// rotate.rs:36-37 — its update never fires) saturate at +-FLT_MAX: such a slab passes every ray in either precision
static float sat_f32(double v) {
    const double big = 3.40282346638528859811704183484516925e+38;
    return v > big ? (float)big : (v < -big ? (float)-big : (float)v);
}
static void put_box(float mn[3], float mx[3], const AABB &b) {
    mn[0] = sat_f32(b.min.x); mn[1] = sat_f32(b.min.y); mn[2] = sat_f32(b.min.z);
    mx[0] = sat_f32(b.max.x); mx[1] = sat_f32(b.max.y); mx[2] = sat_f32(b.max.z);
}

// BVHNode -> rtmi_bvh_node records in preorder; leaves appended left to right (so the primitive
// index is the in-order rank the fast-cull tie rule needs).  A node over one object
// (bvh.rs:44-45: left and right are the same Rc) stores that leaf once, referenced twice.
// Pruned ("fast-cull") traversal assumes that whatever a subtree can report lies inside the subtree's box.
// The reference's boxes do not guarantee it: Rect::bounding_box (rect.rs:71-75) is right for the XY plane
// only, and a MovingSphere leaves its box outside the time range it was built for.  true_bounds() is the
// geometry's real extent (a MovingSphere over its own [time0, time1]); contained() checks every node of a
// BVH against it.  A BVH that fails is lowered with pruning disabled (item.scale = 1e30, unbounded leaf
// boxes): the fast kernels then visit exactly what BVHNode::hit visits.
static bool true_bounds_inner(const Hittable *h, AABB &out);
// object -> world through a chain (outermost first), in f64: innermost wrapper first (traslate.rs:21-22, rotate.rs:94-105)
static Vec3 chain_to_world(const std::vector<rtmi_xform> &chain, Vec3 p) {
    for (size_t k = chain.size(); k-- > 0;) {
        const rtmi_xform &X = chain[k];
        if (X.kind == RTMI_XF_TRANSLATE) { p = p + Vec3(X.x, X.y, X.z); continue; }
        const int r = X.kind - RTMI_XF_ROTATE_X, a = (r + 1) % 3, b = (r + 2) % 3;
        const double s = X.x, c = X.y, pa = p[a], pb = p[b];
        p[a] = c * pa - s * pb;
        p[b] = s * pa + c * pb;
    }
    return p;
}
static bool true_bounds(const Hittable *h, AABB &out) {
    bool dummy = false;
    std::vector<rtmi_xform> chain;
    const Hittable *inner = strip_wrappers(h, dummy, &chain);
    if (auto l = dynamic_cast<const HittableList *>(inner)) { // a list as a BVH child: the union of what its members can report
        if (!chain.empty()) return false;
        bool any = false;
        for (const auto &m : l->items()) {
            AABB mb(Vec3(0, 0, 0), Vec3(0, 0, 0));
            if (!true_bounds(m.get(), mb)) continue;
            out = any ? surrounding_box(out, mb) : mb;
            any = true;
        }
        return any;
    }
    AABB ib(Vec3(0, 0, 0), Vec3(0, 0, 0));
    if (!true_bounds_inner(inner, ib)) return false;
    if (chain.empty()) { out = ib; return true; }
    // the box of the eight transformed corners (with the fp32 sines / cosines the device uses), widened by a relative
    // 1e-6 for the fp32 rounding of the device's transforms: far inside the absolute padding every consumer adds
    bool first = true;
    for (int i = 0; i < 8; i++) {
        const Vec3 c((i & 1) ? ib.max.x : ib.min.x, (i & 2) ? ib.max.y : ib.min.y, (i & 4) ? ib.max.z : ib.min.z);
        const Vec3 w = chain_to_world(chain, c);
        if (first) { out = AABB(w, w); first = false; }
        else out = surrounding_box(out, AABB(w, w));
    }
    for (int k = 0; k < 3; k++) {
        const double e = 1e-6 * std::fmax(std::fabs(out.min[k]), std::fabs(out.max[k]));
        out.min[k] -= e; out.max[k] += e;
    }
    return true;
}
static bool true_bounds_inner(const Hittable *h, AABB &out) {
    if (auto r = dynamic_cast<const Rect *>(h)) {
        int k, a, b;
        plane_axes((int)r->plane_, k, a, b);
        if (r->x0_ > r->x1_ || r->y0_ > r->y1_) return false; // never hit (rect.rs:51): no extent
        out.min[k] = r->k_; out.max[k] = r->k_;
        out.min[a] = r->x0_; out.max[a] = r->x1_;
        out.min[b] = r->y0_; out.max[b] = r->y1_;
        return true;
    }
    if (auto m = dynamic_cast<const MovingSphere *>(h)) {
        const Vec3 rr(std::fabs(m->radius_), std::fabs(m->radius_), std::fabs(m->radius_));
        const AABB b0(m->center0_ - rr, m->center0_ + rr), b1(m->center1_ - rr, m->center1_ + rr);
        out = surrounding_box(b0, b1);
        return true;
    }
    if (auto s = dynamic_cast<const Sphere *>(h)) {
        const Vec3 rr(std::fabs(s->radius_), std::fabs(s->radius_), std::fabs(s->radius_));
        out = AABB(s->center_ - rr, s->center_ + rr);
        return true;
    }
    if (auto b = h->bounding_box(0.0, 1.0)) { out = *b; return true; } // Cube: exact
    return false;
}
static bool contained(const BVHNode &n, double tol, AABB &out, bool &any) {
    bool ok = true;
    any = false;
    const Hittable *ch[2] = {n.left_.get(), n.right_.get()};
    for (int c = 0; c < 2; c++) {
        bool dummy = false, have = false;
        AABB tb(Vec3(0, 0, 0), Vec3(0, 0, 0));
        const Hittable *h = strip_flips(ch[c], dummy);
        if (auto sub = dynamic_cast<const BVHNode *>(h)) ok = contained(*sub, tol, tb, have) && ok;
        else have = true_bounds(h, tb);
        if (!have) continue;
        out = any ? surrounding_box(out, tb) : tb;
        any = true;
    }
    if (any)
        for (int k = 0; k < 3; k++)
            if (out.min[k] < n.bbox_.min[k] - tol || out.max[k] > n.bbox_.max[k] + tol) ok = false;
    return ok;
}
static void moving_time_range(const Hittable *h, float &lo, float &hi) {
    bool dummy = false;
    h = strip_wrappers(h, dummy, nullptr);
    if (auto m = dynamic_cast<const MovingSphere *>(h)) {
        lo = std::fmax(lo, (float)std::fmin(m->time0_, m->time1_));
        hi = std::fmin(hi, (float)std::fmax(m->time0_, m->time1_));
    } else if (auto n = dynamic_cast<const BVHNode *>(h)) {
        moving_time_range(n->left_.get(), lo, hi);
        moving_time_range(n->right_.get(), lo, hi);
    } else if (auto l = dynamic_cast<const HittableList *>(h)) {
        for (const auto &m : l->items()) moving_time_range(m.get(), lo, hi);
    }
}

// ---- ConstantMedium as a child of a BVHNode (r04; bvh.rs:11-12 takes any Hittable, medium.rs:11-15) ----------------
// BVHNode::hit hands BOTH children the query's own (t_min, t_max) and keeps the closer hit (bvh.rs:75-81).  A medium child
// therefore (i) is evaluated with the t_max the BVH was entered with — not shrunk by what its siblings report —, (ii) draws
// its random number whenever its clamped boundary interval is not empty, hit or no hit behind it (medium.rs:30-40), in
// the in-order position of the traversal, (iii) is reached iff every ancestor's box passes, i.e. (nested boxes, monotone
// slab test) iff its PARENT's box passes.  The device keeps media out of its trees: such a child is lowered as a DEFERRED
// medium item right after the BVH item (in-order), evaluated against the saved t_max (RTMI_ITEMFLAG_SAVE_T0 on the BVH
// item), gated by the parent's box (prim_gate of its boundary), accepted if closer than what the tree found.  A node of
// one element (bvh.rs:44-45: left and right are the same Rc) evaluates — and draws — twice: two deferred items.
// (An exact tie between a medium's random distance and a primitive's t is decided for the primitive: probability zero.)
static bool is_medium_child(const Hittable *h) {
    bool dummy = false;
    return dynamic_cast<const ConstantMedium *>(strip_wrappers(h, dummy, nullptr)) != nullptr;
}
// A BVHNode inside Traslate / Rotate as a child of a BVHNode — an instanced subtree (traslate.rs:6-9, rotate.rs:21-28 wrap
// any Hittable).  Like a medium child it receives the query's own (t_min, t_max) and is reached iff its parent's box passes;
// it is lowered as a DEFERRED BVH item behind the enclosing BVH item (rtmi.h): transforms = the enclosing item's, then its
// own; gate = the parent's box in the two xform records behind its chain; query with t_max = T0.  (FlipNormals alone
// around an inner BVHNode stays in the tree: the flip goes down to the primitives.)
static bool is_instanced_bvh_child(const Hittable *h) {
    bool dummy = false;
    std::vector<rtmi_xform> chain;
    const Hittable *core = strip_wrappers(h, dummy, &chain);
    return dynamic_cast<const BVHNode *>(core) != nullptr && !chain.empty();
}
// A HittableList with media among its members (nested lists and FlipNormals looked through) as a child of a BVHNode: its
// scan hands every member the closest hit of the members before it, starting from the t_max the BVH was entered with, so
// the whole list leaves the tree — a group of DEFERRED member items and a terminator behind the BVH item (rtmi.h, LISTSCAN).
static bool list_holds_media(const HittableList &l) {
    for (const auto &m : l.items()) {
        bool d = false;
        const Hittable *s = strip_flips(m.get(), d);
        if (auto sub = dynamic_cast<const HittableList *>(s)) {
            if (list_holds_media(*sub)) return true;
        } else if (is_medium_child(s)) {
            return true;
        }
    }
    return false;
}
static bool is_media_list_child(const Hittable *h) {
    bool d = false;
    auto l = dynamic_cast<const HittableList *>(strip_flips(h, d));
    return l != nullptr && list_holds_media(*l);
}
static bool is_deferred_child(const Hittable *h) { return is_medium_child(h) || is_instanced_bvh_child(h) || is_media_list_child(h); }
static bool has_prims(const Hittable *h) {
    bool dummy = false;
    const Hittable *s = strip_flips(h, dummy);
    if (auto n = dynamic_cast<const BVHNode *>(s)) return has_prims(n->left_.get()) || has_prims(n->right_.get());
    return !is_deferred_child(h);
}
// the media below `h` (a child of `parent`) in traversal order; only called for subtrees without primitives
void SceneBuilder::collect_media(const Hittable *h, const BVHNode &parent, bool flip_all) {
    bool flip = false;
    const Hittable *s = strip_flips(h, flip);
    if (auto n = dynamic_cast<const BVHNode *>(s)) { // (FlipNormals around an inner BVHNode: the flip goes down, as in lower_bvh)
        collect_media(n->left_.get(), *n, flip_all != flip);
        collect_media(n->right_.get(), *n, flip_all != flip); // the same object on both sides: evaluated, and drawn, twice
        return;
    }
    pending_media_.push_back({h, parent.bbox_, (int32_t)out.prim_meta.size(), flip_all});
}

int32_t SceneBuilder::lower_bvh(const BVHNode &n, uint32_t depth, bool force_moving, double pad, bool unbounded_leaves, bool flip_all) {
    const Hittable *ch[2] = {n.left_.get(), n.right_.get()};
    const bool hp[2] = {has_prims(ch[0]), has_prims(ch[1])};
    if (!hp[0] && !hp[1]) { // nothing but media below: no node; they become deferred items
        collect_media(ch[0], n, flip_all);
        collect_media(ch[1], n, flip_all);
        return RTMI_NO_SUBTREE;
    }
    if (depth > out.max_bvh_depth) out.max_bvh_depth = depth;
    const int32_t id = (int32_t)out.nodes.size();
    out.nodes.push_back(rtmi_bvh_node{});
    int32_t child[2] = {0, 0};
    const size_t pend_begin = pending_media_.size();
    for (int c = 0; c < 2; c++) {
        if (c == 1 && ch[1] == ch[0]) { // same object twice
            child[1] = child[0];
            rtmi_bvh_node &me = out.nodes[(size_t)id];
            for (int k = 0; k < 3; k++) { me.rmin[k] = me.lmin[k]; me.rmax[k] = me.lmax[k]; }
            // the reference evaluates the object on both sides (bvh.rs:73-74): for its primitives that is the same answer
            // twice, but every medium below it is evaluated — and draws — a second time, after all of the first visit's
            const size_t pend_end = pending_media_.size();
            for (size_t q = pend_begin; q < pend_end; q++) pending_media_.push_back(pending_media_[q]);
            break;
        }
        if (!hp[c]) { // media only on this side: the slot repeats the sibling (right == left is legal; visited once)
            collect_media(ch[c], n, flip_all);
            child[c] = RTMI_NO_SUBTREE;
            continue;
        }
        bool flip = false;
        const Hittable *h = strip_flips(ch[c], flip);
        if (auto sub = dynamic_cast<const BVHNode *>(h)) {
            // FlipNormals around an inner BVHNode (hittable.rs:67-88 delegates and negates the normal of whatever is
            // hit below): the flip goes down to every primitive of the subtree
            child[c] = lower_bvh(*sub, depth + 1, force_moving, pad, unbounded_leaves, flip_all != flip);
            rtmi_bvh_node &me = out.nodes[(size_t)id];
            put_box(c == 0 ? me.lmin : me.rmin, c == 0 ? me.lmax : me.rmax, sub->bbox_);
        } else if (auto lst = dynamic_cast<const HittableList *>(h)) {
            // a HittableList as a child (bvh.rs:11-12 takes any Hittable): a subtree of always-passing nodes over its members
            AABB cb(Vec3(0, 0, 0), Vec3(0, 0, 0));
            child[c] = lower_list_leaf(*lst, n, depth + 1, flip != flip_all, force_moving, pad, unbounded_leaves, cb);
            rtmi_bvh_node &me = out.nodes[(size_t)id];
            put_box(c == 0 ? me.lmin : me.rmin, c == 0 ? me.lmax : me.rmax, cb);
        } else {
            AABB lb(Vec3(0, 0, 0), Vec3(0, 0, 0));
            child[c] = lower_leaf(*h, n, flip != flip_all, force_moving, pad, unbounded_leaves, lb);
            rtmi_bvh_node &me = out.nodes[(size_t)id];
            put_box(c == 0 ? me.lmin : me.rmin, c == 0 ? me.lmax : me.rmax, lb);
        }
    }
    {
        rtmi_bvh_node &me = out.nodes[(size_t)id];
        if (child[0] == RTMI_NO_SUBTREE) { child[0] = child[1]; for (int k = 0; k < 3; k++) { me.lmin[k] = me.rmin[k]; me.lmax[k] = me.rmax[k]; } }
        if (child[1] == RTMI_NO_SUBTREE) { child[1] = child[0]; for (int k = 0; k < 3; k++) { me.rmin[k] = me.lmin[k]; me.rmax[k] = me.lmax[k]; } }
        me.left = child[0];
        me.right = child[1];
    }
    return id;
}

// One primitive as a leaf below the reference node `n`: its planes, its gate (the box of `n`), its culling box `lb`.
int32_t SceneBuilder::lower_leaf(const Hittable &hh, const BVHNode &n, bool flip, bool force_moving, double pad, bool unbounded_leaves,
                                 AABB &lb) {
    const Hittable *h = &hh;
    const int prim = push_prim(*h, flip, force_moving);
    const int32_t ref = RTMI_LEAF(out.prim_meta[(size_t)prim].type, prim);
    { // gate = the box of THIS node, the leaf's parent in the reference tree, rounded like every node box
        float gmn[3], gmx[3];
        put_box(gmn, gmx, n.bbox_);
        float *g = &out.prim_gate[(size_t)prim * 8];
        g[0] = gmn[0]; g[1] = gmn[1]; g[2] = gmn[2]; g[4] = gmx[0]; g[5] = gmx[1]; g[6] = gmx[2];
        AABB tbx(Vec3(0, 0, 0), Vec3(0, 0, 0));
        if (true_bounds(h, tbx)) { out.prim_box[(size_t)prim] = tbx; out.prim_has_box[(size_t)prim] = 1; }
    }
    // A leaf child has no box test in the reference (bvh.rs:72-73).  The box stored here is
    // used only by the fast-cull prefilter, so it must contain every hit the primitive's own
    // fp32 test can report, at EVERY ray time: the exact box padded by `pad` for static
    // primitives, unbounded for moving spheres and for Rect (whose bounding_box ignores
    // the plane, rect.rs:72-73).
    const double big = 3.40282346638528859811704183484516925e+38;
    lb = AABB(Vec3(-big, -big, -big), Vec3(big, big, big));
    bool dummy2 = false;
    const Hittable *inner = strip_wrappers(h, dummy2, nullptr);
    if (!unbounded_leaves && !dynamic_cast<const MovingSphere *>(inner) && !dynamic_cast<const Rect *>(inner)) {
        // the primitive's TRUE extent (|radius| for a sphere), not bounding_box(): Sphere::bounding_box of a
        // negative radius (the hollow-glass idiom) is an inverted box that no ray passes, and the reference has
        // no leaf box test at all (bvh.rs:72-73)
        AABB tl(Vec3(0, 0, 0), Vec3(0, 0, 0));
        if (true_bounds(h, tl)) {
            const Vec3 pd(pad, pad, pad);
            lb = AABB(tl.min - pd, tl.max + pd);
        }
    }
    return ref;
}

// A HittableList as the child of a BVHNode (bvh.rs:11-12: children are any Rc<dyn Hittable>; its bounding box is the
// union of its members', hittable.rs:49-64).  The reference scans the members in order with a shrinking t_max
// (hittable.rs:37-47): the result is the member with the smallest t, and on an exact tie the LAST rect-like member among
// the tied ones if there is one (Rect / Cube report at t == t_max, rect.rs:47), else the FIRST sphere (Sphere reports only
// below t_max, sphere.rs:44).  The BVH fold is "smallest t, ties -> the later leaf", so the members are numbered spheres in
// reverse order first, then rect-likes in order, and hang below a balanced subtree of nodes whose boxes pass every ray (the
// reference tests no box below the node that holds the list): same winner, bit for bit, from every kernel.  Their gate is
// the box of the holding node, like any leaf's.  Nested lists are flattened (an inner scan continues the outer one).
static void flatten_list_leaf(const HittableList &l, bool flip, std::vector<std::pair<const Hittable *, bool>> &members) {
    for (const auto &m : l.items()) {
        bool f = flip;
        const Hittable *h = strip_flips(m.get(), f);
        if (auto sub = dynamic_cast<const HittableList *>(h)) { flatten_list_leaf(*sub, f, members); continue; }
        members.push_back({h, f});
    }
}
int32_t SceneBuilder::lower_list_leaf(const HittableList &l, const BVHNode &holder, uint32_t depth, bool flip, bool force_moving,
                                      double pad, bool unbounded_leaves, AABB &box_out) {
    std::vector<std::pair<const Hittable *, bool>> members, ordered;
    flatten_list_leaf(l, flip, members);
    if (members.empty()) throw Panic("BVHNode over an empty HittableList: no bounding box (bvh.rs:30)");
    for (size_t k = members.size(); k-- > 0;) { // spheres, last first
        bool d = false;
        const Hittable *in = strip_wrappers(members[k].first, d, nullptr);
        if (dynamic_cast<const Sphere *>(in) || dynamic_cast<const MovingSphere *>(in)) ordered.push_back(members[k]);
    }
    for (const auto &m : members) { // then the rect-likes, first first
        bool d = false;
        const Hittable *in = strip_wrappers(m.first, d, nullptr);
        if (dynamic_cast<const Sphere *>(in) || dynamic_cast<const MovingSphere *>(in)) continue;
        if (!dynamic_cast<const Rect *>(in) && !dynamic_cast<const Cube *>(in))
            throw Unsupported("a HittableList that is a BVH leaf may hold primitives (also wrapped in Traslate / Rotate / FlipNormals) and lists of them only");
        ordered.push_back(m);
    }
    // leaves first (consecutive primitive numbers in the order above), then the subtree over them
    std::vector<int32_t> refs(ordered.size());
    std::vector<AABB> boxes(ordered.size(), AABB(Vec3(0, 0, 0), Vec3(0, 0, 0)));
    for (size_t k = 0; k < ordered.size(); k++)
        refs[k] = lower_leaf(*ordered[k].first, holder, ordered[k].second, force_moving, pad, unbounded_leaves, boxes[k]);
    const double big = 3.40282346638528859811704183484516925e+38;
    const AABB everything(Vec3(-big, -big, -big), Vec3(big, big, big));
    std::function<int32_t(size_t, size_t, uint32_t, AABB &)> build = [&](size_t lo, size_t hi, uint32_t dp, AABB &bo) -> int32_t {
        if (hi - lo == 1) { bo = boxes[lo]; return refs[lo]; }
        if (dp > out.max_bvh_depth) out.max_bvh_depth = dp;
        const int32_t id = (int32_t)out.nodes.size();
        out.nodes.push_back(rtmi_bvh_node{});
        const size_t mid = lo + (hi - lo) / 2;
        AABB bl(Vec3(0, 0, 0), Vec3(0, 0, 0)), br(Vec3(0, 0, 0), Vec3(0, 0, 0));
        const int32_t cl = build(lo, mid, dp + 1, bl), cr = build(mid, hi, dp + 1, br);
        rtmi_bvh_node &me = out.nodes[(size_t)id];
        put_box(me.lmin, me.lmax, bl); put_box(me.rmin, me.rmax, br);
        me.left = cl; me.right = cr;
        bo = everything; // an internal node of the list: no box test in the reference
        return id;
    };
    return build(0, ordered.size(), depth, box_out);
}

// Alternative tree over the primitives [lo,hi) of `prims` for the pruned kernels: binned SAH on the primitives'
// TRUE extents, child boxes padded by `pad`.  Its boxes only cull; a primitive is accepted iff its own test
// passes AND its gate passes (the box of its parent in the reference tree): BVHNode::hit reaches a leaf iff
// every ancestor's box passes AABB::hit with the query's (t_min, t_max), the slab test is monotone in the box
// and every ancestor's box contains the parent's, so that is equivalent to the parent's box passing.  With
// "minimum t, ties -> larger primitive index" (primitives are numbered in the reference's leaf order) any
// conservative tree therefore returns the reference's result bit for bit.  Returns a child reference and the
// padded box of the subtree.
static double box_area(const AABB &b) {
    const double dx = b.max.x - b.min.x, dy = b.max.y - b.min.y, dz = b.max.z - b.min.z;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
int32_t SceneBuilder::build_alt_tree(std::vector<int> &prims, size_t lo, size_t hi, uint32_t depth, double pad, AABB *box_out) {
    const size_t n = hi - lo;
    const Vec3 pd(pad, pad, pad);
    if (n == 1) {
        const int prim = prims[lo];
        const AABB &b = out.prim_box[(size_t)prim];
        *box_out = AABB(b.min - pd, b.max + pd);
        return RTMI_LEAF(out.prim_meta[(size_t)prim].type, prim);
    }
    auto centroid = [&](int p, int a) { const AABB &b = out.prim_box[(size_t)p]; return 0.5 * (b.min[a] + b.max[a]); };
    size_t mid = lo + n / 2;
    bool split_done = false;
    if (depth < 40) { // SAH split; deeper than that (degenerate inputs) fall back to balanced medians
        double best_cost = 1e300;
        int best_axis = -1, best_bin = -1;
        const int NB = 16;
        double cmin[3] = {1e300, 1e300, 1e300}, cmax[3] = {-1e300, -1e300, -1e300};
        for (size_t i = lo; i < hi; i++)
            for (int a = 0; a < 3; a++) { const double c = centroid(prims[i], a); cmin[a] = std::fmin(cmin[a], c); cmax[a] = std::fmax(cmax[a], c); }
        for (int a = 0; a < 3; a++) {
            if (!(cmax[a] - cmin[a] > 1e-12) || !(cmax[a] - cmin[a] < 1e30)) continue;
            AABB bb[NB];
            size_t cnt[NB] = {0};
            bool used[NB] = {false};
            const double scale = NB / (cmax[a] - cmin[a]);
            for (size_t i = lo; i < hi; i++) {
                int b = (int)((centroid(prims[i], a) - cmin[a]) * scale);
                b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                bb[b] = used[b] ? surrounding_box(bb[b], out.prim_box[(size_t)prims[i]]) : out.prim_box[(size_t)prims[i]];
                used[b] = true; cnt[b]++;
            }
            AABB lbox[NB], rbox[NB], run;
            size_t lc[NB], rc[NB], acc = 0;
            bool has = false;
            for (int b = 0; b < NB; b++) { // prefix boxes / counts from the left
                if (used[b]) { run = has ? surrounding_box(run, bb[b]) : bb[b]; has = true; }
                lbox[b] = run; acc += cnt[b]; lc[b] = acc;
            }
            acc = 0; has = false;
            for (int b = NB - 1; b >= 0; b--) { // suffix boxes / counts from the right
                if (used[b]) { run = has ? surrounding_box(run, bb[b]) : bb[b]; has = true; }
                rbox[b] = run; acc += cnt[b]; rc[b] = acc;
            }
            for (int b = 0; b < NB - 1; b++) {
                if (lc[b] == 0 || rc[b + 1] == 0) continue;
                const double cost = box_area(lbox[b]) * (double)lc[b] + box_area(rbox[b + 1]) * (double)rc[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        if (best_axis >= 0) {
            const double scale = 16 / (cmax[best_axis] - cmin[best_axis]);
            auto it = std::stable_partition(prims.begin() + (long)lo, prims.begin() + (long)hi, [&](int p) {
                int b = (int)((centroid(p, best_axis) - cmin[best_axis]) * scale);
                b = b < 0 ? 0 : (b >= 16 ? 15 : b);
                return b <= best_bin;
            });
            mid = (size_t)(it - prims.begin());
            split_done = mid > lo && mid < hi;
        }
    }
    if (!split_done) { // balanced median split along the widest centroid axis
        int axis = 0; double ext = -1.0;
        for (int a = 0; a < 3; a++) {
            double mn = 1e300, mx = -1e300;
            for (size_t i = lo; i < hi; i++) { const double c = centroid(prims[i], a); mn = std::fmin(mn, c); mx = std::fmax(mx, c); }
            if (mx - mn > ext && mx - mn < 1e30) { ext = mx - mn; axis = a; }
        }
        mid = lo + n / 2;
        std::stable_sort(prims.begin() + (long)lo, prims.begin() + (long)hi, [&](int a, int b) { return centroid(a, axis) < centroid(b, axis); });
    }
    const int32_t id = (int32_t)alt_scratch_.size();
    alt_scratch_.push_back(rtmi_bvh_node{});
    AABB lb, rb;
    const int32_t l = build_alt_tree(prims, lo, mid, depth + 1, pad, &lb);
    const int32_t r = build_alt_tree(prims, mid, hi, depth + 1, pad, &rb);
    rtmi_bvh_node &me = alt_scratch_[(size_t)id];
    put_box(me.lmin, me.lmax, lb);
    put_box(me.rmin, me.rmax, rb);
    me.left = l; me.right = r;
    *box_out = surrounding_box(lb, rb);
    return id;
}

// Optimal collapse of the binary SAH tree into 4-wide nodes (the dynamic programme of Ylitie, Karras, Laine 2017, for
// width 4) under a HEIGHT bound: every 4-wide node costs one four-slot visit for every ray that enters its box, so the
// cost of a collapse is the summed box area of its nodes (the leaves cost the same in every collapse); and the cooperative
// traversal is bound by its longest chains, so among the collapses the lowest tree whose cost is within 2 % of the
// unconstrained optimum is taken.  For a binary node v and a height budget h (a leaf slot has height 0)
//   node[v][h]      = area(v) + min_{a = 1..3} (forest[left][a][h-1] + forest[right][4 - a][h-1])   v becomes a node
//   forest[v][j][h] = min(node[v][h], min_{a = 1..j-1} (forest[left][a][h] + forest[right][j - a][h]))  <= j slots
// with forest[leaf][j][h] = 0, forest[v][1][h] = node[v][h], node[v][0] = infinity.  The fixed "grandchildren" rule it
// replaces left 37 % of the nodes with two children (3.08 per node; now 3.5).
void SceneBuilder::plan_collapse(int32_t root) {
    alt_plan_height_ = 0;
    if (root < 0) return;
    const double inf = 1e300;
    // bounded = with the height dimension (heights 0..H); otherwise one level that stands for "any height" — for trees too
    // large for the table (17 x 5 doubles per binary node) or too unbalanced for H
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool bounded = attempt == 0 && alt_scratch_.size() <= 65536;
        if (attempt == 0 && !bounded) continue;
        const int H = bounded ? RTMI_ALT_PLAN_H : 0;
        alt_plan_levels_ = H + 1;
        alt_forest_.assign(alt_scratch_.size() * (size_t)alt_plan_levels_ * 5, inf);
        alt_split_.assign(alt_scratch_.size() * (size_t)alt_plan_levels_ * 5, 0);
        auto at = [&](int32_t v, int h, int j) { return ((size_t)v * (size_t)alt_plan_levels_ + (size_t)h) * 5 + (size_t)j; };
        std::function<void(int32_t)> go = [&](int32_t v) {
            const rtmi_bvh_node &n = alt_scratch_[(size_t)v];
            if (n.left >= 0) go(n.left);
            if (n.right >= 0) go(n.right);
            auto F = [&](int32_t c, int j, int h) { return c < 0 ? 0.0 : alt_forest_[at(c, h, j)]; };
            double mn[3], mx[3];
            for (int k = 0; k < 3; k++) { mn[k] = std::fmin((double)n.lmin[k], (double)n.rmin[k]); mx[k] = std::fmax((double)n.lmax[k], (double)n.rmax[k]); }
            const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
            double area = 2.0 * (dx * dy + dy * dz + dz * dx);
            if (!(area < 1e200)) area = 1e200; // unbounded leaf boxes (Rect, MovingSphere: +-FLT_MAX): finite, and no inf * 0
            for (int h = 0; h <= H; h++) {
                double best = inf;
                int best_a = 1;
                if (!bounded || h > 0)
                    for (int a = 1; a <= 3; a++) {
                        const double c = F(n.left, a, bounded ? h - 1 : 0) + F(n.right, 4 - a, bounded ? h - 1 : 0);
                        if (c < best) { best = c; best_a = a; }
                    }
                const double node = best < inf ? area + best : inf;
                alt_split_[at(v, h, 0)] = (int8_t)best_a; // slot 0 of the table: how this node, as a node, shares its four slots
                alt_forest_[at(v, h, 1)] = node;
                for (int j = 2; j <= 4; j++) {
                    double f = node;
                    int sp = 0;
                    for (int a = 1; a < j; a++) {
                        const double c = F(n.left, a, h) + F(n.right, j - a, h);
                        if (c < f) { f = c; sp = a; }
                    }
                    alt_forest_[at(v, h, j)] = f;
                    alt_split_[at(v, h, j)] = (int8_t)sp;
                }
            }
        };
        go(root);
        const double optimum = alt_forest_[at(root, H, 1)];
        if (bounded && !(optimum < inf)) continue; // does not fit H levels: plan without the bound
        alt_plan_height_ = H;
        // (final_scene: the 400 cubes reach the optimum at height 5, the lowest possible; the 1000 spheres cost 1.002 x
        // the optimum at height 6, their lowest, and reach it at 7)
        if (bounded)
            for (int h = 1; h <= H; h++)
                if (alt_forest_[at(root, h, 1)] <= optimum * 1.02) { alt_plan_height_ = h; break; }
        return;
    }
}

// Binary SAH tree (alt_scratch_) -> 4-wide nodes along the plan of plan_collapse().  Returns a child reference for the
// 4-wide tree.
int32_t SceneBuilder::collapse_alt(int32_t ref, uint32_t depth, int height) {
    if (ref < 0) return ref; // leaf
    if (depth > out.alt_max_depth) out.alt_max_depth = depth;
    const int32_t id = (int32_t)out.alt_nodes.size();
    out.alt_nodes.push_back(rtmi_bvh4_node{});
    struct Slot { int32_t ref; float mn[3], mx[3]; };
    std::vector<Slot> slots;
    const rtmi_bvh_node b = alt_scratch_[(size_t)ref];
    auto add = [&](int32_t r, const float *mn, const float *mx) {
        Slot sl; sl.ref = r;
        for (int k = 0; k < 3; k++) { sl.mn[k] = mn[k]; sl.mx[k] = mx[k]; }
        slots.push_back(sl);
    };
    // Which descendants of `ref` become the (up to four) children of this node is decided by plan_collapse(): the
    // choice that minimises the summed box area of all 4-wide nodes of the subtree (each node costs one four-slot visit
    // for every ray that enters its box) — see there.
    const bool bounded = alt_plan_levels_ > 1;
    const int hb = bounded ? height - 1 : 0; // height budget of the slots of this node
    auto split_of = [&](int32_t v, int h, int j) { return (int)alt_split_[((size_t)v * (size_t)alt_plan_levels_ + (size_t)h) * 5 + (size_t)j]; };
    std::function<void(int32_t, const float *, const float *, int)> emit = [&](int32_t r, const float *mn, const float *mx, int j) {
        if (r < 0 || j == 1) { add(r, mn, mx); return; }
        const int a = split_of(r, hb, j);
        if (a == 0) { add(r, mn, mx); return; } // cheaper as a node of its own
        const rtmi_bvh_node g = alt_scratch_[(size_t)r];
        emit(g.left, g.lmin, g.lmax, a);
        emit(g.right, g.rmin, g.rmax, j - a);
    };
    {
        const int a = split_of(ref, bounded ? height : 0, 0);
        emit(b.left, b.lmin, b.lmax, a);
        emit(b.right, b.rmin, b.rmax, 4 - a);
    }
    rtmi_bvh4_node me{};
    const float big = 3.40282346638528859811704183484516925e+38f;
    for (int c = 0; c < 4; c++) {
        if (c < (int)slots.size()) {
            me.minx[c] = slots[c].mn[0]; me.miny[c] = slots[c].mn[1]; me.minz[c] = slots[c].mn[2];
            me.maxx[c] = slots[c].mx[0]; me.maxy[c] = slots[c].mx[1]; me.maxz[c] = slots[c].mx[2];
            me.child[c] = collapse_alt(slots[c].ref, depth + 1, height - 1);
        } else { // empty slot: a box no ray can hit
            me.minx[c] = me.miny[c] = me.minz[c] = big;
            me.maxx[c] = me.maxy[c] = me.maxz[c] = -big;
            me.child[c] = RTMI_NO_CHILD;
        }
    }
    out.alt_nodes[(size_t)id] = me;
    return id;
}

// The members of a list with media that was a child of a BVHNode, in scan order, then the terminator (rtmi.h, LISTSCAN).
void SceneBuilder::lower_scan_group(const Hittable &top, const DeferredMedium &deferred) {
    bool flip = deferred.flip;
    const auto *list = dynamic_cast<const HittableList *>(strip_flips(&top, flip));
    std::vector<std::pair<const Hittable *, bool>> all, members;
    flatten_list_leaf(*list, flip, all);
    for (const auto &m : all) {
        if (!is_medium_child(m.first) && never_hit(m.first)) continue; // no hit, no draw: left out of the scan
        bool d = false;
        if (dynamic_cast<const BVHNode *>(strip_wrappers(m.first, d, nullptr)))
            throw Unsupported("a BVHNode as a member of a HittableList that holds media and is a BVH child is not lowered");
        members.push_back(m);
    }
    for (size_t k = 0; k < members.size(); k++) {
        DeferredMedium dm = deferred;
        dm.flip = members[k].second;
        dm.save_t0 = deferred.save_t0 && k == 0;
        dm.scan = RTMI_ITEMFLAG_LISTSCAN_MEMBER | (k == 0 ? RTMI_ITEMFLAG_LISTSCAN_BEGIN : 0u);
        lower_item(*members[k].first, &dm);
    }
    rtmi_item end{};
    end.kind = RTMI_ITEM_LIST;
    end.first = deferred.rank; // leaves of the enclosing tree that precede the list in traversal order (ties)
    end.alt_first = -1;
    end.flags = RTMI_ITEMFLAG_DEFERRED | RTMI_ITEMFLAG_LISTSCAN_END;
    out.items.push_back(end);
    run_item_ = -1;
}

void SceneBuilder::lower_item(const Hittable &top, const DeferredMedium *deferred) {
    if (deferred && deferred->scan == 0u && is_media_list_child(&top)) { lower_scan_group(top, *deferred); return; }
    rtmi_item it{};
    it.alt_first = -1;
    it.xform_first = (int32_t)out.xforms.size();
    bool flip = false, medium = false, nested = false;
    float inner_neg_inv_density = 0.0f;
    uint32_t medium_outer = 0;
    const Hittable *h = &top;
    if (deferred) { // a child of a BVHNode lowered as an item: it sits inside the transforms of that BVH item — a copy of them first —
        flip = deferred->flip; // ... and inside the FlipNormals around that item or around its ancestors within the tree
        for (int k = 0; k < deferred->chain_count; k++) out.xforms.push_back(out.xforms[(size_t)(deferred->chain_first + k)]);
        it.xform_count = deferred->chain_count;
    }
    for (;;) { // peel wrappers, outermost first
        if (auto f = dynamic_cast<const FlipNormals *>(h)) { flip = !flip; h = f->inner().get(); continue; }
        if (auto m = dynamic_cast<const ConstantMedium *>(h)) {
            if (medium) { // a medium as the boundary of a medium (medium.rs:11-15 is generic): one level, no wrappers in between
                if (nested) throw Unsupported("ConstantMedium nested more than once is not lowered");
                if ((uint32_t)it.xform_count != medium_outer) throw Unsupported("Traslate / Rotate between a ConstantMedium and the ConstantMedium that is its boundary is not lowered");
                nested = true;
                inner_neg_inv_density = -(1.0f / (float)m->density_); // (its phase function never shows: the hit record is the outer medium's)
                h = m->boundary_.get();
                continue;
            }
            if (it.xform_count > 15) throw Unsupported("ConstantMedium inside more than 15 Traslate/Rotate wrappers");
            medium_outer = (uint32_t)it.xform_count; // the wrappers peeled so far hold the medium itself, not its boundary
            medium = true;
            it.medium_material = material_index(m->phase_function_.get());
            it.neg_inv_density = -(1.0f / (float)m->density_);
            h = m->boundary_.get();
            continue;
        }
        if (auto t = dynamic_cast<const Traslate *>(h)) {
            rtmi_xform x{};
            x.kind = RTMI_XF_TRANSLATE; x.x = (float)t->offset_.x; x.y = (float)t->offset_.y; x.z = (float)t->offset_.z;
            out.xforms.push_back(x); it.xform_count++;
            h = t->hitable_.get();
            continue;
        }
        if (auto r = dynamic_cast<const Rotate *>(h)) {
            rtmi_xform x{};
            x.kind = RTMI_XF_ROTATE_X + (int)r->axis_; x.x = (float)r->sin_theta_; x.y = (float)r->cos_theta_;
            out.xforms.push_back(x); it.xform_count++;
            h = r->hittable_.get();
            continue;
        }
        break;
    }
    it.flags = (flip ? RTMI_ITEMFLAG_FLIP : 0u) | (medium ? RTMI_ITEMFLAG_MEDIUM : 0u) | (medium_outer << RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) |
               (nested ? RTMI_ITEMFLAG_NESTED_MEDIUM : 0u);
    const auto push_inner_medium = [&]() { // the record behind the chain (behind the gate records of a DEFERRED BVH item)
        if (!nested) return;
        rtmi_xform r{};
        r.kind = RTMI_XF_INNER_MEDIUM; r.x = inner_neg_inv_density;
        out.xforms.push_back(r);
    };
    if (!deferred) push_inner_medium();
    if (deferred) {
        const bool is_bvh = dynamic_cast<const BVHNode *>(h) != nullptr;
        if (!medium && !is_bvh && deferred->scan == 0u) throw Panic("lower_item: a deferred item must be a ConstantMedium, an instanced BVHNode or a member of a list scan");
        it.flags |= deferred->scan;
        if (deferred->chain_count > 15 || it.xform_count > 15) throw Unsupported("a deferred child of a BVHNode inside more than 15 Traslate/Rotate wrappers");
        it.flags |= RTMI_ITEMFLAG_DEFERRED | ((uint32_t)deferred->chain_count << RTMI_ITEMFLAG_GATE_OUTER_SHIFT) |
                    (deferred->save_t0 ? RTMI_ITEMFLAG_SAVE_T0 : 0u);
        if (is_bvh) { // geometry = a BVHNode (an instanced subtree, or a medium's boundary): the gate travels in two records
                      // behind the chain (rtmi.h), before the primitives' own chains; its primitives keep their own gates
            float gmn[3], gmx[3];
            put_box(gmn, gmx, deferred->gate);
            rtmi_xform g0{}, g1{};
            g0.kind = RTMI_XF_GATE_MIN; g0.x = gmn[0]; g0.y = gmn[1]; g0.z = gmn[2];
            g1.kind = RTMI_XF_GATE_MAX; g1.x = gmx[0]; g1.y = gmx[1]; g1.z = gmx[2];
            out.xforms.push_back(g0); out.xforms.push_back(g1);
        }
        push_inner_medium();
    }
    if (auto bvh = dynamic_cast<const BVHNode *>(h)) {
        if (!has_prims(bvh)) { // nothing but media below: no BVH item at all, only the deferred ones
            if (medium) throw Unsupported("a ConstantMedium over a BVHNode of media is not lowered");
            pending_media_.clear();
            collect_media(bvh->left_.get(), *bvh, false);
            collect_media(bvh->right_.get(), *bvh, false);
            const std::vector<PendingMedium> pend = std::move(pending_media_);
            pending_media_.clear();
            run_item_ = -1;
            for (size_t k = 0; k < pend.size(); k++) {
                const DeferredMedium dm{pend[k].gate, it.xform_first, it.xform_count, k == 0 && !deferred, pend[k].rank, pend[k].flip != flip};
                lower_item(*pend[k].obj, &dm);
            }
            return;
        }
        pending_media_.clear();
        it.kind = RTMI_ITEM_BVH;
        put_box(it.root_min, it.root_max, bvh->bbox_);
        double scale = 0.0;
        for (int k = 0; k < 3; k++) scale = std::fmax(scale, std::fmax(std::fabs(bvh->bbox_.min[k]), std::fabs(bvh->bbox_.max[k])));
        AABB tb(Vec3(0, 0, 0), Vec3(0, 0, 0));
        bool any = false;
        if (!(scale < 1e30)) {
            // a Rotate somewhere below makes every ancestor's box the whole space (rotate.rs:36-37): the margins of the
            // pruned traversal then scale with the geometry's TRUE extent, not with 1.8e308
            (void)contained(*bvh, 1e300, tb, any);
            scale = 0.0;
            if (any)
                for (int k = 0; k < 3; k++) scale = std::fmax(scale, std::fmax(std::fabs(tb.min[k]), std::fabs(tb.max[k])));
            if (!any || !(scale < 1e30)) scale = 1e30;
            any = false;
        }
        const bool prunable = contained(*bvh, scale / 65536.0, tb, any);
        it.scale = prunable ? (float)scale : 1e30f; // 1e30: the pruning margin swallows every distance
        const size_t prim_begin = out.prim_meta.size();
        it.first = lower_bvh(*bvh, 1, contains_moving(bvh), scale / 8192.0, !prunable, false);
        moving_time_range(bvh, out.bvh_time_lo, out.bvh_time_hi);
        if (prunable) { // alternative (SAH) tree over the same primitives, traversed by the cooperative kernel
            std::vector<int> prims;
            for (size_t q = prim_begin; q < out.prim_meta.size(); q++)
                if (out.prim_has_box[q]) prims.push_back((int)q); // a primitive without extent is never hit
            if (prims.size() >= 2) {
                AABB rootbox(Vec3(0, 0, 0), Vec3(0, 0, 0));
                alt_scratch_.clear();
                const int32_t broot = build_alt_tree(prims, 0, prims.size(), 1, scale / 8192.0, &rootbox);
                plan_collapse(broot);
                it.alt_first = collapse_alt(broot, 1, alt_plan_height_);
            }
        }
    } else if (auto list = dynamic_cast<const HittableList *>(h)) {
        it.kind = RTMI_ITEM_LIST;
        it.first = (int32_t)out.prim_meta.size();
        for (const auto &e : list->items()) {
            bool f2 = false;
            const Hittable *p = strip_flips(e.get(), f2);
            if (never_hit(p)) continue; // left out of the scan
            push_prim(*p, f2, false);
            it.count++;
        }
    } else {
        // A run of consecutive plain primitives of the world list (no transform, no medium) becomes ONE list
        // item: the device scans its primitives in order with the shrinking t_max exactly as it scans items
        // (hittable.rs:37-47), without the per-item overhead.  FlipNormals goes to the primitive's flag.
        if (!medium && it.xform_count == 0 && !deferred) {
            if (never_hit(h)) return; // left out of the scan; the run goes on
            if (run_item_ >= 0) {
                rtmi_item &run = out.items[(size_t)run_item_];
                if (run.first + run.count == (int32_t)out.prim_meta.size()) {
                    push_prim(*h, flip, false);
                    run.count++;
                    return;
                }
            }
            it.kind = RTMI_ITEM_LIST;
            it.flags = 0u;
            it.first = push_prim(*h, flip, false);
            it.count = 1;
            out.items.push_back(it);
            run_item_ = (int)out.items.size() - 1;
            return;
        }
        it.kind = RTMI_ITEM_LIST;
        it.first = push_prim(*h, false, false);
        it.count = 1;
    }
    run_item_ = -1;
    if (deferred && !medium && it.kind == RTMI_ITEM_BVH) it.count = deferred->rank; // leaves of the enclosing tree that precede it in traversal order (ties)
    if (deferred && it.kind == RTMI_ITEM_LIST) { // the gate: the box of the BVHNode the medium (the list) was a child of, with every primitive (rtmi.h)
        if (it.count < 1) throw Unsupported("a member of a list scan without a primitive that can be hit");
        float gmn[3], gmx[3];
        put_box(gmn, gmx, deferred->gate);
        for (int32_t q = it.first; q < it.first + it.count; q++) {
            float *g = &out.prim_gate[(size_t)q * 8];
            g[0] = gmn[0]; g[1] = gmn[1]; g[2] = gmn[2]; g[4] = gmx[0]; g[5] = gmx[1]; g[6] = gmx[2];
        }
    }
    if (it.kind == RTMI_ITEM_BVH && !pending_media_.empty()) { // media that were children of this BVH: deferred items, in order
        if (medium) throw Unsupported("a ConstantMedium whose boundary BVHNode holds media or instanced subtrees is not lowered");
        if (!deferred) it.flags |= RTMI_ITEMFLAG_SAVE_T0; // (a deferred BVH item's own deferred children share its group's T0)
        out.items.push_back(it);
        const std::vector<PendingMedium> pend = std::move(pending_media_);
        pending_media_.clear();
        for (const PendingMedium &pm : pend) {
            const DeferredMedium dm{pm.gate, it.xform_first, it.xform_count, false, pm.rank, pm.flip != flip};
            lower_item(*pm.obj, &dm);
        }
        return;
    }
    out.items.push_back(it);
}

void SceneBuilder::lower_world(const Hittable &world) {
    // world.hit(ray, 0.001, MAX) on a HittableList == the scan the device performs over items;
    // any other world is a list of one
    if (auto list = dynamic_cast<const HittableList *>(&world)) {
        for (const auto &e : list->items()) lower_item(*e);
        if (list->items().empty()) throw Unsupported("empty world");
    } else {
        lower_item(world);
    }
    if (out.max_bvh_depth > RTMI_MAX_BVH_DEPTH) throw Unsupported("BVH deeper than RTMI_MAX_BVH_DEPTH");
}

rtmi_scene_desc LoweredScene::desc() const {
    rtmi_scene_desc d{};
    d.abi_version = RTMI_ABI_VERSION;
    d.n_items = (uint32_t)items.size(); d.items = items.data();
    d.n_prims = (uint32_t)prim_meta.size(); d.prim_a = prim_a.data(); d.prim_b = prim_b.data(); d.prim_meta = prim_meta.data();
    d.n_nodes = (uint32_t)nodes.size(); d.nodes = nodes.data();
    d.n_xforms = (uint32_t)xforms.size(); d.xforms = xforms.data();
    d.n_materials = (uint32_t)materials.size(); d.materials = materials.data();
    d.n_textures = (uint32_t)textures.size(); d.textures = textures.data();
    d.n_perlin = (uint32_t)perlin.size(); d.perlin = perlin.data();
    d.n_images = (uint32_t)images.size(); d.images = images.data();
    d.image_data = image_data.data(); d.image_bytes = image_data.size();
    d.max_bvh_depth = max_bvh_depth;
    d.prim_gate = prim_gate.data();
    d.alt_max_depth = alt_max_depth;
    d.n_alt_nodes = (uint32_t)alt_nodes.size(); d.alt_nodes = alt_nodes.data();
    d.bvh_time_lo = bvh_time_lo; d.bvh_time_hi = bvh_time_hi;
    return d;
}
LoweredScene lower_scene(const Hittable &world) {
    SceneBuilder b;
    b.lower_world(world);
    return std::move(b.out);
}

// ======================================================================================
// Camera — src/camera.rs:21-67, plus render()/create_image on the device
// ======================================================================================
Camera::Camera(const Vec3 &look_from, const Vec3 &look_at, const Vec3 &view_up, double vertical_fov, double aspect,
               double aperture, double focus_dist, double time0, double time1) {
    const double theta = vertical_fov * kPi / 180.0;
    const double half_height = focus_dist * std::tan(theta / 2.0);
    const double half_width = aspect * half_height;
    const Vec3 w = (look_from - look_at).normalize();
    const Vec3 u = view_up.cross(w).normalize();
    const Vec3 v = w.cross(u);
    origin_ = look_from;
    lower_left_corner_ = look_from - half_width * u - half_height * v - focus_dist * w;
    horizontal_ = 2.0 * half_width * u;
    vertical_ = 2.0 * half_height * v;
    u_ = u; v_ = v;
    time0_ = time0; time1_ = time1;
    lens_radius_ = aperture / 2.0;
}
Ray Camera::get_ray(double s, double t) const {
    Vec3 origin = origin_;
    if (lens_radius_ != 0.0) {
        const Vec3 rd = lens_radius_ * random_in_unit_disk();
        origin = origin_ + (u_ * rd.x + v_ * rd.y);
    }
    const double time = time0_ + render_rng().gen() * (time1_ - time0_);
    return Ray(origin, lower_left_corner_ + s * horizontal_ + t * vertical_ - origin, time);
}
rtmi_camera Camera::lower() const {
    rtmi_camera c{};
    const Vec3 *src[6] = {&origin_, &lower_left_corner_, &horizontal_, &vertical_, &u_, &v_};
    float *dst[6] = {c.origin, c.lower_left_corner, c.horizontal, c.vertical, c.u, c.v};
    for (int i = 0; i < 6; i++) { dst[i][0] = (float)src[i]->x; dst[i][1] = (float)src[i]->y; dst[i][2] = (float)src[i]->z; }
    c.time0 = (float)time0_; c.time1 = (float)time1_; c.lens_radius = (float)lens_radius_;
    return c;
}

static int progress_trampoline(uint64_t done, uint64_t total, void *user) {
    const auto *fn = static_cast<const std::function<bool(uint64_t, uint64_t)> *>(user);
    try { return (*fn)(done, total) ? 0 : 1; } catch (...) { return 1; } // nothing may unwind through the C ABI
}
static rtmi_render_params make_params(uint32_t nx, uint32_t ny, uint32_t ns, const RenderOptions &opt) {
    rtmi_render_params p{};
    p.nx = nx; p.ny = ny; p.ns = ns; p.max_depth = opt.max_depth; p.t_min = (float)opt.t_min; p.flags = opt.flags;
    p.seed = opt.seed; p.tile_rank = 0; p.tile_world = 1; p.spp_chunks = opt.spp_chunks;
    if (opt.progress) {
        p.progress_fn = (uint64_t)(uintptr_t)&progress_trampoline;
        p.progress_user = (uint64_t)(uintptr_t)&opt.progress;
    }
    return p;
}
DeviceScene::DeviceScene(const Hittable &world, const std::vector<int> &devices) : devices_(devices) {
    const LoweredScene ls = lower_scene(world);
    const rtmi_scene_desc d = ls.desc(); // borrowed for the call: rtmi_multi_create copies it to every device
    if (int rc = rtmi_multi_create(&d, devices_.data(), (uint32_t)devices_.size(), &handle_))
        throw std::runtime_error(std::string("rtmi_multi_create: ") + rtmi_last_error() + " (code " + std::to_string(rc) + ")");
}
DeviceScene::~DeviceScene() { rtmi_multi_destroy(handle_); }
void DeviceScene::prepare(uint32_t nx, uint32_t ny, uint32_t ns, const RenderOptions &opt) {
    const rtmi_render_params p = make_params(nx, ny, ns, opt);
    if (rtmi_multi_prepare(handle_, &p)) throw std::runtime_error(std::string("rtmi_multi_prepare: ") + rtmi_last_error());
}
Image DeviceScene::render(const Camera &cam, uint32_t nx, uint32_t ny, uint32_t ns, const RenderOptions &opt) {
    const rtmi_render_params p = make_params(nx, ny, ns, opt);
    const rtmi_camera c = cam.lower();
    Image img;
    img.nx = nx; img.ny = ny;
    img.linear.resize((size_t)nx * ny * 3);
    img.rgb8.resize((size_t)nx * ny * 3);
    if (rtmi_multi_render(handle_, &c, &p, img.linear.data(), img.rgb8.data(), &img.stats))
        throw std::runtime_error(std::string("rtmi_multi_render: ") + rtmi_last_error());
    return img;
}
Image Camera::render(const Hittable &world, uint32_t nx, uint32_t ny, uint32_t ns, const RenderOptions &opt) const {
    if (!opt.devices.empty()) // several GPUs of this process: tiles t % n, one gather
        return DeviceScene(world, opt.devices).render(*this, nx, ny, ns, opt);
    const LoweredScene ls = lower_scene(world);
    const rtmi_scene_desc d = ls.desc();
    const rtmi_render_params p = make_params(nx, ny, ns, opt);
    const rtmi_camera c = lower();
    Image img;
    img.nx = nx; img.ny = ny;
    img.linear.resize((size_t)nx * ny * 3);
    img.rgb8.resize((size_t)nx * ny * 3);
    rtmi_scene *scene = nullptr;
    if (int rc = rtmi_scene_create(&d, opt.device, &scene))
        throw std::runtime_error(std::string("rtmi_scene_create: ") + rtmi_last_error() + " (code " + std::to_string(rc) + ")");
    const int rc = rtmi_render(scene, &c, &p, img.linear.data(), img.rgb8.data(), nullptr, &img.stats);
    const std::string err = rc ? rtmi_last_error() : "";
    rtmi_scene_destroy(scene);
    if (rc) throw std::runtime_error("rtmi_render: " + err);
    return img;
}
std::string Image::to_ppm() const {
    std::string s;
    s.resize(rtmi_ppm_p3(nx, ny, rgb8.data(), nullptr, 0));
    s.resize(rtmi_ppm_p3(nx, ny, rgb8.data(), s.data(), s.size()));
    return s;
}
void Image::write_ppm(const std::string &path, bool binary) const {
    if (rtmi_write_ppm(path.c_str(), nx, ny, rgb8.data(), binary ? 6 : 3)) throw std::runtime_error(std::string("rtmi_write_ppm: ") + rtmi_last_error());
}
std::string create_image(size_t ny, size_t nx, size_t ns, const Camera &cam, const Hittable &world, const RenderOptions &opt) {
    return cam.render(world, (uint32_t)nx, (uint32_t)ny, (uint32_t)ns, opt).to_ppm();
}

} // namespace rt
