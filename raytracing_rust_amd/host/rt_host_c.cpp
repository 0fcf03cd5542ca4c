// rt_host_c.cpp — C bindings of the C++ host mirror (rt_host.hpp) for ctypes/FFI users.
// Handles are opaque pointers owned by this library until rth_free_all().  No exception
// crosses the boundary: constructors return NULL and functions return a non-zero code,
// with the message in rth_last_error().  Where the reference panics the code is RTH_PANIC.
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_host.hpp"

using namespace rt;

#define RTH_API extern "C" __attribute__((visibility("default")))
enum { RTH_OK = 0, RTH_ERROR = 1, RTH_PANIC = 2, RTH_UNSUPPORTED = 3 };

namespace {
thread_local std::string g_err;
thread_local int g_code = 0;

struct Obj {
    enum Kind { TEX, MAT, HIT, LIST, CAM, LOWERED } kind;
    TexturePtr tex;
    MaterialPtr mat;
    HittablePtr hit;
    std::shared_ptr<HittableList> list;
    std::shared_ptr<Camera> cam;
    std::shared_ptr<LoweredScene> lowered;
    rtmi_scene *dev = nullptr;
    rtmi_multi *multi = nullptr; // the lowered scene resident on a device list (rth_upload_multi)
};
std::mutex g_mu;
std::vector<Obj *> g_objs;

Obj *reg(Obj *o) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_objs.push_back(o);
    return o;
}
int set_err(const std::exception &e) {
    g_err = e.what();
    g_code = RTH_ERROR;
    if (dynamic_cast<const Panic *>(&e)) g_code = RTH_PANIC;
    if (dynamic_cast<const Unsupported *>(&e)) g_code = RTH_UNSUPPORTED;
    return g_code;
}
template <typename F>
void *guard_new(F f) {
    try {
        return f();
    } catch (const std::exception &e) {
        set_err(e);
        return nullptr;
    }
}
template <typename F>
int guard(F f) {
    try {
        return f();
    } catch (const std::exception &e) {
        return set_err(e);
    }
}
Obj *O(void *h) { return static_cast<Obj *>(h); }
void *new_tex(TexturePtr t) { Obj *o = new Obj{Obj::TEX}; o->tex = std::move(t); return reg(o); }
void *new_mat(MaterialPtr m) { Obj *o = new Obj{Obj::MAT}; o->mat = std::move(m); return reg(o); }
void *new_hit(HittablePtr h) { Obj *o = new Obj{Obj::HIT}; o->hit = std::move(h); return reg(o); }
HittablePtr H(void *h) {
    Obj *o = O(h);
    if (!o || (o->kind != Obj::HIT && o->kind != Obj::LIST)) throw std::runtime_error("handle is not a Hittable");
    return o->kind == Obj::LIST ? std::static_pointer_cast<const Hittable>(o->list) : o->hit;
}
TexturePtr T(void *h) {
    if (!h || O(h)->kind != Obj::TEX) throw std::runtime_error("handle is not a Texture");
    return O(h)->tex;
}
MaterialPtr M(void *h) {
    if (!h || O(h)->kind != Obj::MAT) throw std::runtime_error("handle is not a Material");
    return O(h)->mat;
}
} // namespace

RTH_API const char *rth_last_error(void) { return g_err.c_str(); }
RTH_API int rth_last_error_code(void) { return g_code; }
RTH_API void rth_free_all(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Obj *o : g_objs) {
        if (o->dev) rtmi_scene_destroy(o->dev);
        if (o->multi) rtmi_multi_destroy(o->multi);
        delete o;
    }
    g_objs.clear();
}
RTH_API void rth_seed_scene_rng(uint64_t seed) { scene_rng().seed(seed, 0, 0, 1); }
RTH_API double rth_scene_uniform(void) { return scene_rng().gen(); }
RTH_API void rth_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }

// ---- textures -------------------------------------------------------------------------
RTH_API void *rth_tex_solid(double r, double g, double b) {
    return guard_new([&] { return new_tex(std::make_shared<SolidTexture>(r, g, b)); });
}
RTH_API void *rth_tex_checker(void *odd, void *even) {
    return guard_new([&] { return new_tex(std::make_shared<CheckerTexture>(T(odd), T(even))); });
}
RTH_API void *rth_tex_noise(double scale) {
    return guard_new([&] { return new_tex(std::make_shared<NoiseTexture>(scale)); });
}
RTH_API void *rth_tex_image(const uint8_t *data, uint32_t nx, uint32_t ny) {
    return guard_new([&] {
        std::vector<uint8_t> v(data, data + (size_t)nx * ny * 3);
        return new_tex(std::make_shared<ImageTexture>(std::move(v), nx, ny));
    });
}
RTH_API int rth_perlin_tables(void *tex, double *ranvec768, int *perm768) {
    return guard([&] {
        auto n = std::dynamic_pointer_cast<const NoiseTexture>(T(tex));
        if (!n) throw std::runtime_error("not a NoiseTexture");
        for (size_t i = 0; i < 256; i++) {
            ranvec768[3 * i] = n->noise_.ran_vec_[i].x; ranvec768[3 * i + 1] = n->noise_.ran_vec_[i].y;
            ranvec768[3 * i + 2] = n->noise_.ran_vec_[i].z;
            perm768[i] = (int)n->noise_.perm_x_[i]; perm768[256 + i] = (int)n->noise_.perm_y_[i];
            perm768[512 + i] = (int)n->noise_.perm_z_[i];
        }
        return RTH_OK;
    });
}
// ---- materials ------------------------------------------------------------------------
RTH_API void *rth_mat_lambertian(void *tex) { return guard_new([&] { return new_mat(std::make_shared<Lambertian>(T(tex))); }); }
RTH_API void *rth_mat_metal(void *tex, double fuzz) { return guard_new([&] { return new_mat(std::make_shared<Metal>(T(tex), fuzz)); }); }
RTH_API void *rth_mat_dielectric(double ri) { return guard_new([&] { return new_mat(std::make_shared<Dielectric>(ri)); }); }
RTH_API void *rth_mat_diffuse_light(void *tex) { return guard_new([&] { return new_mat(std::make_shared<DiffuseLight>(T(tex))); }); }
RTH_API void *rth_mat_isotropic(void *tex) { return guard_new([&] { return new_mat(std::make_shared<Isotropic>(T(tex))); }); }
// ---- hittables ------------------------------------------------------------------------
RTH_API void *rth_sphere(double cx, double cy, double cz, double r, void *mat) {
    return guard_new([&] { return new_hit(std::make_shared<Sphere>(Vec3(cx, cy, cz), r, M(mat))); });
}
RTH_API void *rth_moving_sphere(double ax, double ay, double az, double bx, double by, double bz, double t0, double t1,
                                double r, void *mat) {
    return guard_new([&] {
        return new_hit(std::make_shared<MovingSphere>(Vec3(ax, ay, az), Vec3(bx, by, bz), t0, t1, r, M(mat)));
    });
}
RTH_API void *rth_rect(int plane, double x0, double y0, double x1, double y1, double k, void *mat) {
    return guard_new([&] {
        if (plane < 0 || plane > 2) throw std::runtime_error("bad Plane");
        return new_hit(std::make_shared<Rect>((Plane)plane, x0, y0, x1, y1, k, M(mat)));
    });
}
RTH_API void *rth_cube(double ax, double ay, double az, double bx, double by, double bz, void *mat) {
    return guard_new([&] { return new_hit(std::make_shared<Cube>(Vec3(ax, ay, az), Vec3(bx, by, bz), M(mat))); });
}
RTH_API void *rth_flip_normals(void *h) { return guard_new([&] { return new_hit(std::make_shared<FlipNormals>(H(h))); }); }
RTH_API void *rth_translate(void *h, double ox, double oy, double oz) {
    return guard_new([&] { return new_hit(std::make_shared<Traslate>(H(h), Vec3(ox, oy, oz))); });
}
RTH_API void *rth_rotate(int axis, void *h, double angle) {
    return guard_new([&] {
        if (axis < 0 || axis > 2) throw std::runtime_error("bad Axis");
        return new_hit(std::make_shared<Rotate>((Axis)axis, H(h), angle));
    });
}
RTH_API void *rth_constant_medium(void *boundary, double density, void *tex) {
    return guard_new([&] { return new_hit(std::make_shared<ConstantMedium>(H(boundary), density, T(tex))); });
}
RTH_API void *rth_list_new(void) {
    return guard_new([&] {
        Obj *o = new Obj{Obj::LIST};
        o->list = std::make_shared<HittableList>();
        return (void *)reg(o);
    });
}
RTH_API int rth_list_push(void *list, void *h) {
    return guard([&] {
        if (!list || O(list)->kind != Obj::LIST) throw std::runtime_error("handle is not a HittableList");
        O(list)->list->push(H(h));
        return RTH_OK;
    });
}
RTH_API void *rth_bvh(void **items, int n, double t0, double t1) {
    return guard_new([&] {
        std::vector<HittablePtr> v;
        for (int i = 0; i < n; i++) v.push_back(H(items[i]));
        return new_hit(std::make_shared<BVHNode>(v, t0, t1));
    });
}
RTH_API void *rth_camera(double fx, double fy, double fz, double ax, double ay, double az, double ux, double uy,
                         double uz, double vfov, double aspect, double aperture, double focus_dist, double t0, double t1) {
    return guard_new([&] {
        Obj *o = new Obj{Obj::CAM};
        o->cam = std::make_shared<Camera>(Vec3(fx, fy, fz), Vec3(ax, ay, az), Vec3(ux, uy, uz), vfov, aspect, aperture,
                                          focus_dist, t0, t1);
        return (void *)reg(o);
    });
}
static Camera &CAM(void *h) {
    if (!h || O(h)->kind != Obj::CAM) throw std::runtime_error("handle is not a Camera");
    return *O(h)->cam;
}
RTH_API int rth_camera_lower(void *cam, rtmi_camera *out) { return guard([&] { *out = CAM(cam).lower(); return RTH_OK; }); }
RTH_API int rth_camera_state(void *cam, double *out21) {
    return guard([&] {
        Camera &c = CAM(cam);
        const Vec3 *v[6] = {&c.origin_, &c.lower_left_corner_, &c.horizontal_, &c.vertical_, &c.u_, &c.v_};
        for (int i = 0; i < 6; i++) { out21[3 * i] = v[i]->x; out21[3 * i + 1] = v[i]->y; out21[3 * i + 2] = v[i]->z; }
        out21[18] = c.time0_; out21[19] = c.time1_; out21[20] = c.lens_radius_;
        return RTH_OK;
    });
}

// ---- lowering + device ------------------------------------------------------------------
static Obj *LOW(void *h) {
    if (!h || O(h)->kind != Obj::LOWERED) throw std::runtime_error("handle is not a lowered scene");
    return O(h);
}
RTH_API void *rth_lower(void *world) {
    return guard_new([&] {
        auto ls = std::make_shared<LoweredScene>(lower_scene(*H(world)));
        Obj *o = new Obj{Obj::LOWERED};
        o->lowered = std::move(ls);
        return (void *)reg(o);
    });
}
RTH_API int rth_lowered_desc(void *lowered, rtmi_scene_desc *out) {
    return guard([&] { *out = LOW(lowered)->lowered->desc(); return RTH_OK; });
}
RTH_API int rth_upload(void *lowered, int device) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (o->dev) { rtmi_scene_destroy(o->dev); o->dev = nullptr; }
        const rtmi_scene_desc d = o->lowered->desc();
        if (int rc = rtmi_scene_create(&d, device, &o->dev)) throw std::runtime_error(std::string("rtmi_scene_create: ") + rtmi_last_error() + " (code " + std::to_string(rc) + ")");
        return RTH_OK;
    });
}
RTH_API int rth_render(void *lowered, void *cam, const rtmi_render_params *p, float *out_linear, uint8_t *out_rgb8,
                       uint64_t *out_path_sig, rtmi_stats *stats) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (!o->dev) throw std::runtime_error("scene not uploaded: call rth_upload first");
        const rtmi_camera c = CAM(cam).lower();
        if (rtmi_render(o->dev, &c, p, out_linear, out_rgb8, out_path_sig, stats)) throw std::runtime_error(std::string("rtmi_render: ") + rtmi_last_error());
        return RTH_OK;
    });
}
RTH_API int rth_render_device(void *lowered, void *cam, const rtmi_render_params *p, void *d_texels, void *stream,
                              rtmi_stats *stats) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (!o->dev) throw std::runtime_error("scene not uploaded: call rth_upload first");
        const rtmi_camera c = CAM(cam).lower();
        if (rtmi_render_device(o->dev, &c, p, d_texels, stream, stats)) throw std::runtime_error(std::string("rtmi_render_device: ") + rtmi_last_error());
        return RTH_OK;
    });
}
// overflow report of the asynchronous render calls since the last report (rtmi_scene_status)
RTH_API int rth_scene_status(void *lowered) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (!o->dev) throw std::runtime_error("scene not uploaded: call rth_upload first");
        if (rtmi_scene_status(o->dev, nullptr)) throw std::runtime_error(std::string("rtmi_scene_status: ") + rtmi_last_error());
        return RTH_OK;
    });
}
// whole image on several GPUs of this process (rtmi_render_multi): uploads the lowered scene to every listed device
RTH_API int rth_render_multi(void *lowered, void *cam, const rtmi_render_params *p, const int *devices, uint32_t n,
                             float *out_linear, uint8_t *out_rgb8, rtmi_stats *stats) {
    return guard([&] {
        const rtmi_scene_desc d = LOW(lowered)->lowered->desc();
        const rtmi_camera c = CAM(cam).lower();
        if (rtmi_render_multi(&d, devices, n, &c, p, out_linear, out_rgb8, stats)) throw std::runtime_error(std::string("rtmi_render_multi: ") + rtmi_last_error());
        return RTH_OK;
    });
}
// RTMI_FLAG_PROGRESSIVE: the image of the passes finished so far (only from inside the progress callback of rth_render)
RTH_API int rth_partial_image(void *lowered, const rtmi_render_params *p, float *out_linear, uint8_t *out_rgb8, uint32_t *spp_done) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (!o->dev) throw std::runtime_error("scene not uploaded: call rth_upload first");
        if (rtmi_partial_image(o->dev, p, out_linear, out_rgb8, spp_done)) throw std::runtime_error(std::string("rtmi_partial_image: ") + rtmi_last_error());
        return RTH_OK;
    });
}
// the lowered scene resident on a list of GPUs of this process (rtmi_multi_create); replaces an earlier list
RTH_API int rth_upload_multi(void *lowered, const int *devices, uint32_t n) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (o->multi) { rtmi_multi_destroy(o->multi); o->multi = nullptr; }
        const rtmi_scene_desc d = o->lowered->desc();
        if (int rc = rtmi_multi_create(&d, devices, n, &o->multi)) throw std::runtime_error(std::string("rtmi_multi_create: ") + rtmi_last_error() + " (code " + std::to_string(rc) + ")");
        return RTH_OK;
    });
}
RTH_API int rth_multi_free(void *lowered) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (o->multi) { rtmi_multi_destroy(o->multi); o->multi = nullptr; }
        return RTH_OK;
    });
}
// which exchange the resident handle's renders perform (RTMI_COLLECTIVE_*), -1 without a handle
RTH_API int rth_multi_collective(void *lowered) {
    Obj *o = LOW(lowered);
    return o && o->multi ? rtmi_multi_collective(o->multi) : -1;
}
RTH_API int rth_multi_prepare(void *lowered, const rtmi_render_params *p) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (!o->multi) throw std::runtime_error("scene not resident on a device list: call rth_upload_multi first");
        if (rtmi_multi_prepare(o->multi, p)) throw std::runtime_error(std::string("rtmi_multi_prepare: ") + rtmi_last_error());
        return RTH_OK;
    });
}
RTH_API int rth_multi_render(void *lowered, void *cam, const rtmi_render_params *p, float *out_linear, uint8_t *out_rgb8,
                             rtmi_stats *stats) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (!o->multi) throw std::runtime_error("scene not resident on a device list: call rth_upload_multi first");
        const rtmi_camera c = CAM(cam).lower();
        if (rtmi_multi_render(o->multi, &c, p, out_linear, out_rgb8, stats)) throw std::runtime_error(std::string("rtmi_multi_render: ") + rtmi_last_error());
        return RTH_OK;
    });
}
// allocate the render buffers for `p` ahead of the first render call (optional)
RTH_API int rth_render_prepare(void *lowered, const rtmi_render_params *p) {
    return guard([&] {
        Obj *o = LOW(lowered);
        if (!o->dev) throw std::runtime_error("scene not uploaded: call rth_upload first");
        if (rtmi_render_prepare(o->dev, p)) throw std::runtime_error(std::string("rtmi_render_prepare: ") + rtmi_last_error());
        return RTH_OK;
    });
}
// Camera::render / create_image in one call (lower + upload + render + free)
RTH_API int rth_camera_render(void *cam, void *world, uint32_t nx, uint32_t ny, uint32_t ns, uint64_t seed, uint32_t flags,
                              int device, float *out_linear, uint8_t *out_rgb8, rtmi_stats *stats) {
    return guard([&] {
        RenderOptions opt;
        opt.seed = seed; opt.flags = flags; opt.device = device;
        Image img = CAM(cam).render(*H(world), nx, ny, ns, opt);
        if (out_linear) memcpy(out_linear, img.linear.data(), img.linear.size() * sizeof(float));
        if (out_rgb8) memcpy(out_rgb8, img.rgb8.data(), img.rgb8.size());
        if (stats) *stats = img.stats;
        return RTH_OK;
    });
}

// ---- CPU evaluation of the mirror (f64, the reference's arithmetic) ------------------------
RTH_API int rth_hit(void *h, const double *o, const double *d, double time, double t_min, double t_max, uint64_t seed,
                    double *out9, int *found) {
    return guard([&] {
        render_rng().seed(seed, 0, 0, 0);
        const double mx = 1.79769313486231570814527423731704357e+308;
        auto r = H(h)->hit(Ray(Vec3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2]), time), t_min <= -1.7e308 ? -mx : t_min,
                           t_max >= 1.7e308 ? mx : t_max);
        *found = r ? 1 : 0;
        if (r) {
            out9[0] = r->t; out9[1] = r->u; out9[2] = r->v;
            out9[3] = r->p.x; out9[4] = r->p.y; out9[5] = r->p.z;
            out9[6] = r->normal.x; out9[7] = r->normal.y; out9[8] = r->normal.z;
        }
        return RTH_OK;
    });
}
RTH_API int rth_bounding_box(void *h, double t0, double t1, double *out6, int *found) {
    return guard([&] {
        auto b = H(h)->bounding_box(t0, t1);
        *found = b ? 1 : 0;
        if (b) { out6[0] = b->min.x; out6[1] = b->min.y; out6[2] = b->min.z; out6[3] = b->max.x; out6[4] = b->max.y; out6[5] = b->max.z; }
        return RTH_OK;
    });
}
RTH_API int rth_tex_value(void *tex, double u, double v, const double *p, double *out3) {
    return guard([&] {
        const Vec3 c = T(tex)->value(u, v, Vec3(p[0], p[1], p[2]));
        out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
        return RTH_OK;
    });
}
RTH_API int rth_scatter(void *mat, const double *ro, const double *rd, double time, const double *rec9, uint64_t seed,
                        double *out10, int *scattered) {
    return guard([&] {
        render_rng().seed(seed, 0, 0, 0);
        HitRecord rec;
        rec.t = rec9[0]; rec.u = rec9[1]; rec.v = rec9[2];
        rec.p = Vec3(rec9[3], rec9[4], rec9[5]); rec.normal = Vec3(rec9[6], rec9[7], rec9[8]);
        rec.material = M(mat).get();
        auto s = M(mat)->scatter(Ray(Vec3(ro[0], ro[1], ro[2]), Vec3(rd[0], rd[1], rd[2]), time), rec);
        *scattered = s ? 1 : 0;
        if (s) {
            const Vec3 o = s->first.origin(), d = s->first.direction();
            out10[0] = o.x; out10[1] = o.y; out10[2] = o.z; out10[3] = d.x; out10[4] = d.y; out10[5] = d.z;
            out10[6] = s->first.time(); out10[7] = s->second.x; out10[8] = s->second.y; out10[9] = s->second.z;
        }
        return RTH_OK;
    });
}
RTH_API int rth_emitted(void *mat, double u, double v, const double *p, double *out3) {
    return guard([&] {
        const Vec3 c = M(mat)->emitted(u, v, Vec3(p[0], p[1], p[2]));
        out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
        return RTH_OK;
    });
}
RTH_API int rth_get_ray(void *cam, double s, double t, uint64_t seed, double *out7) {
    return guard([&] {
        render_rng().seed(seed, 0, 0, 0);
        const Ray r = CAM(cam).get_ray(s, t);
        const Vec3 o = r.origin(), d = r.direction();
        out7[0] = o.x; out7[1] = o.y; out7[2] = o.z; out7[3] = d.x; out7[4] = d.y; out7[5] = d.z; out7[6] = r.time();
        return RTH_OK;
    });
}
RTH_API void rth_set_sky_background(int on) { set_sky_background(on != 0); }
RTH_API void rth_set_face_forward(int on) { set_face_forward(on != 0); }
RTH_API void rth_set_uv_book(int on) { set_uv_book(on != 0); }
// color() of one camera sample on the CPU (f64), same stream layout as the device
RTH_API int rth_color_sample(void *cam, void *world, uint32_t nx, uint32_t ny, uint32_t i, uint32_t j, uint32_t s,
                             uint64_t seed, double *out3) {
    return guard([&] {
        Rng &rng = render_rng();
        rng.seed(seed, s, j * nx + i, 0);
        const double u = ((double)i + rng.gen()) / (double)nx;
        const double v = ((double)j + rng.gen()) / (double)ny;
        const Ray ray = CAM(cam).get_ray(u, v);
        const Vec3 c = color(ray, *H(world), 0);
        out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
        return RTH_OK;
    });
}
