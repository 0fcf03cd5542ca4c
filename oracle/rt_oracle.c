/* oracle/rt_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the hot path of DrStiev/raytracing_rust
 *   create_image (tests/test.rs:55-85) -> Camera::get_ray (src/camera.rs:53-67)
 *   -> color (src/color.rs:6-23) -> Hittable::hit (src/{hittable,aabb,bvh,sphere,rect,
 *   cube,traslate,rotate,medium}.rs) -> Material::scatter/emitted (src/material.rs)
 *   -> Texture::value (src/texture.rs, src/perlin.rs), samplers (src/util.rs).
 * Structure follows the reference: an object graph walked by recursive, dynamically
 * dispatched `hit` calls and a recursive `color`.  Each function cites the reference
 * lines it restates.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (raytracing_rust_amd/) never does.
 *
 * Built twice from this one source (oracle/Makefile):
 *   liborc_f64.so  REAL = double : the literal restatement (the reference's arithmetic).
 *   liborc_f32.so  REAL = float  : same algorithm in the device's fp32 arithmetic
 *                                  contract (DESIGN.md "fp32 arithmetic contract"):
 *                                  selected at run time with ORC_ARITH_DEVICE.
 *
 * PINNING.  The reference is Rust and cannot be built here (no cargo/rustc); its RNG is
 * OS-seeded (rand::thread_rng), so its lit images are not reproducible.  The only golden
 * vectors its repo holds for this path are output/final_scene.ppm and
 * output/cornell_smoke.ppm (sha256 a78e19cf...b0eb5b, all-black 800x800 P3); this oracle
 * reproduces both byte-for-byte (tests/test_oracle_golden.py).  Everything that depends
 * on the random stream is "parity unpinned" against the reference and is pinned only
 * against the formulas in the cited source lines via known-answer tests.
 *
 * RNG.  rand::thread_rng()/gen::<f64>() (rand 0.8.5, not vendored) is replaced by a
 * Philox4x32-10 counter stream: counter = (block, sample, pixel, stream_id), key = seed;
 * the n-th draw of a sample is word n%4 of block n/4, mapped to [0,1) with 24 bits.
 * Draw ORDER is the reference's program order (SURVEY.md §8 "Per-sample RNG draw order").
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rtmi_math.h"

#ifdef ORC_F32
typedef float REAL;
#define R_SQRT sqrtf
#define R_FMA fmaf
#define R_FMAX fmaxf
#define R_FMIN fminf
#define R_FLOOR floorf
#define R_FABS fabsf
#define R_MAX 3.40282346638528859811704183484516925e+38f
#define ORC_IS_F32 1
#else
typedef double REAL;
#define R_SQRT sqrt
#define R_FMA fma
#define R_FMAX fmax
#define R_FMIN fmin
#define R_FLOOR floor
#define R_FABS fabs
#define R_MAX 1.79769313486231570814527423731704357e+308
#define ORC_IS_F32 0
#endif

#define ORC_API __attribute__((visibility("default")))

/* ---- run-time switches -------------------------------------------------------- */
enum {
    ORC_ARITH_DEVICE = 1,  /* fp32 arithmetic contract substitutions      */
    ORC_THROUGHPUT_FORM = 2, /* iterative L += T*e form of color() instead of the recursion */
    ORC_SKY = 4,             /* opt-in extension: the gradient background the reference keeps
                              * commented out at color.rs:18-20 (default: black, color.rs:21)  */
    ORC_FACE_FORWARD = 8,    /* opt-in extension: Lambertian / Metal / Isotropic scatter about the normal turned
                              * against the incoming ray (the reference never turns it: sphere.rs:50, rect.rs:58-59);
                              * Dielectric keeps the geometric normal (material.rs:106-114)                      */
    ORC_UV_BOOK = 16         /* opt-in extension: get_sphere_uv adds pi/2 (the book) instead of FRAC_2_PI
                              * (sphere.rs:13)                                                                  */
};
static int g_flags = 0;
#define DEVICE_ARITH (g_flags & ORC_ARITH_DEVICE)

/* ---- instrumentation: operation counts that define the algorithmic work per sample
 *      (SURVEY.md §8(d) cost table) ------------------------------------------------ */
enum {
    C_SAMPLES, C_QUERIES, C_AABB, C_SPHERE, C_MSPHERE, C_RECT, C_XFORM, C_MEDIUM, C_MEDIUM_DRAW,
    C_MAT_FETCH, C_TEX_SOLID, C_TEX_CHECKER, C_TEX_NOISE, C_TEX_IMAGE, C_SC_LAMBERT, C_SC_METAL,
    C_SC_DIELECTRIC, C_SC_ISOTROPIC, C_EMIT, C_DRAWS, C_SPHERE_TRIALS, C_DISK_TRIALS, C_SPHERE_ACCEPT,
    C_RECT_ACCEPT, C_NCOUNTERS
};
static uint64_t g_cnt[C_NCOUNTERS];
#ifdef ORC_NO_COUNT /* the timed CPU-baseline build: no instrumentation on the hot path */
#define COUNT(k) ((void)0)
#else
#define COUNT(k) (g_cnt[k]++)
#endif

/* ---- Philox4x32-10 (Salmon et al., SC'11) ------------------------------------- */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    uint32_t key[2];
    uint32_t ctr[4]; /* ctr[0] = next block index */
    uint32_t buf[4];
    int pos; /* 4 = empty */
} Stream;

static void stream_init(Stream *s, uint64_t seed, uint32_t sample, uint32_t pixel, uint32_t stream_id) {
    s->key[0] = (uint32_t)seed;
    s->key[1] = (uint32_t)(seed >> 32);
    s->ctr[0] = 0;
    s->ctr[1] = sample;
    s->ctr[2] = pixel;
    s->ctr[3] = stream_id;
    s->pos = 4;
}
static uint32_t stream_u32(Stream *s) {
    if (s->pos == 4) {
        philox4x32_10(s->ctr, s->key, s->buf);
        s->ctr[0]++;
        s->pos = 0;
    }
    return s->buf[s->pos++];
}

/* the `rand::thread_rng()` of the render path: one stream per (pixel, sample) */
static Stream g_rng;
/* the `rand::thread_rng()` of scene construction (BVH axes, Perlin tables) */
static Stream g_scene_rng;

/* rng.gen::<f64>() — tests/test.rs:66-67, camera.rs:61, util.rs:8,19, material.rs:118,
 * medium.rs:40.  24-bit uniform (see rtmi_math.h rtmi_u01). */
static REAL rng_uniform(void) {
    COUNT(C_DRAWS);
    return (REAL)rtmi_u01(stream_u32(&g_rng));
}
static double scene_uniform(void) { return (double)rtmi_u01(stream_u32(&g_scene_rng)); }
/* rng.gen_range(0..n) — bvh.rs:40, perlin.rs:7 */
static uint32_t scene_range(uint32_t n) { return (uint32_t)(((uint64_t)stream_u32(&g_scene_rng) * n) >> 32); }

/* ---- Vector3 (nalgebra 0.32 semantics for the ops the path uses) ---------------- */
typedef struct { REAL x, y, z; } V3;
typedef struct { double x, y, z; } D3;
static inline V3 v3(REAL x, REAL y, REAL z) { V3 v = {x, y, z}; return v; }
static inline V3 v_add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 v_sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 v_scale(V3 a, REAL s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 v_div(V3 a, REAL s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline V3 v_neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline V3 v_mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline REAL v_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline REAL v_norm(V3 a) { return R_SQRT(v_dot(a, a)); }
static inline V3 v_normalize(V3 a) { return v_div(a, v_norm(a)); }
static inline REAL v_get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline void v_set(V3 *a, int i, REAL v) { if (i == 0) a->x = v; else if (i == 1) a->y = v; else a->z = v; }
static inline V3 d3_to_v3(D3 d) { return v3((REAL)d.x, (REAL)d.y, (REAL)d.z); }

/* ---- Ray (src/ray.rs:3-29).  inv_d / inv_a are derived values cached per ray frame:
 *      inv_d[a] is exactly the `1.0 / ray.direction()[a]` of aabb.rs:33. ------------ */
typedef struct {
    V3 o, d;
    REAL time;
    V3 inv_d;
    REAL inv_a;
} Ray;
static Ray ray_new(V3 o, V3 d, REAL time) {
    Ray r;
    r.o = o; r.d = d; r.time = time;
    r.inv_d = v3((REAL)1 / d.x, (REAL)1 / d.y, (REAL)1 / d.z);
    r.inv_a = (REAL)1 / v_dot(d, d);
    return r;
}
static inline V3 ray_at(const Ray *r, REAL t) { return v_add(r->o, v_scale(r->d, t)); } /* ray.rs:23-25 */

/* ---- transcendental dispatch --------------------------------------------------- */
static inline REAL m_sin(REAL x) {
#ifdef ORC_F32
    if (DEVICE_ARITH) return rtmi_sinf(x);
    return sinf(x);
#else
    return sin(x);
#endif
}
static inline REAL m_log(REAL x) {
#ifdef ORC_F32
    if (DEVICE_ARITH) return rtmi_logf(x);
    return logf(x);
#else
    return log(x);
#endif
}
static inline REAL m_atan2(REAL y, REAL x) {
#ifdef ORC_F32
    if (DEVICE_ARITH) return rtmi_atan2f(y, x);
    return atan2f(y, x);
#else
    return atan2(y, x);
#endif
}
static inline REAL m_asin(REAL x) {
#ifdef ORC_F32
    if (DEVICE_ARITH) return rtmi_asinf(x);
    return asinf(x);
#else
    return asin(x);
#endif
}

/* ---- object graph --------------------------------------------------------------- */
typedef struct Texture Texture;
typedef struct Material Material;
typedef struct Hittable Hittable;

typedef struct {
    V3 ran_vec[256];
    int perm_x[256], perm_y[256], perm_z[256];
} Perlin;

enum { TEX_SOLID = 0, TEX_CHECKER = 1, TEX_NOISE = 2, TEX_IMAGE = 3 };
struct Texture {
    int kind;
    V3 color;
    const Texture *odd, *even;
    REAL scale;
    Perlin *noise;
    uint8_t *data;
    uint32_t nx, ny;
};

enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_DIFFUSE_LIGHT = 3, MAT_ISOTROPIC = 4 };
struct Material {
    int kind;
    const Texture *tex;
    REAL param; /* fuzz | ref_idx */
};

typedef struct {
    REAL t, u, v;
    V3 p, normal;
    const Material *mat;
} HitRecord; /* hittable.rs:9-16 */

typedef struct { D3 min, max; } AABBd; /* scene set-up stays f64 like the reference */
typedef struct { V3 min, max; } AABB;

enum { H_SPHERE = 0, H_MOVING_SPHERE, H_RECT, H_CUBE, H_LIST, H_FLIP, H_TRANSLATE, H_ROTATE, H_MEDIUM, H_BVH };
struct Hittable {
    int kind;
    /* sphere / moving sphere */
    D3 c0d, c1d; double rd, t0d, t1d;
    V3 c0, c1; REAL radius, t0, t1;
    /* rect */
    int plane; double x0d, y0d, x1d, y1d, kd;
    REAL x0, y0, x1, y1, k;
    /* cube */
    D3 pmind, pmaxd;
    Hittable *sides; /* H_LIST of 6 rects */
    /* list */
    Hittable **items; int n, cap;
    /* wrappers */
    Hittable *child;
    D3 offsetd; V3 offset;
    int axis; double sind, cosd; REAL sin_t, cos_t; int has_bbox; AABBd rot_bbox;
    /* medium */
    REAL density; Material phase;
    /* bvh */
    Hittable *left, *right; AABBd boxd; AABB box;
    const Material *mat;
};

/* ---- allocation arena ------------------------------------------------------------ */
typedef struct Blk { struct Blk *next; char pad[8]; } Blk; /* 16 bytes: payload stays 16-aligned */
static Blk *g_blocks = NULL;
static void *arena_alloc(size_t n) {
    Blk *b = (Blk *)calloc(1, sizeof(Blk) + n);
    if (!b) { fprintf(stderr, "orc: out of memory\n"); abort(); }
    b->next = g_blocks;
    g_blocks = b;
    return (char *)b + sizeof(Blk);
}
ORC_API void orc_free_all(void) {
    while (g_blocks) { Blk *n = g_blocks->next; free(g_blocks); g_blocks = n; }
}

/* ================================================================================== */
/* Textures — src/texture.rs, src/perlin.rs                                           */
/* ================================================================================== */

/* perlin.rs:38-56 */
static REAL perlin_interpolation(V3 c[2][2][2], REAL u, REAL v, REAL w) {
    REAL uu = u * u * ((REAL)3.0 - (REAL)2.0 * u);
    REAL vv = v * v * ((REAL)3.0 - (REAL)2.0 * v);
    REAL ww = w * w * ((REAL)3.0 - (REAL)2.0 * w);
    REAL accum = 0;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                V3 weight = v3(u - (REAL)i, v - (REAL)j, w - (REAL)k);
                accum += ((REAL)i * uu + (REAL)(1 - i) * ((REAL)1.0 - uu)) *
                         ((REAL)j * vv + (REAL)(1 - j) * ((REAL)1.0 - vv)) *
                         ((REAL)k * ww + (REAL)(1 - k) * ((REAL)1.0 - ww)) * v_dot(c[i][j][k], weight);
            }
    return accum;
}

/* Rust `f64 as usize`: saturating, NaN -> 0 (perlin.rs:83-85, texture.rs:91-92) */
static uint64_t as_usize(REAL x) {
    if (!(x > 0)) return 0; /* negative, -0, NaN */
    if (x >= (REAL)18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}

/* perlin.rs:76-97 */
static REAL perlin_noise(const Perlin *pn, V3 p) {
    REAL u = p.x - R_FLOOR(p.x);
    REAL v = p.y - R_FLOOR(p.y);
    REAL w = p.z - R_FLOOR(p.z);
    uint64_t i = as_usize(R_FLOOR(p.x));
    uint64_t j = as_usize(R_FLOOR(p.y));
    uint64_t k = as_usize(R_FLOOR(p.z));
    V3 c[2][2][2];
    for (uint64_t di = 0; di < 2; di++)
        for (uint64_t dj = 0; dj < 2; dj++)
            for (uint64_t dk = 0; dk < 2; dk++)
                c[di][dj][dk] =
                    pn->ran_vec[pn->perm_x[(i + di) & 255] ^ pn->perm_y[(j + dj) & 255] ^ pn->perm_z[(k + dk) & 255]];
    return perlin_interpolation(c, u, v, w);
}

/* perlin.rs:99-109 */
static REAL perlin_turb(const Perlin *pn, V3 p, int depth) {
    REAL accum = 0;
    V3 temp_p = p;
    REAL weight = 1.0;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= (REAL)0.5;
        temp_p = v_scale(temp_p, (REAL)2.0);
    }
    return R_FABS(accum);
}

/* texture.rs:4-6 and the four impls */
static V3 tex_value(const Texture *t, REAL u, REAL v, V3 p) {
    switch (t->kind) {
    case TEX_SOLID: /* texture.rs:21-25 */
        COUNT(C_TEX_SOLID);
        return t->color;
    case TEX_CHECKER: { /* texture.rs:39-48 */
        COUNT(C_TEX_CHECKER);
        REAL s = m_sin((REAL)10.0 * p.x) * m_sin((REAL)10.0 * p.y) * m_sin((REAL)10.0 * p.z);
        if (s < 0) return tex_value(t->odd, u, v, p);
        return tex_value(t->even, u, v, p);
    }
    case TEX_NOISE: { /* texture.rs:65-71 — turb gets the UNSCALED p */
        COUNT(C_TEX_NOISE);
        REAL g = (REAL)0.5 * ((REAL)1.0 + m_sin(t->scale * p.x + (REAL)5.0 * perlin_turb(t->noise, p, 7)));
        /* Vector3::new(1,1,1) * 0.5 * (1 + sin(..)): ((1*0.5) * x) per component == 0.5*x */
        return v3(g, g, g);
    }
    case TEX_IMAGE: { /* texture.rs:86-108 */
        COUNT(C_TEX_IMAGE);
        uint64_t nx = t->nx, ny = t->ny;
        uint64_t i = as_usize(u * (REAL)nx);
        uint64_t j = as_usize(((REAL)1.0 - v) * (REAL)ny);
        if (i > nx - 1) i = nx - 1;
        if (j > ny - 1) j = ny - 1;
        uint64_t idx = 3 * i + 3 * nx * j;
        return v3((REAL)t->data[idx] / (REAL)255.0, (REAL)t->data[idx + 1] / (REAL)255.0,
                  (REAL)t->data[idx + 2] / (REAL)255.0);
    }
    }
    return v3(0, 0, 0);
}

/* ================================================================================== */
/* Samplers — src/util.rs                                                             */
/* ================================================================================== */
/* util.rs:4-13 : draws x, y, z per trial */
static V3 random_in_unit_sphere(void) {
    for (;;) {
        COUNT(C_SPHERE_TRIALS);
        REAL x = rng_uniform(), y = rng_uniform(), z = rng_uniform();
        V3 p = v3((REAL)2.0 * x - (REAL)1.0, (REAL)2.0 * y - (REAL)1.0, (REAL)2.0 * z - (REAL)1.0);
        if (v_dot(p, p) < (REAL)1.0) return p;
    }
}
/* util.rs:15-24 : draws x, y per trial */
static V3 random_in_unit_disk(void) {
    for (;;) {
        COUNT(C_DISK_TRIALS);
        REAL x = rng_uniform(), y = rng_uniform();
        V3 p = v3((REAL)2.0 * x - (REAL)1.0, (REAL)2.0 * y - (REAL)1.0, (REAL)2.0 * (REAL)0.0 - (REAL)0.0);
        if (v_dot(p, p) < (REAL)1.0) return p;
    }
}

/* ================================================================================== */
/* Materials — src/material.rs                                                        */
/* ================================================================================== */
static V3 reflect(V3 v, V3 n) { /* material.rs:9-11 : v - 2.0 * v.dot(n) * n */
    REAL s = (REAL)2.0 * v_dot(v, n);
    return v_sub(v, v_scale(n, s));
}
static int refract(V3 v, V3 n, REAL ni_over_nt, V3 *out) { /* material.rs:13-23 */
    V3 uv = v_normalize(v);
    REAL dt = v_dot(uv, n);
    REAL disc = (REAL)1.0 - ni_over_nt * ni_over_nt * ((REAL)1.0 - dt * dt);
    if (disc > 0) {
        *out = v_sub(v_scale(v_sub(uv, v_scale(n, dt)), ni_over_nt), v_scale(n, R_SQRT(disc)));
        return 1;
    }
    return 0;
}
static REAL schlick(REAL cosine, REAL ref_idx) { /* material.rs:25-28 ; powi(5) = x*(x^2)^2 */
    REAL r0 = ((REAL)1.0 - ref_idx) / ((REAL)1.0 + ref_idx);
    r0 = r0 * r0;
    REAL x = (REAL)1.0 - cosine;
    REAL x2 = x * x;
    REAL x4 = x2 * x2;
    return r0 + ((REAL)1.0 - r0) * (x * x4);
}

/* Material::emitted — material.rs:32 and impls (:55-57, :89-91, :128-130, :148-150, :170-172) */
static V3 mat_emitted(const Material *m, REAL u, REAL v, V3 p) {
    if (m->kind == MAT_DIFFUSE_LIGHT) { COUNT(C_EMIT); return tex_value(m->tex, u, v, p); }
    return v3(0, 0, 0);
}

/* Material::scatter — material.rs:31 and impls */
static int mat_scatter(const Material *m, const Ray *ray, const HitRecord *hit, Ray *scattered, V3 *attenuation) {
    COUNT(C_MAT_FETCH);
    HitRecord turned;
    if ((g_flags & ORC_FACE_FORWARD) && m->kind != MAT_DIELECTRIC && v_dot(ray->d, hit->normal) > (REAL)0.0) {
        turned = *hit;
        turned.normal = v3(-hit->normal.x, -hit->normal.y, -hit->normal.z);
        hit = &turned;
    }
    switch (m->kind) {
    case MAT_LAMBERTIAN: { /* material.rs:49-53 */
        COUNT(C_SC_LAMBERT);
        V3 rs = random_in_unit_sphere();
        V3 dir;
        if (DEVICE_ARITH) {
            dir = v_add(hit->normal, rs); /* contract: (p+n+r)-p evaluated without the p round trip */
        } else {
            V3 target = v_add(v_add(hit->p, hit->normal), rs);
            dir = v_sub(target, hit->p);
        }
        *scattered = ray_new(hit->p, dir, ray->time);
        *attenuation = tex_value(m->tex, hit->u, hit->v, hit->p);
        return 1;
    }
    case MAT_METAL: { /* material.rs:75-87 (fuzz clamp is in the constructor, :70) */
        COUNT(C_SC_METAL);
        V3 reflected = reflect(v_normalize(ray->d), hit->normal);
        if (m->param > 0) reflected = v_add(reflected, v_scale(random_in_unit_sphere(), m->param));
        if (v_dot(reflected, hit->normal) > 0) {
            *scattered = ray_new(hit->p, reflected, ray->time);
            *attenuation = tex_value(m->tex, hit->u, hit->v, hit->p);
            return 1;
        }
        return 0;
    }
    case MAT_DIELECTRIC: { /* material.rs:106-126 */
        COUNT(C_SC_DIELECTRIC);
        *attenuation = v3(1.0, 1.0, 1.0);
        V3 outward_normal;
        REAL ni_over_nt, cosine;
        REAL ddn = v_dot(ray->d, hit->normal);
        if (ddn > 0) {
            cosine = m->param * ddn / v_norm(ray->d);
            outward_normal = v_neg(hit->normal);
            ni_over_nt = m->param;
        } else {
            cosine = -ddn / v_norm(ray->d);
            outward_normal = hit->normal;
            ni_over_nt = (REAL)1.0 / m->param;
        }
        V3 refracted;
        if (refract(ray->d, outward_normal, ni_over_nt, &refracted)) {
            REAL reflect_prob = schlick(cosine, m->param);
            if (rng_uniform() >= reflect_prob) {
                *scattered = ray_new(hit->p, refracted, ray->time);
                return 1;
            }
        }
        *scattered = ray_new(hit->p, reflect(ray->d, hit->normal), ray->time);
        return 1;
    }
    case MAT_DIFFUSE_LIGHT: /* material.rs:144-146 */
        return 0;
    case MAT_ISOTROPIC: { /* material.rs:165-168 */
        COUNT(C_SC_ISOTROPIC);
        V3 rs = random_in_unit_sphere();
        *scattered = ray_new(hit->p, rs, ray->time);
        *attenuation = tex_value(m->tex, hit->u, hit->v, hit->p);
        return 1;
    }
    }
    return 0;
}

/* ================================================================================== */
/* Geometry — hit()                                                                   */
/* ================================================================================== */
static int hit(const Hittable *h, const Ray *r, REAL t_min, REAL t_max, HitRecord *rec);

/* aabb.rs:31-44 */
static int aabb_hit(const AABB *b, const Ray *r, REAL t_min, REAL t_max) {
    COUNT(C_AABB);
    for (int a = 0; a < 3; a++) {
        REAL inv_d = v_get(r->inv_d, a);
        REAL t0 = (v_get(b->min, a) - v_get(r->o, a)) * inv_d;
        REAL t1 = (v_get(b->max, a) - v_get(r->o, a)) * inv_d;
        if (inv_d < 0) { REAL tmp = t0; t0 = t1; t1 = tmp; }
        t_min = R_FMAX(t_min, t0);
        t_max = R_FMIN(t_max, t1);
        if (t_max <= t_min) return 0;
    }
    return 1;
}

/* sphere.rs:9-15 */
static void get_sphere_uv(V3 p, REAL *u, REAL *v) {
    REAL phi = m_atan2(p.z, p.x);
    REAL theta = m_asin(p.y);
#ifdef ORC_F32
    const REAL PI = RTMI_PI_F, FRAC_2_PI = RTMI_2_OVER_PI_F;
#else
    const REAL PI = 3.14159265358979323846264338327950288, FRAC_2_PI = 0.636619772367581343075535053490057448;
#endif
    *u = (REAL)1.0 - (phi + PI) / ((REAL)2.0 * PI);
#ifdef ORC_F32
    const REAL FRAC_PI_2 = RTMI_PIO2_F;
#else
    const REAL FRAC_PI_2 = 1.57079632679489661923132169163975144;
#endif
    *v = (theta + ((g_flags & ORC_UV_BOOK) ? FRAC_PI_2 : FRAC_2_PI)) / PI; /* sic: FRAC_2_PI, not FRAC_PI_2 */
}

/* sphere.rs:37-77 and :122-164 (identical bodies, centre differs) */
static int sphere_hit_at(V3 center, REAL radius, const Material *mat, const Ray *r, REAL t_min, REAL t_max,
                         HitRecord *rec) {
    V3 oc = v_sub(r->o, center);
    REAL a = v_dot(r->d, r->d);
    REAL b = v_dot(oc, r->d);
    REAL c = v_dot(oc, oc) - radius * radius;
    REAL disc = b * b - a * c;
    if (DEVICE_ARITH) {
        /* contract substitution 5: the same discriminant as a (r^2 - |oc - (b/a) d|^2) — the squared
         * distance of the centre from the ray instead of a difference of two numbers of size |oc|^2 a, which in fp32
         * puts the hit point of a small far sphere up to 5e-3 off its surface (the scattered ray then meets the same
         * sphere again beyond t_min: final_scene's small spheres 10 % too dark against the f64 literal).  Explicit
         * fma (one rounding each) exactly as csrc/rtmi_geom.hpp sphere_disc(). */
        REAL q = b * r->inv_a;
        REAL lx = R_FMA(-q, r->d.x, oc.x), ly = R_FMA(-q, r->d.y, oc.y), lz = R_FMA(-q, r->d.z, oc.z);
        REAL l2 = R_FMA(lz, lz, R_FMA(ly, ly, lx * lx));
        disc = a * (radius * radius - l2);
    }
    if (disc > 0) {
        REAL sq = R_SQRT(disc);
        REAL t = DEVICE_ARITH ? (-b - sq) * r->inv_a : (-b - sq) / a;
        if (t < t_max && t > t_min) goto accept;
        t = DEVICE_ARITH ? (-b + sq) * r->inv_a : (-b + sq) / a;
        if (t < t_max && t > t_min) goto accept;
        return 0;
    accept:
        COUNT(C_SPHERE_ACCEPT);
        rec->t = t;
        rec->p = ray_at(r, t);
        rec->normal = v_div(v_sub(rec->p, center), radius); /* outward; never face-forwarded */
        get_sphere_uv(rec->normal, &rec->u, &rec->v);
        rec->mat = mat;
        return 1;
    }
    return 0;
}

/* sphere.rs:115-118 */
static V3 moving_center(const Hittable *h, REAL time) {
    REAL f = DEVICE_ARITH ? (time - h->t0) * ((REAL)1.0 / (h->t1 - h->t0)) : (time - h->t0) / (h->t1 - h->t0);
    return v_add(h->c0, v_scale(v_sub(h->c1, h->c0), f));
}
static D3 moving_center_d(const Hittable *h, double time) {
    double f = (time - h->t0d) / (h->t1d - h->t0d);
    D3 c = {h->c0d.x + f * (h->c1d.x - h->c0d.x), h->c0d.y + f * (h->c1d.y - h->c0d.y),
            h->c0d.z + f * (h->c1d.z - h->c0d.z)};
    return c;
}

static void plane_axes(int plane, int *k, int *a, int *b) { /* rect.rs:40-44 */
    switch (plane) {
    case 0: *k = 0; *a = 1; *b = 2; break; /* YZ */
    case 1: *k = 1; *a = 2; *b = 0; break; /* ZX */
    default: *k = 2; *a = 0; *b = 1; break; /* XY */
    }
}

/* rect.rs:39-69 */
static int rect_hit(const Hittable *h, const Ray *r, REAL t_min, REAL t_max, HitRecord *rec) {
    COUNT(C_RECT);
    int ka, aa, ba;
    plane_axes(h->plane, &ka, &aa, &ba);
    REAL t = DEVICE_ARITH ? (h->k - v_get(r->o, ka)) * v_get(r->inv_d, ka) : (h->k - v_get(r->o, ka)) / v_get(r->d, ka);
    if (t < t_min || t > t_max) return 0;
    REAL x = v_get(r->o, aa) + t * v_get(r->d, aa);
    REAL y = v_get(r->o, ba) + t * v_get(r->d, ba);
    if (x < h->x0 || x > h->x1 || y < h->y0 || y > h->y1) return 0;
    COUNT(C_RECT_ACCEPT);
    rec->u = (x - h->x0) / (h->x1 - h->x0);
    rec->v = (y - h->y0) / (h->y1 - h->y0);
    rec->t = t;
    rec->p = ray_at(r, t);
    rec->normal = v3(0, 0, 0);
    v_set(&rec->normal, ka, 1.0);
    rec->mat = h->mat;
    return 1;
}

/* hittable.rs:37-47 */
static int list_hit(const Hittable *h, const Ray *r, REAL t_min, REAL t_max, HitRecord *rec) {
    REAL closest = t_max;
    int hit_anything = 0;
    HitRecord tmp;
    for (int i = 0; i < h->n; i++) {
        if (hit(h->items[i], r, t_min, closest, &tmp)) {
            closest = tmp.t;
            *rec = tmp;
            hit_anything = 1;
        }
    }
    return hit_anything;
}

static int hit(const Hittable *h, const Ray *r, REAL t_min, REAL t_max, HitRecord *rec) {
    switch (h->kind) {
    case H_SPHERE: /* sphere.rs:37-77 */
        COUNT(C_SPHERE);
        return sphere_hit_at(h->c0, h->radius, h->mat, r, t_min, t_max, rec);
    case H_MOVING_SPHERE: /* sphere.rs:122-164 */
        COUNT(C_MSPHERE);
        return sphere_hit_at(moving_center(h, r->time), h->radius, h->mat, r, t_min, t_max, rec);
    case H_RECT:
        return rect_hit(h, r, t_min, t_max, rec);
    case H_CUBE: /* cube.rs:84-86 */
        return list_hit(h->sides, r, t_min, t_max, rec);
    case H_LIST:
        return list_hit(h, r, t_min, t_max, rec);
    case H_FLIP: /* hittable.rs:78-83 */
        if (hit(h->child, r, t_min, t_max, rec)) {
            rec->normal = v_neg(rec->normal);
            return 1;
        }
        return 0;
    case H_TRANSLATE: { /* traslate.rs:18-24 */
        COUNT(C_XFORM);
        Ray moved = ray_new(v_sub(r->o, h->offset), r->d, r->time);
        if (hit(h->child, &moved, t_min, t_max, rec)) {
            rec->p = v_add(rec->p, h->offset);
            return 1;
        }
        return 0;
    }
    case H_ROTATE: { /* rotate.rs:85-113 */
        COUNT(C_XFORM);
        int ra, aa, ba;
        plane_axes(h->axis, &ra, &aa, &ba); /* Axis::{X,Y,Z} -> same (r,a,b) triples as Plane */
        V3 o = r->o, d = r->d;
        v_set(&o, aa, h->cos_t * v_get(r->o, aa) + h->sin_t * v_get(r->o, ba));
        v_set(&o, ba, -h->sin_t * v_get(r->o, aa) + h->cos_t * v_get(r->o, ba));
        v_set(&d, aa, h->cos_t * v_get(r->d, aa) + h->sin_t * v_get(r->d, ba));
        v_set(&d, ba, -h->sin_t * v_get(r->d, aa) + h->cos_t * v_get(r->d, ba));
        Ray rot = ray_new(o, d, r->time);
        if (hit(h->child, &rot, t_min, t_max, rec)) {
            V3 p = rec->p, n = rec->normal;
            v_set(&p, aa, h->cos_t * v_get(rec->p, aa) - h->sin_t * v_get(rec->p, ba));
            v_set(&p, ba, h->sin_t * v_get(rec->p, aa) + h->cos_t * v_get(rec->p, ba));
            v_set(&n, aa, h->cos_t * v_get(rec->normal, aa) - h->sin_t * v_get(rec->normal, ba));
            v_set(&n, ba, h->sin_t * v_get(rec->normal, aa) + h->cos_t * v_get(rec->normal, ba));
            rec->p = p;
            rec->normal = n;
            return 1;
        }
        return 0;
    }
    case H_MEDIUM: { /* medium.rs:28-56 */
        COUNT(C_MEDIUM);
        HitRecord h1, h2;
        if (hit(h->child, r, -R_MAX, R_MAX, &h1)) {
            if (hit(h->child, r, h1.t + (REAL)0.0001, R_MAX, &h2)) {
                if (h1.t < t_min) h1.t = t_min;
                if (h2.t > t_max) h2.t = t_max;
                if (h1.t < h2.t) {
                    REAL dist_inside = (h2.t - h1.t) * v_norm(r->d);
                    COUNT(C_MEDIUM_DRAW);
                    REAL hit_distance = -((REAL)1.0 / h->density) * m_log(rng_uniform());
                    if (hit_distance < dist_inside) {
                        REAL t = h1.t + hit_distance / v_norm(r->d);
                        rec->t = t;
                        rec->u = 0; rec->v = 0;
                        rec->p = ray_at(r, t);
                        rec->normal = v3(1.0, 0.0, 0.0);
                        rec->mat = &h->phase;
                        return 1;
                    }
                }
            }
        }
        return 0;
    }
    case H_BVH: { /* bvh.rs:70-89 */
        if (aabb_hit(&h->box, r, t_min, t_max)) {
            HitRecord l, rr;
            int hl = hit(h->left, r, t_min, t_max, &l);
            int hr = hit(h->right, r, t_min, t_max, &rr);
            if (hl && hr) { *rec = (l.t < rr.t) ? l : rr; return 1; } /* tie -> right */
            if (hl) { *rec = l; return 1; }
            if (hr) { *rec = rr; return 1; }
        }
        return 0;
    }
    }
    return 0;
}

/* ================================================================================== */
/* bounding_box (f64, scene set-up) — hittable.rs:49-64, aabb.rs:6-18, sphere.rs:79-84,  */
/* :165-174, rect.rs:71-75, cube.rs:88-93, traslate.rs:26-32, rotate.rs:115-117,         */
/* medium.rs:58-60, bvh.rs:91-93                                                         */
/* ================================================================================== */
static AABBd surrounding_box(AABBd a, AABBd b) {
    AABBd r;
    r.min.x = fmin(a.min.x, b.min.x); r.min.y = fmin(a.min.y, b.min.y); r.min.z = fmin(a.min.z, b.min.z);
    r.max.x = fmax(a.max.x, b.max.x); r.max.y = fmax(a.max.y, b.max.y); r.max.z = fmax(a.max.z, b.max.z);
    return r;
}
static int bounding_box(const Hittable *h, double t0, double t1, AABBd *out) {
    switch (h->kind) {
    case H_SPHERE:
        out->min.x = h->c0d.x - h->rd; out->min.y = h->c0d.y - h->rd; out->min.z = h->c0d.z - h->rd;
        out->max.x = h->c0d.x + h->rd; out->max.y = h->c0d.y + h->rd; out->max.z = h->c0d.z + h->rd;
        return 1;
    case H_MOVING_SPHERE: {
        D3 ca = moving_center_d(h, t0), cb = moving_center_d(h, t1);
        AABBd a = {{ca.x - h->rd, ca.y - h->rd, ca.z - h->rd}, {ca.x + h->rd, ca.y + h->rd, ca.z + h->rd}};
        AABBd b = {{cb.x - h->rd, cb.y - h->rd, cb.z - h->rd}, {cb.x + h->rd, cb.y + h->rd, cb.z + h->rd}};
        *out = surrounding_box(a, b);
        return 1;
    }
    case H_RECT: /* sic: ignores the plane (rect.rs:72-73) */
        out->min.x = h->x0d; out->min.y = h->y0d; out->min.z = h->kd - 0.0001;
        out->max.x = h->x1d; out->max.y = h->y1d; out->max.z = h->kd + 0.0001;
        return 1;
    case H_CUBE:
        out->min = h->pmind; out->max = h->pmaxd;
        return 1;
    case H_LIST: {
        if (h->n == 0) return 0;
        AABBd acc;
        if (!bounding_box(h->items[0], t0, t1, &acc)) return 0;
        for (int i = 1; i < h->n; i++) {
            AABBd b;
            if (!bounding_box(h->items[i], t0, t1, &b)) return 0;
            acc = surrounding_box(acc, b);
        }
        *out = acc;
        return 1;
    }
    case H_FLIP:
    case H_MEDIUM:
        return bounding_box(h->child, t0, t1, out);
    case H_TRANSLATE:
        if (!bounding_box(h->child, t0, t1, out)) return 0;
        out->min.x += h->offsetd.x; out->min.y += h->offsetd.y; out->min.z += h->offsetd.z;
        out->max.x += h->offsetd.x; out->max.y += h->offsetd.y; out->max.z += h->offsetd.z;
        return 1;
    case H_ROTATE:
        if (!h->has_bbox) return 0;
        *out = h->rot_bbox;
        return 1;
    case H_BVH:
        *out = h->boxd;
        return 1;
    }
    return 0;
}

/* ================================================================================== */
/* color — src/color.rs:6-23                                                           */
/* ================================================================================== */
typedef struct { const Hittable *world; int max_depth; REAL t_min; } RenderCtx;

/* Path signature (validation aid, see include/rtmi.h): every hit query that finds a hit adds
 * mix(bits of (float)t, bounce index) to the pixel's wrapping uint64 signature. */
static uint64_t g_sig;
static uint32_t sig_mix(uint32_t x, uint32_t k) {
    x ^= (k + 1u) * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
static void sig_add(REAL t, int depth) { g_sig += (uint64_t)sig_mix(rtmi_f2u((float)t), (uint32_t)depth); }

/* color.rs:18-20 (commented out in the reference; rendered only under ORC_SKY) */
static V3 sky_color(V3 d) {
    V3 unit = v_normalize(d);
    REAL t = (REAL)0.5 * (unit.y + (REAL)1.0);
    REAL a = (REAL)1.0 - t;
    return v3(a * (REAL)1.0 + t * (REAL)0.5, a * (REAL)1.0 + t * (REAL)0.7, a * (REAL)1.0 + t * (REAL)1.0);
}

static V3 color(const RenderCtx *cx, const Ray *ray, int depth) {
    HitRecord rec;
    COUNT(C_QUERIES);
    if (hit(cx->world, ray, cx->t_min, R_MAX, &rec)) {
        sig_add(rec.t, depth);
        V3 emitted = mat_emitted(rec.mat, rec.u, rec.v, rec.p);
        if (depth < cx->max_depth) {
            Ray scattered;
            V3 att;
            if (mat_scatter(rec.mat, ray, &rec, &scattered, &att)) {
                V3 c = color(cx, &scattered, depth + 1);
                return v_add(emitted, v_mul(att, c));
            }
        }
        return emitted;
    }
    if (g_flags & ORC_SKY) return sky_color(ray->d);
    return v3(0, 0, 0); /* background is black (color.rs:21) */
}

/* Same estimator, unrolled: L = sum_k (prod_{i<k} att_i) * emitted_k.  Equal to the
 * recursion in exact arithmetic; differs by rounding order only. */
static V3 color_throughput(const RenderCtx *cx, Ray ray) {
    V3 L = v3(0, 0, 0), T = v3(1, 1, 1);
    for (int depth = 0;; depth++) {
        HitRecord rec;
        COUNT(C_QUERIES);
        if (!hit(cx->world, &ray, cx->t_min, R_MAX, &rec)) {
            if (g_flags & ORC_SKY) L = v_add(L, v_mul(T, sky_color(ray.d)));
            break;
        }
        sig_add(rec.t, depth);
        V3 emitted = mat_emitted(rec.mat, rec.u, rec.v, rec.p);
        L = v_add(L, v_mul(T, emitted));
        if (depth >= cx->max_depth) break;
        Ray scattered;
        V3 att;
        if (!mat_scatter(rec.mat, &ray, &rec, &scattered, &att)) break;
        T = v_mul(T, att);
        ray = scattered;
    }
    return L;
}

/* ================================================================================== */
/* Camera — src/camera.rs                                                              */
/* ================================================================================== */
typedef struct {
    D3 origind, llcd, hord, verd, ud, vd;
    double time0d, time1d, lens_radiusd;
    V3 origin, llc, horizontal, vertical, u, v;
    REAL time0, time1, lens_radius;
} Camera;

/* camera.rs:53-67 */
static Ray camera_get_ray(const Camera *c, REAL s, REAL t) {
    V3 origin;
    if (c->lens_radius == 0) {
        origin = c->origin;
    } else {
        V3 rd = v_scale(random_in_unit_disk(), c->lens_radius);
        V3 offset = v_add(v_scale(c->u, rd.x), v_scale(c->v, rd.y));
        origin = v_add(c->origin, offset);
    }
    REAL time = c->time0 + rng_uniform() * (c->time1 - c->time0);
    V3 dir = v_sub(v_add(v_add(c->llc, v_scale(c->horizontal, s)), v_scale(c->vertical, t)), origin);
    return ray_new(origin, dir, time);
}

/* ================================================================================== */
/* exported builder API                                                                */
/* ================================================================================== */
ORC_API int orc_is_f32(void) { return ORC_IS_F32; }
ORC_API void orc_set_flags(int flags) { g_flags = flags; }
ORC_API void orc_seed_scene_rng(uint64_t seed) { stream_init(&g_scene_rng, seed, 0, 0, 1); }
ORC_API void orc_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }
ORC_API double orc_scene_uniform(void) { return scene_uniform(); }
ORC_API uint32_t orc_scene_range(uint32_t n) { return scene_range(n); }

static Texture *new_tex(int kind) { Texture *t = (Texture *)arena_alloc(sizeof(Texture)); t->kind = kind; return t; }
ORC_API void *orc_tex_solid(double r, double g, double b) {
    Texture *t = new_tex(TEX_SOLID);
    t->color = v3((REAL)r, (REAL)g, (REAL)b);
    return t;
}
ORC_API void *orc_tex_checker(void *odd, void *even) {
    Texture *t = new_tex(TEX_CHECKER);
    t->odd = (Texture *)odd; t->even = (Texture *)even;
    return t;
}
/* Perlin::new — perlin.rs:12-36, 67-74.  Tables are drawn in f64 (scene set-up), then
 * rounded once to the working precision. */
ORC_API void *orc_tex_noise(double scale) {
    Texture *t = new_tex(TEX_NOISE);
    t->scale = (REAL)scale;
    Perlin *pn = (Perlin *)arena_alloc(sizeof(Perlin));
    for (int i = 0; i < 256; i++) { /* perlin_generate :12-26 */
        double x = -1.0 + 2.0 * scene_uniform();
        double y = -1.0 + 2.0 * scene_uniform();
        double z = -1.0 + 2.0 * scene_uniform();
        double n = sqrt(x * x + y * y + z * z);
        pn->ran_vec[i] = v3((REAL)(x / n), (REAL)(y / n), (REAL)(z / n));
    }
    int *perms[3] = {pn->perm_x, pn->perm_y, pn->perm_z};
    for (int k = 0; k < 3; k++) { /* perlin_generate_perm :28-36 + permute :4-10 */
        int *p = perms[k];
        for (int i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i >= 0; i--) {
            uint32_t target = scene_range((uint32_t)i + 1);
            int tmp = p[i]; p[i] = p[target]; p[target] = tmp;
        }
    }
    t->noise = pn;
    return t;
}
ORC_API void *orc_tex_image(const uint8_t *data, uint32_t nx, uint32_t ny) {
    Texture *t = new_tex(TEX_IMAGE);
    size_t n = (size_t)nx * ny * 3;
    t->data = (uint8_t *)arena_alloc(n);
    memcpy(t->data, data, n);
    t->nx = nx; t->ny = ny;
    return t;
}
ORC_API void orc_perlin_tables(void *tex, double *ranvec768, int *perm768) {
    Texture *t = (Texture *)tex;
    for (int i = 0; i < 256; i++) {
        ranvec768[3 * i] = t->noise->ran_vec[i].x; ranvec768[3 * i + 1] = t->noise->ran_vec[i].y;
        ranvec768[3 * i + 2] = t->noise->ran_vec[i].z;
        perm768[i] = t->noise->perm_x[i]; perm768[256 + i] = t->noise->perm_y[i]; perm768[512 + i] = t->noise->perm_z[i];
    }
}

static Material *new_mat(int kind, void *tex, double param) {
    Material *m = (Material *)arena_alloc(sizeof(Material));
    m->kind = kind; m->tex = (Texture *)tex; m->param = (REAL)param;
    return m;
}
ORC_API void *orc_mat_lambertian(void *tex) { return new_mat(MAT_LAMBERTIAN, tex, 0); }
ORC_API void *orc_mat_metal(void *tex, double fuzz) { return new_mat(MAT_METAL, tex, fuzz < 1.0 ? fuzz : 1.0); } /* :70 */
ORC_API void *orc_mat_dielectric(double ref_idx) { return new_mat(MAT_DIELECTRIC, NULL, ref_idx); }
ORC_API void *orc_mat_diffuse_light(void *tex) { return new_mat(MAT_DIFFUSE_LIGHT, tex, 0); }
ORC_API void *orc_mat_isotropic(void *tex) { return new_mat(MAT_ISOTROPIC, tex, 0); }

static Hittable *new_hit(int kind) { Hittable *h = (Hittable *)arena_alloc(sizeof(Hittable)); h->kind = kind; return h; }
ORC_API void *orc_sphere(double cx, double cy, double cz, double r, void *mat) {
    Hittable *h = new_hit(H_SPHERE);
    h->c0d = (D3){cx, cy, cz}; h->rd = r;
    h->c0 = d3_to_v3(h->c0d); h->radius = (REAL)r; h->mat = (Material *)mat;
    return h;
}
ORC_API void *orc_moving_sphere(double c0x, double c0y, double c0z, double c1x, double c1y, double c1z, double t0,
                                double t1, double r, void *mat) {
    Hittable *h = new_hit(H_MOVING_SPHERE);
    h->c0d = (D3){c0x, c0y, c0z}; h->c1d = (D3){c1x, c1y, c1z}; h->rd = r; h->t0d = t0; h->t1d = t1;
    h->c0 = d3_to_v3(h->c0d); h->c1 = d3_to_v3(h->c1d); h->radius = (REAL)r; h->t0 = (REAL)t0; h->t1 = (REAL)t1;
    h->mat = (Material *)mat;
    return h;
}
static Hittable *make_rect(int plane, double x0, double y0, double x1, double y1, double k, const Material *mat) {
    Hittable *h = new_hit(H_RECT);
    h->plane = plane;
    h->x0d = x0; h->y0d = y0; h->x1d = x1; h->y1d = y1; h->kd = k;
    h->x0 = (REAL)x0; h->y0 = (REAL)y0; h->x1 = (REAL)x1; h->y1 = (REAL)y1; h->k = (REAL)k;
    h->mat = mat;
    return h;
}
ORC_API void *orc_rect(int plane, double x0, double y0, double x1, double y1, double k, void *mat) {
    return make_rect(plane, x0, y0, x1, y1, k, (Material *)mat);
}
static void list_push(Hittable *l, Hittable *h) {
    if (l->n == l->cap) {
        int ncap = l->cap ? l->cap * 2 : 8;
        Hittable **ni = (Hittable **)arena_alloc(sizeof(Hittable *) * ncap);
        if (l->n) memcpy(ni, l->items, sizeof(Hittable *) * l->n);
        l->items = ni; l->cap = ncap;
    }
    l->items[l->n++] = h;
}
ORC_API void *orc_list_new(void) { return new_hit(H_LIST); }
ORC_API void orc_list_push(void *list, void *h) { list_push((Hittable *)list, (Hittable *)h); }
/* cube.rs:15-80 : XY@max.z, XY@min.z, ZX@max.y, ZX@min.y, YZ@max.x, YZ@min.x — none flipped */
ORC_API void *orc_cube(double ax, double ay, double az, double bx, double by, double bz, void *mat) {
    Hittable *h = new_hit(H_CUBE);
    const Material *m = (Material *)mat;
    h->pmind = (D3){ax, ay, az}; h->pmaxd = (D3){bx, by, bz};
    Hittable *s = new_hit(H_LIST);
    list_push(s, make_rect(2, ax, ay, bx, by, bz, m));
    list_push(s, make_rect(2, ax, ay, bx, by, az, m));
    list_push(s, make_rect(1, az, ax, bz, bx, by, m));
    list_push(s, make_rect(1, az, ax, bz, bx, ay, m));
    list_push(s, make_rect(0, ay, az, by, bz, bx, m));
    list_push(s, make_rect(0, ay, az, by, bz, ax, m));
    h->sides = s;
    return h;
}
ORC_API void *orc_flip_normals(void *child) { Hittable *h = new_hit(H_FLIP); h->child = (Hittable *)child; return h; }
ORC_API void *orc_translate(void *child, double ox, double oy, double oz) {
    Hittable *h = new_hit(H_TRANSLATE);
    h->child = (Hittable *)child; h->offsetd = (D3){ox, oy, oz}; h->offset = d3_to_v3(h->offsetd);
    return h;
}
/* rotate.rs:30-81 ; the bbox loop never updates (min starts at f64::MIN, max at f64::MAX) */
ORC_API void *orc_rotate(int axis, void *child, double angle) {
    Hittable *h = new_hit(H_ROTATE);
    h->axis = axis; h->child = (Hittable *)child;
    double radians = (3.14159265358979323846264338327950288 / 180.0) * angle;
    h->sind = sin(radians); h->cosd = cos(radians);
    h->sin_t = (REAL)h->sind; h->cos_t = (REAL)h->cosd;
    AABBd b;
    h->has_bbox = bounding_box(h->child, 0.0, 1.0, &b);
    if (h->has_bbox) {
        const double MX = 1.79769313486231570814527423731704357e+308;
        h->rot_bbox.min = (D3){-MX, -MX, -MX};
        h->rot_bbox.max = (D3){MX, MX, MX};
    }
    return h;
}
ORC_API void *orc_constant_medium(void *boundary, double density, void *tex) {
    Hittable *h = new_hit(H_MEDIUM);
    h->child = (Hittable *)boundary; h->density = (REAL)density;
    h->phase.kind = MAT_ISOTROPIC; h->phase.tex = (Texture *)tex; h->phase.param = 0;
    return h;
}

/* BVHNode::new — bvh.rs:17-66.  `sort_unstable_by` with the reference's Less/Greater-only
 * comparator has no defined result on ties; this restatement sorts STABLY by
 * bbox.min[axis] (strict `a - b < 0`), which is one valid outcome. */
static int g_bvh_error = 0;
static void stable_sort(Hittable **v, double *key, int n) {
    /* insertion sort is O(n^2) but n <= a few thousand at set-up; merge for larger */
    if (n < 2) return;
    Hittable **tv = (Hittable **)malloc(sizeof(Hittable *) * n);
    double *tk = (double *)malloc(sizeof(double) * n);
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (key[j] - key[i] < 0.0) { tv[o] = v[j]; tk[o++] = key[j++]; }
                else { tv[o] = v[i]; tk[o++] = key[i++]; }
            }
            while (i < mid) { tv[o] = v[i]; tk[o++] = key[i++]; }
            while (j < hi) { tv[o] = v[j]; tk[o++] = key[j++]; }
        }
        memcpy(v, tv, sizeof(Hittable *) * n);
        memcpy(key, tk, sizeof(double) * n);
    }
    free(tv); free(tk);
}
static Hittable *bvh_new(Hittable **list, int len, double t0, double t1) {
    uint32_t axis = scene_range(3); /* bvh.rs:40 */
    double *key = (double *)malloc(sizeof(double) * len);
    for (int i = 0; i < len; i++) {
        AABBd b;
        if (!bounding_box(list[i], t0, t1, &b)) { g_bvh_error = 1; b.min = (D3){0, 0, 0}; }
        key[i] = axis == 0 ? b.min.x : (axis == 1 ? b.min.y : b.min.z);
    }
    stable_sort(list, key, len);
    free(key);
    Hittable *h = new_hit(H_BVH);
    if (len == 1) { h->left = list[0]; h->right = list[0]; }
    else if (len == 2) { h->left = list[0]; h->right = list[1]; }
    else {
        h->left = bvh_new(list, len / 2, t0, t1);
        h->right = bvh_new(list + len / 2, len - len / 2, t0, t1);
    }
    AABBd lb, rb;
    if (!bounding_box(h->left, t0, t1, &lb) || !bounding_box(h->right, t0, t1, &rb)) { g_bvh_error = 1; return h; }
    h->boxd = surrounding_box(lb, rb);
    h->box.min = d3_to_v3(h->boxd.min);
    h->box.max = d3_to_v3(h->boxd.max);
    return h;
}
/* returns NULL where the reference panics ("No bounding box in BVHNode", bvh.rs:30,58) */
ORC_API void *orc_bvh(void **items, int n, double t0, double t1) {
    if (n <= 0) return NULL;
    Hittable **tmp = (Hittable **)malloc(sizeof(Hittable *) * n);
    memcpy(tmp, items, sizeof(Hittable *) * n);
    g_bvh_error = 0;
    Hittable *h = bvh_new(tmp, n, t0, t1);
    free(tmp);
    return g_bvh_error ? NULL : h;
}

/* Camera::new — camera.rs:21-51 (f64, then rounded once to the working precision) */
ORC_API void *orc_camera(double fx, double fy, double fz, double ax, double ay, double az, double ux, double uy,
                         double uz, double vfov, double aspect, double aperture, double focus_dist, double time0,
                         double time1) {
    Camera *c = (Camera *)arena_alloc(sizeof(Camera));
    double theta = vfov * 3.14159265358979323846264338327950288 / 180.0;
    double half_height = focus_dist * tan(theta / 2.0);
    double half_width = aspect * half_height;
    double wx = fx - ax, wy = fy - ay, wz = fz - az;
    double wn = sqrt(wx * wx + wy * wy + wz * wz);
    wx /= wn; wy /= wn; wz /= wn;
    double cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx; /* view_up x w */
    double cn = sqrt(cx * cx + cy * cy + cz * cz);
    cx /= cn; cy /= cn; cz /= cn;
    double vx = wy * cz - wz * cy, vy = wz * cx - wx * cz, vz = wx * cy - wy * cx; /* w x u */
    c->origind = (D3){fx, fy, fz};
    c->llcd = (D3){fx - half_width * cx - half_height * vx - focus_dist * wx,
                   fy - half_width * cy - half_height * vy - focus_dist * wy,
                   fz - half_width * cz - half_height * vz - focus_dist * wz};
    c->hord = (D3){2.0 * half_width * cx, 2.0 * half_width * cy, 2.0 * half_width * cz};
    c->verd = (D3){2.0 * half_height * vx, 2.0 * half_height * vy, 2.0 * half_height * vz};
    c->ud = (D3){cx, cy, cz};
    c->vd = (D3){vx, vy, vz};
    c->time0d = time0; c->time1d = time1; c->lens_radiusd = aperture / 2.0;
    c->origin = d3_to_v3(c->origind); c->llc = d3_to_v3(c->llcd);
    c->horizontal = d3_to_v3(c->hord); c->vertical = d3_to_v3(c->verd);
    c->u = d3_to_v3(c->ud); c->v = d3_to_v3(c->vd);
    c->time0 = (REAL)time0; c->time1 = (REAL)time1; c->lens_radius = (REAL)c->lens_radiusd;
    return c;
}
ORC_API void orc_camera_state(void *cam, double *out21) {
    Camera *c = (Camera *)cam;
    D3 *v[6] = {&c->origind, &c->llcd, &c->hord, &c->verd, &c->ud, &c->vd};
    for (int i = 0; i < 6; i++) { out21[3 * i] = v[i]->x; out21[3 * i + 1] = v[i]->y; out21[3 * i + 2] = v[i]->z; }
    out21[18] = c->time0d; out21[19] = c->time1d; out21[20] = c->lens_radiusd;
}

/* ================================================================================== */
/* create_image — tests/test.rs:55-85                                                  */
/* ================================================================================== */
/* nalgebra::clamp(val, min, max) (tests/test.rs:74): NaN -> min */
static double na_clamp(double val, double lo, double hi) {
    if (val > lo) { if (val < hi) return val; return hi; }
    return lo;
}
/* Rust `f64 as i32` (tests/test.rs:76-78): truncation, saturating, NaN -> 0 */
static int32_t as_i32(double x) {
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}

/* Renders output rows [row_begin,row_end) (row 0 = top = reference j = ny-1) and, inside
 * them, samples [0,ns).  out_linear: ny*nx*3 float (mean radiance before gamma);
 * out_rgb: ny*nx*3 int32 (the ir/ig/ib the reference prints); out_mean: ny*nx*3 double.
 * out_sig: ny*nx uint64 path signatures.  Any output pointer may be NULL. */
ORC_API int orc_render(void *cam_, void *world_, int nx, int ny, int ns, uint64_t seed, int flags, int max_depth,
                       double t_min, int row_begin, int row_end, float *out_linear, int32_t *out_rgb,
                       double *out_mean, uint64_t *out_sig) {
    const Camera *cam = (Camera *)cam_;
    RenderCtx cx;
    cx.world = (Hittable *)world_;
    cx.max_depth = max_depth;
    cx.t_min = (REAL)t_min;
    g_flags = flags;
    if (row_begin < 0) row_begin = 0;
    if (row_end > ny) row_end = ny;
    for (int row = row_begin; row < row_end; row++) {
        int j = ny - 1 - row; /* for j in (0..ny).rev() */
        for (int i = 0; i < nx; i++) {
            double col[3] = {0.0, 0.0, 0.0};
            g_sig = 0;
            for (int s = 0; s < ns; s++) {
                stream_init(&g_rng, seed, (uint32_t)s, (uint32_t)(j * nx + i), 0);
                COUNT(C_SAMPLES);
                REAL u = ((REAL)i + rng_uniform()) / (REAL)nx;
                REAL v = ((REAL)j + rng_uniform()) / (REAL)ny;
                Ray ray = camera_get_ray(cam, u, v);
                V3 c = (flags & ORC_THROUGHPUT_FORM) ? color_throughput(&cx, ray) : color(&cx, &ray, 0);
                col[0] += (double)c.x; col[1] += (double)c.y; col[2] += (double)c.z;
            }
            size_t o = ((size_t)row * nx + i) * 3;
            if (out_sig) out_sig[(size_t)row * nx + i] = g_sig;
            for (int ch = 0; ch < 3; ch++) {
                double m = col[ch] / (double)ns;
                if (out_mean) out_mean[o + ch] = m;
                if (out_linear) out_linear[o + ch] = (float)m;
                if (out_rgb) out_rgb[o + ch] = as_i32(255.99 * na_clamp(sqrt(m), 0.0, 1.0));
            }
        }
    }
    return 0;
}

/* P3 text exactly as tests/test.rs:59,79 : "P3\n{nx} {ny}\n255\n" then "{ir} {ig} {ib}\n" per pixel.
 * Returns bytes written (excluding the NUL), or the size needed if cap is too small. */
ORC_API size_t orc_ppm_text(int nx, int ny, const int32_t *rgb, char *buf, size_t cap) {
    size_t need = 32 + (size_t)nx * ny * 36;
    if (!buf || cap < need) return need;
    size_t n = (size_t)sprintf(buf, "P3\n%d %d\n255\n", nx, ny);
    for (size_t p = 0; p < (size_t)nx * ny; p++)
        n += (size_t)sprintf(buf + n, "%d %d %d\n", rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2]);
    return n;
}

/* ---- probes for known-answer tests ------------------------------------------------ */
ORC_API int orc_hit(void *h, const double *o, const double *d, double time, double t_min, double t_max, int flags,
                    uint64_t seed, double *out9, int *out_mat_kind) {
    g_flags = flags;
    stream_init(&g_rng, seed, 0, 0, 0);
    Ray r = ray_new(v3((REAL)o[0], (REAL)o[1], (REAL)o[2]), v3((REAL)d[0], (REAL)d[1], (REAL)d[2]), (REAL)time);
    HitRecord rec;
    REAL tmn = t_min <= -1.7e308 ? -R_MAX : (REAL)t_min;
    REAL tmx = t_max >= 1.7e308 ? R_MAX : (REAL)t_max;
    if (!hit((Hittable *)h, &r, tmn, tmx, &rec)) return 0;
    out9[0] = rec.t; out9[1] = rec.u; out9[2] = rec.v;
    out9[3] = rec.p.x; out9[4] = rec.p.y; out9[5] = rec.p.z;
    out9[6] = rec.normal.x; out9[7] = rec.normal.y; out9[8] = rec.normal.z;
    if (out_mat_kind) *out_mat_kind = rec.mat ? rec.mat->kind : -1;
    return 1;
}
ORC_API int orc_bounding_box(void *h, double t0, double t1, double *out6) {
    AABBd b;
    if (!bounding_box((Hittable *)h, t0, t1, &b)) return 0;
    out6[0] = b.min.x; out6[1] = b.min.y; out6[2] = b.min.z; out6[3] = b.max.x; out6[4] = b.max.y; out6[5] = b.max.z;
    return 1;
}
ORC_API void orc_tex_value(void *tex, double u, double v, const double *p, int flags, double *out3) {
    g_flags = flags;
    V3 c = tex_value((Texture *)tex, (REAL)u, (REAL)v, v3((REAL)p[0], (REAL)p[1], (REAL)p[2]));
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
/* scatter probe: returns 0 if absorbed; out = scattered o(3) d(3) time, attenuation(3) */
ORC_API int orc_scatter(void *mat, const double *ro, const double *rd, double time, const double *rec9, int flags,
                        uint64_t seed, double *out10) {
    g_flags = flags;
    stream_init(&g_rng, seed, 0, 0, 0);
    Ray r = ray_new(v3((REAL)ro[0], (REAL)ro[1], (REAL)ro[2]), v3((REAL)rd[0], (REAL)rd[1], (REAL)rd[2]), (REAL)time);
    HitRecord rec;
    rec.t = (REAL)rec9[0]; rec.u = (REAL)rec9[1]; rec.v = (REAL)rec9[2];
    rec.p = v3((REAL)rec9[3], (REAL)rec9[4], (REAL)rec9[5]);
    rec.normal = v3((REAL)rec9[6], (REAL)rec9[7], (REAL)rec9[8]);
    rec.mat = (Material *)mat;
    Ray sc; V3 att;
    if (!mat_scatter(rec.mat, &r, &rec, &sc, &att)) return 0;
    out10[0] = sc.o.x; out10[1] = sc.o.y; out10[2] = sc.o.z; out10[3] = sc.d.x; out10[4] = sc.d.y; out10[5] = sc.d.z;
    out10[6] = sc.time; out10[7] = att.x; out10[8] = att.y; out10[9] = att.z;
    return 1;
}
ORC_API void orc_emitted(void *mat, double u, double v, const double *p, double *out3) {
    V3 c = mat_emitted((Material *)mat, (REAL)u, (REAL)v, v3((REAL)p[0], (REAL)p[1], (REAL)p[2]));
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
ORC_API void orc_get_ray(void *cam, double s, double t, uint64_t seed, double *out7) {
    stream_init(&g_rng, seed, 0, 0, 0);
    Ray r = camera_get_ray((Camera *)cam, (REAL)s, (REAL)t);
    out7[0] = r.o.x; out7[1] = r.o.y; out7[2] = r.o.z; out7[3] = r.d.x; out7[4] = r.d.y; out7[5] = r.d.z; out7[6] = r.time;
}
ORC_API void orc_reset_counters(void) { memset(g_cnt, 0, sizeof(g_cnt)); }
ORC_API int orc_get_counters(uint64_t *out, int n) {
    for (int i = 0; i < n && i < C_NCOUNTERS; i++) out[i] = g_cnt[i];
    return C_NCOUNTERS;
}
/* host evaluation of the fp32 transcendental contract, for tests */
ORC_API float orc_rtmi_sinf(float x) { return rtmi_sinf(x); }
ORC_API float orc_rtmi_logf(float x) { return rtmi_logf(x); }
ORC_API float orc_rtmi_atan2f(float y, float x) { return rtmi_atan2f(y, x); }
ORC_API float orc_rtmi_asinf(float x) { return rtmi_asinf(x); }
