"""ctypes binding of the CPU oracle (oracle/rt_oracle.c) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (raytracing_rust_amd/) never does.

The class exposes the reference's constructor names (Sphere, Rect, Lambertian, ...;
reference: src/*.rs `new` functions) so that the scene builders in
raytracing_rust_amd/scenes.py can be run against either backend.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

ARITH_DEVICE = 1
THROUGHPUT_FORM = 2
SKY = 4  # opt-in extension: gradient background of color.rs:18-20 (commented out in the reference)
FACE_FORWARD = 8  # opt-in extension: opaque materials scatter about the normal turned against the ray
UV_BOOK = 16  # opt-in extension: get_sphere_uv with pi/2 (the book) instead of FRAC_2_PI (sphere.rs:13)

COUNTER_NAMES = [
    "samples", "queries", "aabb", "sphere", "msphere", "rect", "xform", "medium", "medium_draw",
    "mat_fetch", "tex_solid", "tex_checker", "tex_noise", "tex_image", "sc_lambert", "sc_metal",
    "sc_dielectric", "sc_isotropic", "emit", "draws", "sphere_trials", "disk_trials", "sphere_accept",
    "rect_accept",
]

PLANE_YZ, PLANE_ZX, PLANE_XY = 0, 1, 2
AXIS_X, AXIS_Y, AXIS_Z = 0, 1, 2


def build():
    """Compile both oracle libraries (gcc, a few seconds)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL)


_NATIVE_BUILT = False


def build_native():
    """The timed CPU-baseline library: no counters, -march=native.  Rebuilt once per process tree on the host
    that runs it (the file may have been compiled on another host); forked workers inherit the flag."""
    global _NATIVE_BUILT
    if not _NATIVE_BUILT:
        subprocess.run(["make", "-B", "-C", _HERE, "native"], check=True, stdout=subprocess.DEVNULL)
        _NATIVE_BUILT = True


def _load(name):
    path = os.path.join(_BUILD, name)
    if name == "liborc_f64_native.so":
        build_native()
    elif not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    vp, d, i, u64 = C.c_void_p, C.c_double, C.c_int, C.c_uint64
    sig = {
        "orc_is_f32": (i, []),
        "orc_set_flags": (None, [i]),
        "orc_seed_scene_rng": (None, [u64]),
        "orc_scene_uniform": (d, []),
        "orc_scene_range": (C.c_uint32, [C.c_uint32]),
        "orc_philox": (None, [vp, vp, vp]),
        "orc_free_all": (None, []),
        "orc_tex_solid": (vp, [d, d, d]),
        "orc_tex_checker": (vp, [vp, vp]),
        "orc_tex_noise": (vp, [d]),
        "orc_tex_image": (vp, [vp, C.c_uint32, C.c_uint32]),
        "orc_perlin_tables": (None, [vp, vp, vp]),
        "orc_mat_lambertian": (vp, [vp]),
        "orc_mat_metal": (vp, [vp, d]),
        "orc_mat_dielectric": (vp, [d]),
        "orc_mat_diffuse_light": (vp, [vp]),
        "orc_mat_isotropic": (vp, [vp]),
        "orc_sphere": (vp, [d, d, d, d, vp]),
        "orc_moving_sphere": (vp, [d] * 9 + [vp]),
        "orc_rect": (vp, [i, d, d, d, d, d, vp]),
        "orc_cube": (vp, [d] * 6 + [vp]),
        "orc_list_new": (vp, []),
        "orc_list_push": (None, [vp, vp]),
        "orc_flip_normals": (vp, [vp]),
        "orc_translate": (vp, [vp, d, d, d]),
        "orc_rotate": (vp, [i, vp, d]),
        "orc_constant_medium": (vp, [vp, d, vp]),
        "orc_bvh": (vp, [vp, i, d, d]),
        "orc_camera": (vp, [d] * 15),
        "orc_camera_state": (None, [vp, vp]),
        "orc_render": (i, [vp, vp, i, i, i, u64, i, i, d, i, i, vp, vp, vp, vp]),
        "orc_ppm_text": (C.c_size_t, [i, i, vp, vp, C.c_size_t]),
        "orc_hit": (i, [vp, vp, vp, d, d, d, i, u64, vp, vp]),
        "orc_bounding_box": (i, [vp, d, d, vp]),
        "orc_tex_value": (None, [vp, d, d, vp, i, vp]),
        "orc_scatter": (i, [vp, vp, vp, d, vp, i, u64, vp]),
        "orc_emitted": (None, [vp, d, d, vp, vp]),
        "orc_get_ray": (None, [vp, d, d, u64, vp]),
        "orc_reset_counters": (None, []),
        "orc_get_counters": (i, [vp, i]),
        "orc_rtmi_sinf": (C.c_float, [C.c_float]),
        "orc_rtmi_logf": (C.c_float, [C.c_float]),
        "orc_rtmi_atan2f": (C.c_float, [C.c_float, C.c_float]),
        "orc_rtmi_asinf": (C.c_float, [C.c_float]),
    }
    for name_, (res, args) in sig.items():
        fn = getattr(lib, name_)
        fn.restype = res
        fn.argtypes = args
    return lib


class _Obj:
    """Opaque handle to an oracle object (lives until Oracle.free_all())."""

    __slots__ = ("h", "keep")

    def __init__(self, h, keep=()):
        if not h:
            raise RuntimeError("oracle constructor failed (reference would panic here)")
        self.h = h
        self.keep = keep


class _List(_Obj):
    def push(self, hittable):
        self.keep = self.keep + (hittable,)
        self._lib.orc_list_push(self.h, hittable.h)


def _d3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


class Oracle:
    """One precision build of the oracle.  precision = 'f64' (literal) or 'f32'."""

    PLANE_YZ, PLANE_ZX, PLANE_XY = PLANE_YZ, PLANE_ZX, PLANE_XY
    AXIS_X, AXIS_Y, AXIS_Z = AXIS_X, AXIS_Y, AXIS_Z

    def __init__(self, precision="f64", native=False):
        """native=True: the uninstrumented -march=native f64 build that bench.py times as the CPU baseline."""
        assert precision in ("f64", "f32")
        assert not native or precision == "f64"
        self.precision = precision
        self.lib = _load("liborc_f64_native.so" if native else "liborc_%s.so" % precision)
        assert self.lib.orc_is_f32() == (1 if precision == "f32" else 0)

    # ---- scene RNG (the reference's thread_rng during construction) ----
    def seed_scene_rng(self, seed):
        self.lib.orc_seed_scene_rng(int(seed))

    def free_all(self):
        self.lib.orc_free_all()

    # ---- textures (src/texture.rs) ----
    def SolidTexture(self, r, g, b):
        return _Obj(self.lib.orc_tex_solid(r, g, b))

    def CheckerTexture(self, odd, even):
        return _Obj(self.lib.orc_tex_checker(odd.h, even.h), (odd, even))

    def NoiseTexture(self, scale):
        return _Obj(self.lib.orc_tex_noise(scale))

    def ImageTexture(self, data, nx, ny):
        arr = np.ascontiguousarray(np.asarray(data, dtype=np.uint8).reshape(-1))
        assert arr.size == nx * ny * 3
        return _Obj(self.lib.orc_tex_image(arr.ctypes.data, nx, ny))

    # ---- materials (src/material.rs) ----
    def Lambertian(self, tex):
        return _Obj(self.lib.orc_mat_lambertian(tex.h), (tex,))

    def Metal(self, tex, fuzz):
        return _Obj(self.lib.orc_mat_metal(tex.h, fuzz), (tex,))

    def Dielectric(self, ref_idx):
        return _Obj(self.lib.orc_mat_dielectric(ref_idx))

    def DiffuseLight(self, tex):
        return _Obj(self.lib.orc_mat_diffuse_light(tex.h), (tex,))

    def Isotropic(self, tex):
        return _Obj(self.lib.orc_mat_isotropic(tex.h), (tex,))

    # ---- hittables ----
    def Sphere(self, center, radius, material):
        c = _d3(center)
        return _Obj(self.lib.orc_sphere(c[0], c[1], c[2], radius, material.h), (material,))

    def MovingSphere(self, center0, center1, time0, time1, radius, material):
        a, b = _d3(center0), _d3(center1)
        return _Obj(self.lib.orc_moving_sphere(a[0], a[1], a[2], b[0], b[1], b[2], time0, time1, radius,
                                               material.h), (material,))

    def Rect(self, plane, x0, y0, x1, y1, k, material):
        return _Obj(self.lib.orc_rect(plane, x0, y0, x1, y1, k, material.h), (material,))

    def Cube(self, p_min, p_max, material):
        a, b = _d3(p_min), _d3(p_max)
        return _Obj(self.lib.orc_cube(a[0], a[1], a[2], b[0], b[1], b[2], material.h), (material,))

    def FlipNormals(self, hittable):
        return _Obj(self.lib.orc_flip_normals(hittable.h), (hittable,))

    def Traslate(self, hittable, offset):
        o = _d3(offset)
        return _Obj(self.lib.orc_translate(hittable.h, o[0], o[1], o[2]), (hittable,))

    def Rotate(self, axis, hittable, angle):
        return _Obj(self.lib.orc_rotate(axis, hittable.h, angle), (hittable,))

    def ConstantMedium(self, boundary, density, texture):
        return _Obj(self.lib.orc_constant_medium(boundary.h, density, texture.h), (boundary, texture))

    def HittableList(self):
        l = _List(self.lib.orc_list_new())
        l._lib = self.lib
        return l

    def BVHNode(self, hittables, time0, time1):
        arr = (C.c_void_p * len(hittables))(*[h.h for h in hittables])
        return _Obj(self.lib.orc_bvh(arr, len(hittables), time0, time1), tuple(hittables))

    def Camera(self, look_from, look_at, view_up, vertical_fov, aspect, aperture, focus_dist, time0, time1):
        f, a, u = _d3(look_from), _d3(look_at), _d3(view_up)
        return _Obj(self.lib.orc_camera(f[0], f[1], f[2], a[0], a[1], a[2], u[0], u[1], u[2], vertical_fov, aspect,
                                        aperture, focus_dist, time0, time1))

    # ---- the hot path ----
    def render(self, cam, world, nx, ny, ns, seed=42, flags=0, max_depth=50, t_min=0.001, rows=None):
        """create_image (tests/test.rs:55-85).  Returns dict(linear f32 [ny,nx,3],
        rgb int32 [ny,nx,3], mean f64 [ny,nx,3], sig u64 [ny,nx] path signatures); row 0 is the top row."""
        r0, r1 = (0, ny) if rows is None else rows
        lin = np.zeros((ny, nx, 3), np.float32)
        rgb = np.zeros((ny, nx, 3), np.int32)
        mean = np.zeros((ny, nx, 3), np.float64)
        sig = np.zeros((ny, nx), np.uint64)
        rc = self.lib.orc_render(cam.h, world.h, nx, ny, ns, int(seed), flags, max_depth, t_min, r0, r1,
                                 lin.ctypes.data, rgb.ctypes.data, mean.ctypes.data, sig.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_render failed")
        return {"linear": lin, "rgb": rgb, "mean": mean, "sig": sig}

    def ppm_text(self, rgb):
        ny, nx = rgb.shape[:2]
        rgb = np.ascontiguousarray(rgb, dtype=np.int32)
        need = self.lib.orc_ppm_text(nx, ny, rgb.ctypes.data, None, 0)
        buf = C.create_string_buffer(need)
        n = self.lib.orc_ppm_text(nx, ny, rgb.ctypes.data, buf, need)
        return buf.raw[:n]

    # ---- probes ----
    def hit(self, hittable, origin, direction, time=0.0, t_min=0.001, t_max=float("inf"), flags=0, seed=0):
        o, d = _d3(origin), _d3(direction)
        out = np.zeros(9)
        mk = C.c_int(-1)
        tmx = 1.8e308 if t_max == float("inf") else t_max
        tmn = -1.8e308 if t_min == -float("inf") else t_min
        ok = self.lib.orc_hit(hittable.h, o.ctypes.data, d.ctypes.data, time, tmn, tmx, flags, seed,
                              out.ctypes.data, C.byref(mk))
        if not ok:
            return None
        return {"t": out[0], "u": out[1], "v": out[2], "p": out[3:6].copy(), "normal": out[6:9].copy(),
                "mat_kind": mk.value}

    def bounding_box(self, hittable, t0=0.0, t1=1.0):
        out = np.zeros(6)
        if not self.lib.orc_bounding_box(hittable.h, t0, t1, out.ctypes.data):
            return None
        return out[:3].copy(), out[3:].copy()

    def tex_value(self, tex, u, v, p, flags=0):
        pp = _d3(p)
        out = np.zeros(3)
        self.lib.orc_tex_value(tex.h, u, v, pp.ctypes.data, flags, out.ctypes.data)
        return out

    def scatter(self, mat, ray_o, ray_d, time, rec, flags=0, seed=0):
        o, d = _d3(ray_o), _d3(ray_d)
        r9 = np.ascontiguousarray(np.concatenate([[rec["t"], rec["u"], rec["v"]], rec["p"], rec["normal"]]),
                                  dtype=np.float64)
        out = np.zeros(10)
        ok = self.lib.orc_scatter(mat.h, o.ctypes.data, d.ctypes.data, time, r9.ctypes.data, flags, seed,
                                  out.ctypes.data)
        if not ok:
            return None
        return {"o": out[0:3].copy(), "d": out[3:6].copy(), "time": out[6], "attenuation": out[7:10].copy()}

    def emitted(self, mat, u, v, p):
        pp = _d3(p)
        out = np.zeros(3)
        self.lib.orc_emitted(mat.h, u, v, pp.ctypes.data, out.ctypes.data)
        return out

    def get_ray(self, cam, s, t, seed=0):
        out = np.zeros(7)
        self.lib.orc_get_ray(cam.h, s, t, seed, out.ctypes.data)
        return out

    def camera_state(self, cam):
        out = np.zeros(21)
        self.lib.orc_camera_state(cam.h, out.ctypes.data)
        return out

    def perlin_tables(self, tex):
        rv = np.zeros(768)
        pm = np.zeros(768, np.int32)
        self.lib.orc_perlin_tables(tex.h, rv.ctypes.data, pm.ctypes.data)
        return rv.reshape(256, 3), pm.reshape(3, 256)

    def philox(self, ctr, key):
        c = np.asarray(ctr, np.uint32)
        k = np.asarray(key, np.uint32)
        o = np.zeros(4, np.uint32)
        self.lib.orc_philox(c.ctypes.data, k.ctypes.data, o.ctypes.data)
        return o

    def reset_counters(self):
        self.lib.orc_reset_counters()

    def counters(self):
        out = np.zeros(len(COUNTER_NAMES), np.uint64)
        self.lib.orc_get_counters(out.ctypes.data, len(COUNTER_NAMES))
        return dict(zip(COUNTER_NAMES, (int(x) for x in out)))
