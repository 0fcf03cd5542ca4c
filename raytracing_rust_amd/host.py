"""Python face of the C++ host mirror (host/rt_host.hpp): the reference's type names
(src/*.rs `new` functions) bound through the C bindings of librt_host.so.

    host = Host()
    world = host.HittableList(); world.push(host.Sphere((0, -10, 0), 10, host.Lambertian(tex)))
    cam = host.Camera(look_from, look_at, vup, vfov, aspect, aperture, focus_dist, t0, t1)
    img = cam.render(world, nx, ny, ns)            # Camera::render — runs on the MI355X
    ppm = host.create_image(ny, nx, ns, cam, world)  # tests/test.rs:55 — P3 text

Objects are evaluated on the CPU in f64 (hit / scatter / value, like the reference) or
lowered to the flat scene of include/rtmi.h and rendered by the HIP kernels.  There is
no CPU fallback for rendering: without the extension or without a GPU, render raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

PLANE_YZ, PLANE_ZX, PLANE_XY = 0, 1, 2
AXIS_X, AXIS_Y, AXIS_Z = 0, 1, 2


class HostError(RuntimeError):
    pass


class Panic(HostError):
    """Raised where the reference panics (e.g. "No bounding box in BVHNode", bvh.rs:30,58)."""


class Unsupported(HostError):
    """The object graph cannot be lowered to the device (open-ended trait impls, exotic nesting)."""


def _d3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


class _Obj:
    __slots__ = ("h", "keep", "host")

    def __init__(self, host, h, keep=()):
        if not h:
            host._raise()
        self.host = host
        self.h = h
        self.keep = keep


class _List(_Obj):
    def push(self, hittable):
        self.keep = self.keep + (hittable,)
        self.host._check(self.host.lib.rth_list_push(self.h, hittable.h))


def default_params(nx, ny, ns, seed=42, flags=0, max_depth=50, t_min=0.001, tile_rank=0, tile_world=1, spp_chunks=0,
                   shade_threshold=0, sample_buffer_bytes=0, progress=None):
    """progress: callable(done_units, total_units) -> falsy to go on / truthy to cancel; called from the blocking
    render calls about every 50 ms (the object returned keeps the ctypes trampoline alive as `_progress_keep`)."""
    p = abi.RenderParams()
    if progress is not None:
        def _tramp(done, total, _user):
            try:
                return 1 if progress(int(done), int(total)) else 0
            except Exception:  # nothing may unwind through the C ABI
                return 1
        cb = abi.PROGRESS_FN(_tramp)
        p._progress_keep = cb
        p.progress_fn = C.cast(cb, C.c_void_p).value
    p.nx, p.ny, p.ns = nx, ny, ns
    p.max_depth, p.t_min, p.flags, p.seed = max_depth, t_min, flags, seed
    p.tile_rank, p.tile_world, p.spp_chunks = tile_rank, tile_world, spp_chunks
    p.shade_threshold = shade_threshold
    p.sample_buffer_bytes = sample_buffer_bytes
    return p


class Scene:
    """A world lowered to the flat device description (and, after upload(), resident in HBM)."""

    def __init__(self, host, world):
        self.host = host
        self.world = world
        self.h = host.lib.rth_lower(world.h)
        if not self.h:
            host._raise()
        self.uploaded = False

    def desc(self):
        d = abi.SceneDesc()
        self.host._check(self.host.lib.rth_lowered_desc(self.h, C.byref(d)))
        return d

    def arrays(self):
        """numpy views of the flat arrays (for inspection / CPU tests of the lowering)."""
        d = self.desc()

        def view(ptr, n, dtype, cols=None):
            if n == 0:
                return np.zeros((0,) if cols is None else (0, cols), dtype)
            a = np.ctypeslib.as_array(ptr, shape=(n,) if cols is None else (n, cols))
            return a.view(dtype).copy() if cols is None else a.copy()

        out = {
            "items": [d.items[i] for i in range(d.n_items)],
            "prim_a": view(d.prim_a, d.n_prims * 4, np.float32).reshape(-1, 4),
            "prim_b": view(d.prim_b, d.n_prims * 4, np.float32).reshape(-1, 4),
            "prim_meta": [d.prim_meta[i] for i in range(d.n_prims)],
            "nodes": [d.nodes[i] for i in range(d.n_nodes)],
            "xforms": [d.xforms[i] for i in range(d.n_xforms)],
            "materials": [d.materials[i] for i in range(d.n_materials)],
            "textures": [d.textures[i] for i in range(d.n_textures)],
            "n_perlin": d.n_perlin, "n_images": d.n_images, "image_bytes": d.image_bytes,
            "max_bvh_depth": d.max_bvh_depth,
            "prim_gate": view(d.prim_gate, d.n_prims * 8, np.float32).reshape(-1, 8) if d.prim_gate else np.zeros((0, 8), np.float32),
        }
        return out

    def upload(self, device=0):
        self.host._check(self.host.lib.rth_upload(self.h, device))
        self.uploaded = True
        return self

    def render(self, cam, nx, ny, ns, sig=False, out=None, **kw):
        """Blocking whole-image render -> dict(linear f32 [ny,nx,3], rgb8 u8 [ny,nx,3], stats[, sig u64 [ny,nx]]).
        `out` = (linear, rgb8) arrays to reuse."""
        if not self.uploaded:
            self.upload(kw.pop("device", 0))
        kw.pop("device", None)
        p = default_params(nx, ny, ns, **kw)
        lin, rgb = out if out is not None else (np.zeros((ny, nx, 3), np.float32), np.zeros((ny, nx, 3), np.uint8))
        sg = np.zeros((ny, nx), np.uint64) if sig else None
        st = abi.Stats()
        self.host._check(self.host.lib.rth_render(self.h, cam.h, C.byref(p), lin.ctypes.data, rgb.ctypes.data,
                                                   sg.ctypes.data if sig else None, C.byref(st)))
        out = {"linear": lin, "rgb8": rgb, "stats": _stats(st)}
        if sig:
            out["sig"] = sg
        return out

    def render_multi(self, cam, nx, ny, ns, devices, **kw):
        """Whole image on several GPUs of this process (rtmi_render_multi): tiles t % len(devices), one gather.
        A device may be listed more than once (single-GPU rehearsal).  Bit-identical to render()."""
        p = default_params(nx, ny, ns, **kw)
        lin = np.zeros((ny, nx, 3), np.float32)
        rgb = np.zeros((ny, nx, 3), np.uint8)
        st = abi.Stats()
        dev = (C.c_int * len(devices))(*devices)
        self.host._check(self.host.lib.rth_render_multi(self.h, cam.h, C.byref(p), dev, len(devices), lin.ctypes.data,
                                                         rgb.ctypes.data, C.byref(st)))
        return {"linear": lin, "rgb8": rgb, "stats": _stats(st)}

    def partial_image(self, nx, ny, ns):
        """RTMI_FLAG_PROGRESSIVE: the image of the passes finished so far of the render() call running on this scene —
        ONLY from inside that call's progress callback.  Returns (spp_done, linear, rgb8); spp_done == 0: nothing yet."""
        p = default_params(nx, ny, ns)
        lin = np.zeros((ny, nx, 3), np.float32)
        rgb = np.zeros((ny, nx, 3), np.uint8)
        spp = C.c_uint32(0)
        self.host._check(self.host.lib.rth_partial_image(self.h, C.byref(p), lin.ctypes.data, rgb.ctypes.data, C.byref(spp)))
        return int(spp.value), lin, rgb

    def upload_multi(self, devices):
        """Keeps the scene resident on a list of GPUs of this process (rtmi_multi_create): uploads once; every later
        render_resident() costs the kernels, one gather and the un-tiling.  A device may be listed more than once."""
        dev = (C.c_int * len(devices))(*devices)
        self.host._check(self.host.lib.rth_upload_multi(self.h, dev, len(devices)))
        self.multi_devices = list(devices)
        return self

    def multi_collective(self):
        """Which exchange render_resident() performs: "none" (one device), "peer_copy" (a device listed twice) or
        "rccl" (one grouped ncclGather; distinct devices, or one device with RTMI_FORCE_RCCL=1 at upload_multi)."""
        return {0: "none", 1: "peer_copy", 2: "rccl"}.get(self.host.lib.rth_multi_collective(self.h), "no handle")

    def free_multi(self):
        self.host._check(self.host.lib.rth_multi_free(self.h))
        self.multi_devices = None
        return self

    def prepare_resident(self, nx, ny, ns, **kw):
        p = default_params(nx, ny, ns, **kw)
        self.host._check(self.host.lib.rth_multi_prepare(self.h, C.byref(p)))
        return self

    def render_resident(self, cam, nx, ny, ns, out=None, **kw):
        """rtmi_multi_render on the device list of upload_multi().  `out` = (linear, rgb8) arrays to reuse."""
        p = default_params(nx, ny, ns, **kw)
        lin, rgb = out if out is not None else (np.zeros((ny, nx, 3), np.float32), np.zeros((ny, nx, 3), np.uint8))
        st = abi.Stats()
        self.host._check(self.host.lib.rth_multi_render(self.h, cam.h, C.byref(p), lin.ctypes.data, rgb.ctypes.data, C.byref(st)))
        return {"linear": lin, "rgb8": rgb, "stats": _stats(st)}

    def check_status(self):
        """Raises if an asynchronous render_device() call since the last check overflowed its traversal pool."""
        self.host._check(self.host.lib.rth_scene_status(self.h))
        return self

    def local_tiles(self, params):
        return abi.load_rtmi().rtmi_local_tiles(C.byref(params))

    def prepare(self, params):
        """Allocate the render buffers for `params` now (per-sample buffer: 12 B x local pixels x samples per pass)."""
        self.host._check(self.host.lib.rth_render_prepare(self.h, C.byref(params)))
        return self

    def render_device(self, cam, params, d_texels_ptr, stream=None, want_stats=False):
        """Enqueue on `stream`; writes rtmi_local_tiles()*64 texels (16 B) at device address d_texels_ptr."""
        st = abi.Stats() if want_stats else None
        self.host._check(self.host.lib.rth_render_device(self.h, cam.h, C.byref(params), C.c_void_p(d_texels_ptr),
                                                          C.c_void_p(stream or 0), C.byref(st) if st else None))
        return _stats(st) if st else None


def _stats(st):
    return {"kernel_ms": st.kernel_ms, "render_ms": st.render_ms, "samples": int(st.samples), "tiles": st.tiles,
            "chunks": st.chunks, "blocks": st.blocks, "kernel": st.kernel}


class _Camera(_Obj):
    def render(self, world, nx, ny, ns, seed=42, flags=0, device=0):
        """Camera::render(world, nx, ny, ns): lower + upload + render + free, like one create_image call."""
        lin = np.zeros((ny, nx, 3), np.float32)
        rgb = np.zeros((ny, nx, 3), np.uint8)
        st = abi.Stats()
        self.host._check(self.host.lib.rth_camera_render(self.h, world.h, nx, ny, ns, seed, flags, device,
                                                          lin.ctypes.data, rgb.ctypes.data, C.byref(st)))
        return {"linear": lin, "rgb8": rgb, "stats": _stats(st)}

    def get_ray(self, s, t, seed=0):
        out = np.zeros(7)
        self.host._check(self.host.lib.rth_get_ray(self.h, s, t, seed, out.ctypes.data))
        return out

    def state(self):
        out = np.zeros(21)
        self.host._check(self.host.lib.rth_camera_state(self.h, out.ctypes.data))
        return out

    def lower(self):
        c = abi.Camera()
        self.host._check(self.host.lib.rth_camera_lower(self.h, C.byref(c)))
        return c


class Host:
    PLANE_YZ, PLANE_ZX, PLANE_XY = PLANE_YZ, PLANE_ZX, PLANE_XY
    AXIS_X, AXIS_Y, AXIS_Z = AXIS_X, AXIS_Y, AXIS_Z
    precision = "host"

    def __init__(self):
        self.lib = abi.load_host()

    def _raise(self):
        msg = (self.lib.rth_last_error() or b"").decode()
        raise {2: Panic, 3: Unsupported}.get(self.lib.rth_last_error_code(), HostError)(msg)

    def _check(self, rc):
        if rc == 0:
            return
        msg = (self.lib.rth_last_error() or b"").decode()
        raise {2: Panic, 3: Unsupported}.get(rc, HostError)(msg)

    def seed_scene_rng(self, seed):
        self.lib.rth_seed_scene_rng(int(seed))

    def free_all(self):
        self.lib.rth_free_all()

    # ---- textures (src/texture.rs) ----
    def SolidTexture(self, r, g, b):
        return _Obj(self, self.lib.rth_tex_solid(r, g, b))

    def CheckerTexture(self, odd, even):
        return _Obj(self, self.lib.rth_tex_checker(odd.h, even.h), (odd, even))

    def NoiseTexture(self, scale):
        return _Obj(self, self.lib.rth_tex_noise(scale))

    def ImageTexture(self, data, nx, ny):
        arr = np.ascontiguousarray(np.asarray(data, dtype=np.uint8).reshape(-1))
        if arr.size != nx * ny * 3:
            raise Panic("ImageTexture: data size != 3*nx*ny")
        return _Obj(self, self.lib.rth_tex_image(arr.ctypes.data, nx, ny))

    # ---- materials (src/material.rs) ----
    def Lambertian(self, tex):
        return _Obj(self, self.lib.rth_mat_lambertian(tex.h), (tex,))

    def Metal(self, tex, fuzz):
        return _Obj(self, self.lib.rth_mat_metal(tex.h, fuzz), (tex,))

    def Dielectric(self, ref_idx):
        return _Obj(self, self.lib.rth_mat_dielectric(ref_idx))

    def DiffuseLight(self, tex):
        return _Obj(self, self.lib.rth_mat_diffuse_light(tex.h), (tex,))

    def Isotropic(self, tex):
        return _Obj(self, self.lib.rth_mat_isotropic(tex.h), (tex,))

    # ---- hittables ----
    def Sphere(self, center, radius, material):
        c = _d3(center)
        return _Obj(self, self.lib.rth_sphere(c[0], c[1], c[2], radius, material.h), (material,))

    def MovingSphere(self, center0, center1, time0, time1, radius, material):
        a, b = _d3(center0), _d3(center1)
        return _Obj(self, self.lib.rth_moving_sphere(a[0], a[1], a[2], b[0], b[1], b[2], time0, time1, radius,
                                                     material.h), (material,))

    def Rect(self, plane, x0, y0, x1, y1, k, material):
        return _Obj(self, self.lib.rth_rect(plane, x0, y0, x1, y1, k, material.h), (material,))

    def Cube(self, p_min, p_max, material):
        a, b = _d3(p_min), _d3(p_max)
        return _Obj(self, self.lib.rth_cube(a[0], a[1], a[2], b[0], b[1], b[2], material.h), (material,))

    def FlipNormals(self, hittable):
        return _Obj(self, self.lib.rth_flip_normals(hittable.h), (hittable,))

    def Traslate(self, hittable, offset):
        o = _d3(offset)
        return _Obj(self, self.lib.rth_translate(hittable.h, o[0], o[1], o[2]), (hittable,))

    def Rotate(self, axis, hittable, angle):
        return _Obj(self, self.lib.rth_rotate(axis, hittable.h, angle), (hittable,))

    def ConstantMedium(self, boundary, density, texture):
        return _Obj(self, self.lib.rth_constant_medium(boundary.h, density, texture.h), (boundary, texture))

    def HittableList(self):
        return _List(self, self.lib.rth_list_new())

    def BVHNode(self, hittables, time0, time1):
        arr = (C.c_void_p * len(hittables))(*[h.h for h in hittables])
        return _Obj(self, self.lib.rth_bvh(arr, len(hittables), time0, time1), tuple(hittables))

    def Camera(self, look_from, look_at, view_up, vertical_fov, aspect, aperture, focus_dist, time0, time1):
        f, a, u = _d3(look_from), _d3(look_at), _d3(view_up)
        return _Camera(self, self.lib.rth_camera(f[0], f[1], f[2], a[0], a[1], a[2], u[0], u[1], u[2], vertical_fov,
                                                 aspect, aperture, focus_dist, time0, time1))

    # ---- lowering / rendering ----
    def lower(self, world):
        return Scene(self, world)

    def create_image(self, ny, nx, ns, cam, world, seed=42, flags=0, device=0):
        """tests/test.rs:55-85: returns the P3 text (bytes).  Argument order (ny, nx, ns, cam, world)."""
        img = cam.render(world, nx, ny, ns, seed=seed, flags=flags, device=device)
        return ppm_p3(img["rgb8"])

    # ---- CPU evaluation of the mirror (f64) ----
    def hit(self, hittable, origin, direction, time=0.0, t_min=0.001, t_max=float("inf"), seed=0):
        o, d = _d3(origin), _d3(direction)
        out = np.zeros(9)
        found = C.c_int(0)
        tmx = 1.8e308 if t_max == float("inf") else t_max
        tmn = -1.8e308 if t_min == -float("inf") else t_min
        self._check(self.lib.rth_hit(hittable.h, o.ctypes.data, d.ctypes.data, time, tmn, tmx, seed, out.ctypes.data,
                                     C.byref(found)))
        if not found.value:
            return None
        return {"t": out[0], "u": out[1], "v": out[2], "p": out[3:6].copy(), "normal": out[6:9].copy()}

    def bounding_box(self, hittable, t0=0.0, t1=1.0):
        out = np.zeros(6)
        found = C.c_int(0)
        self._check(self.lib.rth_bounding_box(hittable.h, t0, t1, out.ctypes.data, C.byref(found)))
        if not found.value:
            return None
        return out[:3].copy(), out[3:].copy()

    def tex_value(self, tex, u, v, p):
        pp = _d3(p)
        out = np.zeros(3)
        self._check(self.lib.rth_tex_value(tex.h, u, v, pp.ctypes.data, out.ctypes.data))
        return out

    def scatter(self, mat, ray_o, ray_d, time, rec, seed=0):
        o, d = _d3(ray_o), _d3(ray_d)
        r9 = np.ascontiguousarray(np.concatenate([[rec["t"], rec["u"], rec["v"]], rec["p"], rec["normal"]]),
                                  dtype=np.float64)
        out = np.zeros(10)
        sc = C.c_int(0)
        self._check(self.lib.rth_scatter(mat.h, o.ctypes.data, d.ctypes.data, time, r9.ctypes.data, seed,
                                         out.ctypes.data, C.byref(sc)))
        if not sc.value:
            return None
        return {"o": out[0:3].copy(), "d": out[3:6].copy(), "time": out[6], "attenuation": out[7:10].copy()}

    def emitted(self, mat, u, v, p):
        pp = _d3(p)
        out = np.zeros(3)
        self._check(self.lib.rth_emitted(mat.h, u, v, pp.ctypes.data, out.ctypes.data))
        return out

    def color_sample(self, cam, world, nx, ny, i, j, s, seed=42, sky=False, face_forward=False, uv_book=False):
        """color() of one camera sample on the CPU mirror.  Opt-in extensions (off by default): sky = background of
        color.rs:18-20; face_forward = opaque materials see the normal turned against the ray; uv_book = pi/2 in
        get_sphere_uv instead of FRAC_2_PI (sphere.rs:13)."""
        out = np.zeros(3)
        self.lib.rth_set_sky_background(1 if sky else 0)
        self.lib.rth_set_face_forward(1 if face_forward else 0)
        self.lib.rth_set_uv_book(1 if uv_book else 0)
        try:
            self._check(self.lib.rth_color_sample(cam.h, world.h, nx, ny, i, j, s, seed, out.ctypes.data))
        finally:
            self.lib.rth_set_sky_background(0)
            self.lib.rth_set_face_forward(0)
            self.lib.rth_set_uv_book(0)
        return out

    def perlin_tables(self, tex):
        rv = np.zeros(768)
        pm = np.zeros(768, np.int32)
        self._check(self.lib.rth_perlin_tables(tex.h, rv.ctypes.data, pm.ctypes.data))
        return rv.reshape(256, 3), pm.reshape(3, 256)


def release_cached():
    """Returns the per-sample buffers that destroyed handles left parked for their successors (rtmi_release_cached)."""
    abi.load_rtmi().rtmi_release_cached()


def ppm_p3(rgb8):
    """The P3 text of create_image (tests/test.rs:59,79) through the native formatter."""
    lib = abi.load_rtmi()
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    ny, nx = rgb8.shape[:2]
    need = lib.rtmi_ppm_p3(nx, ny, rgb8.ctypes.data, None, 0)
    buf = C.create_string_buffer(need)
    n = lib.rtmi_ppm_p3(nx, ny, rgb8.ctypes.data, buf, need)
    return buf.raw[:n]


def write_ppm(path, rgb8, fmt=3):
    """Stream the image to `path` without building the whole-image string: fmt 3 = the P3 text of create_image
    (byte-identical to ppm_p3), fmt 6 = binary P6."""
    lib = abi.load_rtmi()
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    ny, nx = rgb8.shape[:2]
    rc = lib.rtmi_write_ppm(os.fsencode(path), nx, ny, rgb8.ctypes.data, int(fmt))
    if rc != 0:
        raise HostError("rtmi_write_ppm failed (%d): %s" % (rc, lib.rtmi_last_error().decode()))
