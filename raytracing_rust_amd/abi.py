"""ctypes mirrors of the plain-old-data structs of include/rtmi.h and the library loader.

The HIP extension is mandatory: there is no CPU fallback for the render path.  If the
native libraries are missing they are built in-tree (hipcc cross-compiles without a GPU);
if that fails the import fails loudly.
"""
import ctypes as C
import os

from . import build as _build

RTMI_ABI_VERSION = 7
RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT = 8  # MEDIUM items: how many of the first transforms wrap the medium itself (bits 8..11)
RTMI_PRIMFLAG_XF_COUNT_SHIFT = 4   # instanced primitive: number of its own transforms (bits 4..7)
RTMI_PRIMFLAG_XF_FIRST_SHIFT = 12  # ... and the index of the first one in xforms (bits 12..31)
RTMI_MAX_BVH_DEPTH = 24
RTMI_TILE = 8
RTMI_FLAG_FAST_CULL = 1
RTMI_FLAG_PATH_SIG = 2
RTMI_FLAG_PROFILE = 4
RTMI_FLAG_SYNC = 8
RTMI_FLAG_ASYNC = 16
RTMI_FLAG_SKY = 32
RTMI_FLAG_REF_TREE = 64
RTMI_FLAG_BLOCK_COOP = 32768
RTMI_SAMPLE_SLOT_BYTES = 12  # per-sample radiance buffer: three fp32 per finished path
RTMI_COLLECTIVE_NONE, RTMI_COLLECTIVE_PEER_COPY, RTMI_COLLECTIVE_RCCL = 0, 1, 2
RTMI_KERNEL_PERLANE, RTMI_KERNEL_WAVE_COOP, RTMI_KERNEL_ASYNC, RTMI_KERNEL_BLOCK_COOP = 0, 1, 2, 3
RTMI_FLAG_FACE_FORWARD = 128
RTMI_FLAG_UV_BOOK = 4096
RTMI_FLAG_TEST_OVERFLOW = 8192
RTMI_FLAG_PROGRESSIVE = 16384  # opt-in: the framebuffer holds the image of the samples so far after every pass
RTMI_ERR_DEVICE = 3
RTMI_ERR_CANCELLED = 5
RTMI_TEXEL_POISON = 0x80000000

TEX_SOLID, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = 0, 1, 2, 3
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = 0, 1, 2, 3, 4
PRIM_SPHERE, PRIM_MSPHERE, PRIM_RECT, PRIM_CUBE = 0, 1, 2, 3
ITEM_LIST, ITEM_BVH = 0, 1
ITEMFLAG_FLIP, ITEMFLAG_MEDIUM, ITEMFLAG_SAVE_T0, ITEMFLAG_DEFERRED, ITEMFLAG_NESTED_MEDIUM = 1, 2, 4, 8, 16
ITEMFLAG_LISTSCAN_BEGIN, ITEMFLAG_LISTSCAN_MEMBER, ITEMFLAG_LISTSCAN_END = 32, 64, 128
RTMI_ITEMFLAG_GATE_OUTER_SHIFT = 12  # DEFERRED items: how many leading transforms belong to the enclosing BVH item (bits 12..15)
XF_TRANSLATE, XF_ROTATE_X, XF_ROTATE_Y, XF_ROTATE_Z, XF_GATE_MIN, XF_GATE_MAX, XF_INNER_MEDIUM = 0, 1, 2, 3, 4, 5, 6


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("pad", C.c_int32),
                ("f0", C.c_float), ("f1", C.c_float), ("f2", C.c_float), ("f3", C.c_float)]


class Perlin(C.Structure):
    _fields_ = [("ranvec", C.c_float * 1024), ("perm", C.c_int32 * 768)]


class ImageDesc(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("nx", C.c_uint32), ("ny", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tex", C.c_int32), ("param", C.c_float), ("flags", C.c_uint32)]


class PrimMeta(C.Structure):
    _fields_ = [("material", C.c_int32), ("flags", C.c_uint32), ("inv_dt", C.c_float), ("type", C.c_int32)]


class BvhNode(C.Structure):
    _fields_ = [("lmin", C.c_float * 3), ("lmax", C.c_float * 3), ("rmin", C.c_float * 3), ("rmax", C.c_float * 3),
                ("left", C.c_int32), ("right", C.c_int32), ("pad", C.c_int32 * 2)]


class Bvh4Node(C.Structure):
    _fields_ = [("minx", C.c_float * 4), ("miny", C.c_float * 4), ("minz", C.c_float * 4), ("maxx", C.c_float * 4),
                ("maxy", C.c_float * 4), ("maxz", C.c_float * 4), ("child", C.c_int32 * 4), ("pad", C.c_int32 * 4)]


class Xform(C.Structure):
    _fields_ = [("kind", C.c_int32), ("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Item(C.Structure):
    _fields_ = [("kind", C.c_int32), ("first", C.c_int32), ("count", C.c_int32), ("flags", C.c_uint32),
                ("xform_first", C.c_int32), ("xform_count", C.c_int32), ("medium_material", C.c_int32),
                ("neg_inv_density", C.c_float), ("root_min", C.c_float * 3), ("root_max", C.c_float * 3),
                ("scale", C.c_float), ("alt_first", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_items", C.c_uint32), ("items", C.POINTER(Item)),
                ("n_prims", C.c_uint32), ("prim_a", C.POINTER(C.c_float)), ("prim_b", C.POINTER(C.c_float)),
                ("prim_meta", C.POINTER(PrimMeta)), ("prim_gate", C.POINTER(C.c_float)), ("alt_max_depth", C.c_uint32),
                ("n_alt_nodes", C.c_uint32), ("alt_nodes", C.POINTER(Bvh4Node)),
                ("n_nodes", C.c_uint32), ("nodes", C.POINTER(BvhNode)),
                ("n_xforms", C.c_uint32), ("xforms", C.POINTER(Xform)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
                ("n_perlin", C.c_uint32), ("perlin", C.POINTER(Perlin)),
                ("n_images", C.c_uint32), ("images", C.POINTER(ImageDesc)),
                ("image_data", C.POINTER(C.c_uint8)), ("image_bytes", C.c_uint64),
                ("max_bvh_depth", C.c_uint32), ("bvh_time_lo", C.c_float), ("bvh_time_hi", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("time0", C.c_float), ("time1", C.c_float), ("lens_radius", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("nx", C.c_uint32), ("ny", C.c_uint32), ("ns", C.c_uint32), ("max_depth", C.c_uint32),
                ("t_min", C.c_float), ("flags", C.c_uint32), ("seed", C.c_uint64),
                ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32), ("spp_chunks", C.c_uint32), ("shade_threshold", C.c_uint32),
                ("path_sig", C.c_uint64), ("prof", C.c_uint64), ("sample_buffer_bytes", C.c_uint64),
                ("progress_fn", C.c_uint64), ("progress_user", C.c_uint64)]


# rtmi_progress_fn: int (*)(uint64_t done, uint64_t total, void *user)
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint64, C.c_void_p)


class Texel(C.Structure):
    _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float), ("rgb8", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("render_ms", C.c_double), ("samples", C.c_uint64),
                ("tiles", C.c_uint32), ("chunks", C.c_uint32), ("blocks", C.c_uint32), ("kernel", C.c_uint32)]


# every entry point include/rtmi.h declares (tests check that the library exports them all)
RTMI_SYMBOLS = ["rtmi_device_count", "rtmi_last_error", "rtmi_build_hash", "rtmi_scene_create", "rtmi_scene_destroy", "rtmi_release_cached", "rtmi_local_tiles",
                "rtmi_render_prepare", "rtmi_render_device", "rtmi_scene_status", "rtmi_render", "rtmi_render_multi", "rtmi_multi_create",
                "rtmi_multi_prepare", "rtmi_multi_render", "rtmi_multi_destroy", "rtmi_multi_collective", "rtmi_partial_image", "rtmi_untile",
                "rtmi_ppm_p3", "rtmi_write_ppm", "rtmi_probe_math", "rtmi_probe_philox", "rtmi_probe_xform"]

_rtmi = None
_host = None


def load_rtmi():
    """librtmi.so: the C ABI of include/rtmi.h (HIP kernels inside)."""
    global _rtmi
    if _rtmi is not None:
        return _rtmi
    path = _build.LIBRTMI
    if not os.path.exists(path):
        _build.build_rtmi()
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.rtmi_device_count.restype = C.c_int
    lib.rtmi_last_error.restype = C.c_char_p
    lib.rtmi_build_hash.restype = C.c_char_p
    lib.rtmi_scene_create.restype = C.c_int
    lib.rtmi_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(vp)]
    lib.rtmi_scene_destroy.restype = None
    lib.rtmi_scene_destroy.argtypes = [vp]
    lib.rtmi_release_cached.restype = None
    lib.rtmi_release_cached.argtypes = []
    lib.rtmi_local_tiles.restype = C.c_uint32
    lib.rtmi_local_tiles.argtypes = [C.POINTER(RenderParams)]
    lib.rtmi_render_device.restype = C.c_int
    lib.rtmi_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    lib.rtmi_scene_status.restype = C.c_int
    lib.rtmi_scene_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.rtmi_render_prepare.restype = C.c_int
    lib.rtmi_render_prepare.argtypes = [vp, C.POINTER(RenderParams)]
    lib.rtmi_render_multi.restype = C.c_int
    lib.rtmi_render_multi.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_uint32, C.POINTER(Camera), C.POINTER(RenderParams),
                                      vp, vp, C.POINTER(Stats)]
    lib.rtmi_multi_create.restype = C.c_int
    lib.rtmi_multi_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp)]
    lib.rtmi_multi_prepare.restype = C.c_int
    lib.rtmi_multi_prepare.argtypes = [vp, C.POINTER(RenderParams)]
    lib.rtmi_multi_render.restype = C.c_int
    lib.rtmi_multi_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    lib.rtmi_multi_destroy.restype = None
    lib.rtmi_multi_destroy.argtypes = [vp]
    lib.rtmi_multi_collective.restype = C.c_int
    lib.rtmi_multi_collective.argtypes = [vp]
    lib.rtmi_render.restype = C.c_int
    lib.rtmi_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderParams), vp, vp, vp, C.POINTER(Stats)]
    lib.rtmi_partial_image.restype = C.c_int
    lib.rtmi_partial_image.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(C.c_uint32)]
    lib.rtmi_untile.restype = C.c_int
    lib.rtmi_untile.argtypes = [C.POINTER(RenderParams), vp, vp, vp]
    lib.rtmi_ppm_p3.restype = C.c_size_t
    lib.rtmi_ppm_p3.argtypes = [C.c_uint32, C.c_uint32, vp, vp, C.c_size_t]
    lib.rtmi_write_ppm.restype = C.c_int
    lib.rtmi_write_ppm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp, C.c_int]
    lib.rtmi_probe_math.restype = C.c_int
    lib.rtmi_probe_math.argtypes = [C.c_int, vp, vp, vp, C.c_uint32]
    lib.rtmi_probe_philox.restype = C.c_int
    lib.rtmi_probe_philox.argtypes = [vp, vp, vp, C.c_uint32]
    lib.rtmi_probe_xform.restype = C.c_int
    lib.rtmi_probe_xform.argtypes = [C.POINTER(Xform), C.c_uint32, vp, vp, vp, C.c_uint32]
    _rtmi = lib
    return lib


def load_host():
    """librt_host.so: the C++ mirror of the reference's trait surface (links librtmi.so)."""
    global _host
    if _host is not None:
        return _host
    load_rtmi()
    path = _build.LIBHOST
    if not os.path.exists(path):
        _build.build_host()
    lib = C.CDLL(path)
    vp, d, i, u32, u64 = C.c_void_p, C.c_double, C.c_int, C.c_uint32, C.c_uint64
    sig = {
        "rth_last_error": (C.c_char_p, []),
        "rth_last_error_code": (i, []),
        "rth_free_all": (None, []),
        "rth_seed_scene_rng": (None, [u64]),
        "rth_scene_uniform": (d, []),
        "rth_philox": (None, [vp, vp, vp]),
        "rth_tex_solid": (vp, [d, d, d]),
        "rth_tex_checker": (vp, [vp, vp]),
        "rth_tex_noise": (vp, [d]),
        "rth_tex_image": (vp, [vp, u32, u32]),
        "rth_perlin_tables": (i, [vp, vp, vp]),
        "rth_mat_lambertian": (vp, [vp]),
        "rth_mat_metal": (vp, [vp, d]),
        "rth_mat_dielectric": (vp, [d]),
        "rth_mat_diffuse_light": (vp, [vp]),
        "rth_mat_isotropic": (vp, [vp]),
        "rth_sphere": (vp, [d, d, d, d, vp]),
        "rth_moving_sphere": (vp, [d] * 9 + [vp]),
        "rth_rect": (vp, [i, d, d, d, d, d, vp]),
        "rth_cube": (vp, [d] * 6 + [vp]),
        "rth_flip_normals": (vp, [vp]),
        "rth_translate": (vp, [vp, d, d, d]),
        "rth_rotate": (vp, [i, vp, d]),
        "rth_constant_medium": (vp, [vp, d, vp]),
        "rth_list_new": (vp, []),
        "rth_list_push": (i, [vp, vp]),
        "rth_bvh": (vp, [vp, i, d, d]),
        "rth_camera": (vp, [d] * 15),
        "rth_camera_lower": (i, [vp, C.POINTER(Camera)]),
        "rth_camera_state": (i, [vp, vp]),
        "rth_lower": (vp, [vp]),
        "rth_lowered_desc": (i, [vp, C.POINTER(SceneDesc)]),
        "rth_upload": (i, [vp, i]),
        "rth_render": (i, [vp, vp, C.POINTER(RenderParams), vp, vp, vp, C.POINTER(Stats)]),
        "rth_render_device": (i, [vp, vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]),
        "rth_render_prepare": (i, [vp, C.POINTER(RenderParams)]),
        "rth_scene_status": (i, [vp]),
        "rth_render_multi": (i, [vp, vp, C.POINTER(RenderParams), C.POINTER(C.c_int), u32, vp, vp, C.POINTER(Stats)]),
        "rth_partial_image": (i, [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(C.c_uint32)]),
        "rth_upload_multi": (i, [vp, C.POINTER(C.c_int), u32]),
        "rth_multi_free": (i, [vp]),
        "rth_multi_collective": (i, [vp]),
        "rth_multi_prepare": (i, [vp, C.POINTER(RenderParams)]),
        "rth_multi_render": (i, [vp, vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]),
        "rth_camera_render": (i, [vp, vp, u32, u32, u32, u64, u32, i, vp, vp, C.POINTER(Stats)]),
        "rth_hit": (i, [vp, vp, vp, d, d, d, u64, vp, vp]),
        "rth_bounding_box": (i, [vp, d, d, vp, vp]),
        "rth_tex_value": (i, [vp, d, d, vp, vp]),
        "rth_scatter": (i, [vp, vp, vp, d, vp, u64, vp, vp]),
        "rth_emitted": (i, [vp, d, d, vp, vp]),
        "rth_get_ray": (i, [vp, d, d, u64, vp]),
        "rth_set_sky_background": (None, [i]),
        "rth_set_face_forward": (None, [i]),
        "rth_set_uv_book": (None, [i]),
        "rth_color_sample": (i, [vp, vp, u32, u32, u32, u32, u32, u64, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _host = lib
    return lib
