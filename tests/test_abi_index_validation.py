"""Every index a scene description carries is checked before anything reaches the device (rtmi_scene_create's validate):
a table of out-of-range values for each index-bearing field of final_scene's description — items (kind, first, count,
transform range, medium material, alternative root), primitives (material, type, own transform chain), materials
(kind, texture), textures (kind, children, Perlin / image index), transforms (kind), images (offset, size) — each of which
must come back as RTMI_ERR_INVALID with a message, never as a crash or as a scene.  Runs without a GPU: a description that
passes validation ends in RTMI_ERR_DEVICE here.  (The reference has no such boundary: its scene is a graph of Rust objects.)"""
import ctypes as C

from raytracing_rust_amd import abi, scenes

BIG = 0x7FFFFF00


def _private(d):
    """Replace the description's index-bearing arrays by private copies a test may corrupt."""
    out = {}

    def dup(name, field, n, T):
        arr = (T * max(n, 1))()
        if n:
            C.memmove(arr, getattr(d, field), C.sizeof(T) * n)
        setattr(d, field, C.cast(arr, C.POINTER(T)))
        out[name] = arr

    dup("items", "items", d.n_items, abi.Item)
    dup("meta", "prim_meta", d.n_prims, abi.PrimMeta)
    dup("mats", "materials", d.n_materials, abi.Material)
    dup("texs", "textures", d.n_textures, abi.Texture)
    dup("xf", "xforms", d.n_xforms, abi.Xform)
    dup("img", "images", d.n_images, abi.ImageDesc)
    return out


def _rc(d):
    lib = abi.load_rtmi()
    h = C.c_void_p()
    rc = lib.rtmi_scene_create(C.byref(d), 0, C.byref(h))
    msg = (lib.rtmi_last_error() or b"").decode()
    if h.value:
        lib.rtmi_scene_destroy(h)
    return rc, msg


def test_every_out_of_range_index_is_rejected(host):
    cam, world = scenes.build(host, "final_scene", 16, 16, seed=1)
    sc = host.lower(world)
    d = sc.desc()
    a = _private(d)
    assert _rc(d)[0] in (0, 3)  # the untouched description is valid
    accepted = []

    def attempt(label, obj, field, values):
        for v in values:
            old = getattr(obj, field)
            setattr(obj, field, v)
            rc, msg = _rc(d)
            setattr(obj, field, old)
            if rc != 1 or not msg:
                accepted.append((label, v, rc))

    for i in range(d.n_items):
        it = a["items"][i]
        attempt("item%d.kind" % i, it, "kind", [2, -1, 77])
        attempt("item%d.first" % i, it, "first", [-1, BIG, d.n_prims + d.n_nodes + 5])
        if it.kind == abi.ITEM_LIST:
            attempt("item%d.count" % i, it, "count", [-1, BIG, d.n_prims + 1])  # (0 = an empty list: legal, never hit)
        if it.xform_count:
            attempt("item%d.xform_first" % i, it, "xform_first", [-1, BIG])
        attempt("item%d.xform_count" % i, it, "xform_count", [-1, BIG, d.n_xforms + 1])
        if it.flags & abi.ITEMFLAG_MEDIUM:
            attempt("item%d.medium_material" % i, it, "medium_material", [-1, d.n_materials, BIG])
            attempt("item%d.flags(medium outer)" % i, it, "flags", [it.flags | ((it.xform_count + 1) << abi.RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT)])
        if it.kind == abi.ITEM_BVH:
            attempt("item%d.alt_first" % i, it, "alt_first", [d.n_alt_nodes, BIG])  # (any negative value = no alternative tree)
    for i in (0, 400, d.n_prims - 1):
        m = a["meta"][i]
        attempt("prim%d.material" % i, m, "material", [-1, d.n_materials, BIG])
        attempt("prim%d.type" % i, m, "type", [-1, 4, 99])
        attempt("prim%d.flags(own chain)" % i, m, "flags",
                [m.flags | (3 << abi.RTMI_PRIMFLAG_XF_COUNT_SHIFT) | (BIG & 0xFFFFF000),
                 m.flags | (1 << abi.RTMI_PRIMFLAG_XF_COUNT_SHIFT) | (d.n_xforms << abi.RTMI_PRIMFLAG_XF_FIRST_SHIFT)])
    for i in range(d.n_materials):
        attempt("material%d.kind" % i, a["mats"][i], "kind", [-1, 5, 99])
        if a["mats"][i].kind != abi.MAT_DIELECTRIC:  # (a Dielectric has no texture: the field is not read)
            attempt("material%d.tex" % i, a["mats"][i], "tex", [-1, d.n_textures, BIG])
    kinds = set()
    for i in range(d.n_textures):
        t = a["texs"][i]
        kinds.add(t.kind)
        attempt("texture%d.kind" % i, t, "kind", [-1, 4, 99])
        if t.kind == abi.TEX_CHECKER:
            attempt("texture%d.i0" % i, t, "i0", [-1, d.n_textures, BIG, i])  # i: a checker naming itself never ends
            attempt("texture%d.i1" % i, t, "i1", [-1, d.n_textures, BIG, i])
        if t.kind == abi.TEX_NOISE:
            attempt("texture%d.i0 (perlin)" % i, t, "i0", [-1, d.n_perlin, BIG])
        if t.kind == abi.TEX_IMAGE:
            attempt("texture%d.i0 (image)" % i, t, "i0", [-1, d.n_images, BIG])
    assert {abi.TEX_NOISE, abi.TEX_IMAGE} <= kinds
    for i in range(d.n_xforms):
        attempt("xform%d.kind" % i, a["xf"][i], "kind", [-1, 4, 99])
    for i in range(d.n_images):
        im = a["img"][i]
        attempt("image%d.offset" % i, im, "offset", [d.image_bytes, d.image_bytes - 5, 1 << 60])
        attempt("image%d.nx" % i, im, "nx", [0, im.nx * 2, 0xFFFFFFFF])
        attempt("image%d.ny" % i, im, "ny", [0, im.ny * 2, 0xFFFFFFFF])
    attempt("desc.n_items", d, "n_items", [0])
    assert not accepted, accepted


def test_checker_children_are_validated_on_a_scene_that_has_them(host):
    cam, world = scenes.build(host, "two_spheres", 16, 16)
    sc = host.lower(world)
    d = sc.desc()
    a = _private(d)
    idx = [i for i in range(d.n_textures) if a["texs"][i].kind == abi.TEX_CHECKER]
    assert idx
    for i in idx:
        for field in ("i0", "i1"):
            for v in (-1, d.n_textures, BIG, i):
                old = getattr(a["texs"][i], field)
                setattr(a["texs"][i], field, v)
                rc, msg = _rc(d)
                setattr(a["texs"][i], field, old)
                assert rc == 1 and msg, (i, field, v, rc)
    assert _rc(d)[0] in (0, 3)


def test_checker_nesting_depth_is_bounded_loudly(host):
    """CheckerTexture<T, U> nests arbitrarily in the reference (texture.rs:28-48); the device follows 16 levels.  A
    deeper nest is refused by rtmi_scene_create instead of being shaded wrongly."""
    def world_with(levels):
        t = host.SolidTexture(0.9, 0.9, 0.9)
        for k in range(levels):
            t = host.CheckerTexture(host.SolidTexture(0.1 * (k % 7), 0.3, 0.1), t)
        w = host.HittableList()
        w.push(host.Sphere((0.0, 0.0, 0.0), 1.0, host.Lambertian(t)))
        return w

    assert _rc(host.lower(world_with(16)).desc())[0] in (0, 3)
    rc, msg = _rc(host.lower(world_with(17)).desc())
    assert rc == 1 and "nested deeper" in msg


def test_group_flags_of_deferred_items_are_validated(host):
    """ABI 7's item flags for children of a BVHNode that are not primitives (rtmi.h): a list scan is BEGIN on its first member,
    MEMBER on every item up to the terminator and on no other, one END terminator (kind LIST, no primitives); a NESTED_MEDIUM
    item is a MEDIUM with its RTMI_XF_INNER_MEDIUM record behind its chain; all of them are DEFERRED where the header says so."""
    import sys
    import os

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import dump_flat_scene

    sc = host.lower(dump_flat_scene.media_in_bvh(host, 1))
    d = sc.desc()
    a = _private(d)
    assert _rc(d)[0] in (0, 3)
    items = a["items"]
    n = d.n_items
    B, M, E = abi.ITEMFLAG_LISTSCAN_BEGIN, abi.ITEMFLAG_LISTSCAN_MEMBER, abi.ITEMFLAG_LISTSCAN_END
    begin = next(k for k in range(n) if items[k].flags & B)
    end = next(k for k in range(n) if items[k].flags & E)
    nested = next(k for k in range(n) if (items[k].flags & abi.ITEMFLAG_NESTED_MEDIUM) and not (items[k].flags & abi.ITEMFLAG_DEFERRED))
    assert begin < end and all(items[k].flags & M for k in range(begin, end))

    def broken(k, field, value):
        old = getattr(items[k], field)
        setattr(items[k], field, value)
        rc, msg = _rc(d)
        setattr(items[k], field, old)
        return rc == 1 and bool(msg)

    assert broken(begin, "flags", items[begin].flags & ~B)                       # members without a BEGIN
    assert broken(begin + 1, "flags", items[begin + 1].flags & ~M)               # a non-member inside the group
    assert broken(begin + 1, "flags", items[begin + 1].flags | B)                # a second BEGIN inside
    assert broken(end, "flags", items[end].flags & ~E)                           # no terminator (and a LIST item of no primitives that is DEFERRED)
    assert broken(end, "count", 1)                                               # a terminator with primitives
    assert broken(end, "flags", items[end].flags & ~abi.ITEMFLAG_DEFERRED)       # LISTSCAN without DEFERRED
    assert broken(begin - 1, "flags", items[begin - 1].flags | M) or begin == 0  # a member outside any group
    assert broken(nested, "flags", items[nested].flags & ~abi.ITEMFLAG_MEDIUM)   # NESTED_MEDIUM on a non-medium
    xf = a["xf"]
    at = items[nested].xform_first + items[nested].xform_count
    old = xf[at].kind
    xf[at].kind = abi.XF_TRANSLATE                                               # its inner-medium record replaced
    rc, msg = _rc(d)
    xf[at].kind = old
    assert rc == 1 and msg
    assert _rc(d)[0] in (0, 3)
