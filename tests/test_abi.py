"""The C-ABI library loads on a CPU-only box and exports every symbol include/rtmi.h declares;
struct layouts match the header; the host-side helpers (untile, P3 writer) work without a GPU;
entry points that need a device fail loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracing_rust_amd import abi, default_params, dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = abi.load_rtmi()
    names = _declared_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), "librtmi.so does not export " + n
    assert sorted(abi.RTMI_SYMBOLS) == names


def test_struct_sizes_match_header_comments():
    assert C.sizeof(abi.Texture) == 32
    assert C.sizeof(abi.Material) == 16
    assert C.sizeof(abi.PrimMeta) == 16
    assert C.sizeof(abi.BvhNode) == 64
    assert C.sizeof(abi.Xform) == 16
    assert C.sizeof(abi.Item) == 64
    assert C.sizeof(abi.Texel) == 16
    assert C.sizeof(abi.Perlin) == 256 * 4 * 4 + 3 * 256 * 4
    assert C.sizeof(abi.Camera) == 21 * 4


def test_local_tiles_and_untile_roundtrip():
    lib = abi.load_rtmi()
    nx, ny, world = 37, 21, 3  # ragged: 5 x 3 tiles, partial on both edges
    tiles = ((nx + 7) // 8) * ((ny + 7) // 8)
    counts = [lib.rtmi_local_tiles(C.byref(default_params(nx, ny, 1, tile_rank=r, tile_world=world))) for r in range(world)]
    assert sum(counts) == tiles and counts[0] == max(counts)
    p0 = default_params(nx, ny, 1, tile_rank=0, tile_world=world)
    g = np.zeros((world, counts[0] * 64, 4), np.float32)
    txn = (nx + 7) // 8
    for t in range(tiles):
        r, lt = t % world, t // world
        ty, tx = divmod(t, txn)
        for ly in range(8):
            for lx in range(8):
                row, px = ty * 8 + ly, tx * 8 + lx
                g[r, lt * 64 + ly * 8 + lx, :3] = (row, px, t)
                g[r, lt * 64 + ly * 8 + lx, 3] = np.array([(row & 255) | ((px & 255) << 8) | ((t & 255) << 16)], np.uint32).view(np.float32)[0]
    lin, rgb = rdist.untile(p0, g)
    rows, cols = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    assert np.array_equal(lin[..., 0], rows) and np.array_equal(lin[..., 1], cols)
    assert np.array_equal(rgb[..., 0], rows & 255) and np.array_equal(rgb[..., 1], cols & 255)


def test_bad_parameters_are_rejected():
    lib = abi.load_rtmi()
    p = default_params(0, 10, 1)
    assert lib.rtmi_untile(C.byref(p), None, None, None) == 1  # RTMI_ERR_INVALID
    assert b"positive" in lib.rtmi_last_error()
    d = abi.SceneDesc()
    h = C.c_void_p()
    assert lib.rtmi_scene_create(C.byref(d), 0, C.byref(h)) == 1
    assert not h.value


@pytest.mark.skipif(abi.load_rtmi().rtmi_device_count() > 0, reason="CPU-only check")
def test_no_cpu_fallback_for_rendering(host):
    """Without a GPU the render path must fail loudly (no silent CPU route)."""
    from raytracing_rust_amd import HostError, scenes

    cam, world = scenes.build(host, "two_spheres", 16, 16)
    with pytest.raises(HostError) as e:
        host.lower(world).render(cam, 16, 16, 1)
    assert "no HIP device" in str(e.value)
    with pytest.raises(HostError):
        cam.render(world, 16, 16, 1)
    out = np.zeros(4, np.float32)
    assert abi.load_rtmi().rtmi_probe_math(0, out.ctypes.data, None, out.ctypes.data, 4) == 3  # RTMI_ERR_DEVICE


def _desc_with_private_trees(host, name="final_scene"):
    """A lowered scene's description whose node arrays are private ctypes copies (so a test may corrupt them)."""
    from raytracing_rust_amd import scenes

    cam, world = scenes.build(host, name, 16, 16, seed=1)
    sc = host.lower(world)
    d = sc.desc()
    nodes = (abi.BvhNode * d.n_nodes)()
    C.memmove(nodes, d.nodes, C.sizeof(nodes))
    alt = (abi.Bvh4Node * d.n_alt_nodes)()
    C.memmove(alt, d.alt_nodes, C.sizeof(alt))
    d.nodes = C.cast(nodes, C.POINTER(abi.BvhNode))
    d.alt_nodes = C.cast(alt, C.POINTER(abi.Bvh4Node))
    return sc, d, nodes, alt  # keep `sc` alive: the other arrays still belong to it


def _create_rc(d):
    lib = abi.load_rtmi()
    h = C.c_void_p()
    rc = lib.rtmi_scene_create(C.byref(d), 0, C.byref(h))
    msg = (lib.rtmi_last_error() or b"").decode()
    if h.value:
        lib.rtmi_scene_destroy(h)
    return rc, msg


def test_scene_validation_walks_the_trees(host):
    """a description arriving through the public ABI is not trusted — cycles (a persistent wavefront
    would never end), shared subtrees, understated depths (the per-lane LDS stack has the declared number of
    entries) and mistyped leaves are all rejected before anything reaches the device."""
    ok_codes = (0, 3)  # 3 = RTMI_ERR_DEVICE on a box without a GPU: validation passed, no device to upload to
    keep, d, nodes, alt = _desc_with_private_trees(host)
    rc, msg = _create_rc(d)
    assert rc in ok_codes, msg
    root = d.items[0].first
    assert d.items[0].kind == abi.ITEM_BVH and nodes[root].left >= 0

    # child cycle in the reference tree
    keep, d, nodes, alt = _desc_with_private_trees(host)
    inner = nodes[root].left
    saved = nodes[inner].left
    nodes[inner].left = root
    rc, msg = _create_rc(d)
    assert rc == 1 and "reached twice" in msg
    nodes[inner].left = saved
    # right == left is the reference's one-element BVHNode (bvh.rs:44-45): legal, visited once
    saved_r = nodes[root].right
    nodes[root].right = nodes[root].left
    rc, msg = _create_rc(d)
    assert rc in ok_codes, msg
    nodes[root].right = saved_r
    # shared subtree (two different parents, no cycle)
    other = nodes[root].right
    assert other >= 0 and nodes[other].left != inner
    saved_o = nodes[other].left
    nodes[other].left = inner
    rc, msg = _create_rc(d)
    assert rc == 1 and "reached twice" in msg
    nodes[other].left = saved_o

    # understated depths
    keep, d, nodes, alt = _desc_with_private_trees(host)
    true_depth = d.max_bvh_depth
    d.max_bvh_depth = true_depth - 1
    rc, msg = _create_rc(d)
    assert rc == 1 and "deeper than the declared depth" in msg
    d.max_bvh_depth = true_depth
    d.alt_max_depth = d.alt_max_depth - 1
    rc, msg = _create_rc(d)
    assert rc == 1 and "alternative tree" in msg and "deeper" in msg

    # alternative tree: cycle, child out of range, leaf naming a primitive of another type / out of range
    keep, d, nodes, alt = _desc_with_private_trees(host)
    aroot = d.items[0].alt_first
    assert aroot >= 0
    kids = [c for c in alt[aroot].child if 0 <= c < d.n_alt_nodes]
    assert kids
    k0 = kids[0]
    slot = [c for c in range(4) if 0 <= alt[k0].child[c] < d.n_alt_nodes or alt[k0].child[c] < 0][0]
    saved = alt[k0].child[slot]
    alt[k0].child[slot] = aroot
    rc, msg = _create_rc(d)
    assert rc == 1 and "reached twice" in msg
    alt[k0].child[slot] = d.n_alt_nodes + 5
    rc, msg = _create_rc(d)
    assert rc == 1 and "out of range" in msg
    cube0 = [i for i in range(d.n_prims) if d.prim_meta[i].type == abi.PRIM_CUBE][0]  # (the first primitives are the spheres that travel with the BVH)
    alt[k0].child[slot] = C.c_int32(0x80000000 | (abi.PRIM_SPHERE << 28) | cube0).value  # a sphere leaf naming a cube
    rc, msg = _create_rc(d)
    assert rc == 1 and "type mismatch" in msg
    alt[k0].child[slot] = C.c_int32(0x80000000 | (abi.PRIM_CUBE << 28) | (d.n_prims + 7)).value
    rc, msg = _create_rc(d)
    assert rc == 1 and "out of range" in msg
    alt[k0].child[slot] = saved
    rc, msg = _create_rc(d)
    assert rc in ok_codes, msg

    # counts without arrays
    keep, d, nodes, alt = _desc_with_private_trees(host)
    d.nodes = C.POINTER(abi.BvhNode)()
    rc, msg = _create_rc(d)
    assert rc == 1 and "NULL" in msg


def test_leaf_boxes_of_negative_radius_spheres_are_proper(host):
    """the pruned kernels' leaf box of Sphere(c, -r) must be c -+ |r| (+ pad), not the inverted
    bounding_box() of sphere.rs:79-84 that no ray passes."""
    import scenes_extra

    cam, world = scenes_extra.build(host, "hollow_glass", 16, 16, seed=1)
    arr = host.lower(world).arrays()
    assert len(arr["nodes"]) >= 2
    for n in arr["nodes"]:
        for mn, mx, ref in ((n.lmin, n.lmax, n.left), (n.rmin, n.rmax, n.right)):
            if ref < 0:  # leaf children only: an internal child's box is the reference's own (bvh.rs:60-64), inverted or not
                assert all(mn[k] < mx[k] for k in range(3)), "inverted leaf box for child %#x" % (ref & 0xffffffff)


def test_render_multi_rejects_bad_arguments_without_a_device(host):
    """rtmi_render_multi validates its arguments before it touches a device (CPU-only box: then fails loudly)."""
    from raytracing_rust_amd import scenes

    lib = abi.load_rtmi()
    cam, world = scenes.build(host, "two_spheres", 16, 16)
    sc = host.lower(world)
    d = sc.desc()
    c = cam.lower()
    lin = np.zeros((16, 16, 3), np.float32)
    dev = (C.c_int * 2)(0, 0)
    p = default_params(16, 16, 1)
    assert lib.rtmi_render_multi(C.byref(d), dev, 0, C.byref(c), C.byref(p), lin.ctypes.data, None, None) == 1   # no devices
    assert lib.rtmi_render_multi(C.byref(d), None, 2, C.byref(c), C.byref(p), lin.ctypes.data, None, None) == 1  # NULL list
    bad = default_params(16, 16, 1, tile_rank=1, tile_world=2)
    assert lib.rtmi_render_multi(C.byref(d), dev, 2, C.byref(c), C.byref(bad), lin.ctypes.data, None, None) == 1
    assert b"whole image" in lib.rtmi_last_error()
    sig = default_params(16, 16, 1, flags=abi.RTMI_FLAG_PATH_SIG)
    assert lib.rtmi_render_multi(C.byref(d), dev, 2, C.byref(c), C.byref(sig), lin.ctypes.data, None, None) == 1
    if lib.rtmi_device_count() == 0:
        assert lib.rtmi_render_multi(C.byref(d), dev, 2, C.byref(c), C.byref(p), lin.ctypes.data, None, None) == 3  # RTMI_ERR_DEVICE
        assert b"no HIP device" in lib.rtmi_last_error()
        assert lib.rtmi_scene_status(None, None) == 1


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI (no C++ in the signatures): both public headers compile as C99 with -pedantic, and a C
    translation unit that includes them links against librtmi.so and reads the ABI version."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    for h in ("rtmi.h", "rtmi_math.h"):
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, h)], check=True)
    src = tmp_path / "use.c"
    src.write_text('#include <stdio.h>\n#include "rtmi.h"\n'
                   'int main(void) { rtmi_render_params p; rtmi_scene_desc d; (void)p; (void)d;\n'
                   '  printf("%u %d\\n", (unsigned)RTMI_ABI_VERSION, rtmi_device_count() >= 0); return 0; }\n')
    exe = tmp_path / "use"
    libdir = os.path.dirname(abi.lib_path("librtmi.so")) if hasattr(abi, "lib_path") else os.path.join(root, "raytracing_rust_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lrtmi", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE).stdout.decode().split()
    assert int(out[0]) == abi.RTMI_ABI_VERSION and out[1] == "1"
