#!/usr/bin/env python3
"""Golden dumps of the C++ lowering (raytracing_rust_amd/host/rt_host.cpp) for the Rust shim to diff against.

    python tools/dump_flat_scene.py [--check]      writes / checks tests/golden/flat_<scene>.bin.gz

bindings/rust/src/lower.rs mirrors the C++ SceneBuilder function by function and cannot be compiled in this image
(no Rust toolchain).  On a machine with cargo, `rtmi::dump::flat_scene_bytes(&lower_world(&scenes::final_scene(1,
false, earth))?)` must equal the gunzipped bytes of tests/golden/flat_final_scene.bin.gz byte for byte; likewise
cornell_box, and `scenes::compositions(1)` (the compositions beyond the reference's own scenes: list leaves, instanced
primitives, flipped subtrees, a medium inside transforms) against flat_compositions.bin.gz, and `scenes::media_in_bvh(1)` (media as
children of BVHNodes: deferred items) against flat_media_in_bvh.bin.gz.  Format "RTMIFLT1" (little endian):
    8 B magic | 11 x u32: n_items n_prims n_nodes n_alt_nodes n_xforms n_materials n_textures n_perlin n_images
    max_bvh_depth alt_max_depth | 2 x f32: bvh_time_lo bvh_time_hi | u64 image_bytes | u64 FNV-1a of image_data
    then the arrays of include/rtmi.h, raw, in this order: items, prim_a, prim_b, prim_meta, prim_gate, nodes,
    alt_nodes, xforms, materials, textures, perlin, images   (reserved / pad words written as zero)
"""
import ctypes as C
import gzip
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCENES = ["cornell_box", "final_scene", "compositions", "media_in_bvh"]


def fnv1a64(data):
    """FNV-1a, 64 bit, over the bytes (sequential by construction: ~1 s for the 1.5 MB earth texture)."""
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def flat_scene_bytes(desc):
    from raytracing_rust_amd import abi

    d = desc
    out = [b"RTMIFLT1",
           struct.pack("<11I", d.n_items, d.n_prims, d.n_nodes, d.n_alt_nodes, d.n_xforms, d.n_materials, d.n_textures,
                       d.n_perlin, d.n_images, d.max_bvh_depth, d.alt_max_depth),
           struct.pack("<2f", d.bvh_time_lo, d.bvh_time_hi)]
    img = C.string_at(d.image_data, d.image_bytes) if d.image_bytes else b""
    out.append(struct.pack("<2Q", d.image_bytes, fnv1a64(img)))

    def raw(ptr, count, elem_size):
        return C.string_at(ptr, count * elem_size) if count else b""

    out.append(raw(d.items, d.n_items, C.sizeof(abi.Item)))
    out.append(raw(d.prim_a, d.n_prims, 16))
    out.append(raw(d.prim_b, d.n_prims, 16))
    out.append(raw(d.prim_meta, d.n_prims, C.sizeof(abi.PrimMeta)))
    out.append(raw(d.prim_gate, d.n_prims, 32))
    nodes = (abi.BvhNode * d.n_nodes)()
    if d.n_nodes:
        C.memmove(nodes, d.nodes, C.sizeof(nodes))
    for n in nodes:
        n.pad[0] = n.pad[1] = 0
    out.append(bytes(nodes))
    alt = (abi.Bvh4Node * d.n_alt_nodes)()
    if d.n_alt_nodes:
        C.memmove(alt, d.alt_nodes, C.sizeof(alt))
    for n in alt:
        for k in range(4):
            n.pad[k] = 0
    out.append(bytes(alt))
    out.append(raw(d.xforms, d.n_xforms, C.sizeof(abi.Xform)))
    out.append(raw(d.materials, d.n_materials, C.sizeof(abi.Material)))
    out.append(raw(d.textures, d.n_textures, C.sizeof(abi.Texture)))
    out.append(raw(d.perlin, d.n_perlin, C.sizeof(abi.Perlin)))
    out.append(raw(d.images, d.n_images, C.sizeof(abi.ImageDesc)))
    return b"".join(out)


def compositions(api, seed=1):
    """The compositions the lowering accepts beyond the reference's own scenes, in one world (twin of
    bindings/rust/src/scenes.rs `compositions`): a HittableList as a BVH child with exact ties inside (coincident
    spheres, coincident rects, a nested list), instanced primitives (own Traslate / Rotate / FlipNormals chains) as
    BVH leaves and list members, FlipNormals around an inner BVHNode, a ConstantMedium inside Traslate(Rotate(..))."""
    api.seed_scene_rng(seed)
    red = api.DiffuseLight(api.SolidTexture(4.0, 0.2, 0.2))
    green = api.DiffuseLight(api.SolidTexture(0.2, 4.0, 0.2))
    blue = api.DiffuseLight(api.SolidTexture(0.2, 0.2, 4.0))
    grey = api.Lambertian(api.SolidTexture(0.7, 0.7, 0.7))
    glass = api.Dielectric(1.5)
    lst = api.HittableList()
    lst.push(api.Sphere((0.0, 0.5, 0.0), 0.5, red))
    lst.push(api.Sphere((0.0, 0.5, 0.0), 0.5, green))
    lst.push(api.Rect(api.PLANE_XY, -2.0, 0.0, -1.0, 1.0, 0.0, blue))
    lst.push(api.Rect(api.PLANE_XY, -2.0, 0.0, -1.0, 1.0, 0.0, green))
    inner = api.HittableList()
    inner.push(api.Traslate(api.Cube((0.0, 0.0, -0.5), (1.0, 1.0, 0.5), grey), (1.0, 0.0, 0.0)))
    inner.push(api.FlipNormals(api.Sphere((1.5, 1.4, 0.0), 0.4, red)))
    lst.push(inner)
    sub = api.BVHNode([api.Sphere((-3.0, 0.5, 1.0), 0.5, grey), api.Rotate(api.AXIS_Y, api.Cube((-0.3, 0.0, -0.3), (0.3, 0.8, 0.3), grey), 30.0),
                       api.Sphere((-3.0, 0.5, -1.0), 0.5, glass)], 0.0, 1.0)
    bvh = api.BVHNode([lst, api.Sphere((0.0, -100.0, 0.0), 100.0, grey), api.Sphere((3.0, 0.5, 0.0), 0.5, grey),
                       api.FlipNormals(sub), api.FlipNormals(api.Rect(api.PLANE_XY, -4.0, 0.0, 4.0, 3.0, -2.0, grey)),
                       api.Traslate(api.Rotate(api.AXIS_Z, api.MovingSphere((0.0, 0.0, 0.0), (0.0, 0.3, 0.0), 0.0, 1.0, 0.3, grey), 20.0),
                                    (2.0, 2.0, 1.0))], 0.0, 1.0)
    world = api.HittableList()
    world.push(bvh)
    world.push(api.Traslate(api.Rotate(api.AXIS_Y, api.ConstantMedium(api.Sphere((0.0, 0.0, 0.0), 1.0, glass), 0.5,
                                                                      api.SolidTexture(0.9, 0.9, 0.9)), 25.0), (-1.5, 1.5, 2.0)))
    world.push(api.Sphere((0.0, 6.0, 0.0), 1.5, api.DiffuseLight(api.SolidTexture(4.0, 4.0, 4.0))))
    return world


def media_in_bvh(api, seed=1):
    """ConstantMedium as a child of a BVHNode (r04; twin of bindings/rust/src/scenes.rs `media_in_bvh`): a BVH of primitives
    and media inside Traslate(Rotate(..)) — one medium around a plain boundary, one itself inside a Traslate, and an instanced
    subtree (Traslate(Rotate(BVHNode))) —, a BVHNode
    over ONE object that is a BVH with a medium in it (evaluated on both sides: its medium twice), and a BVHNode over one
    medium (no primitives at all: no BVH item, the first deferred item remembers T0 itself); a nested medium; a list with media
    (and a nested flipped list) as a child of a BVHNode."""
    api.seed_scene_rng(seed)
    grey = api.Lambertian(api.SolidTexture(0.7, 0.7, 0.7))
    glass = api.Dielectric(1.5)
    objs = [api.Sphere((-3.0, 0.0, 0.0), 0.9, grey),
            api.ConstantMedium(api.Sphere((-1.0, 0.2, 0.5), 1.0, glass), 1.5, api.SolidTexture(0.9, 0.2, 0.2)),
            api.Cube((0.3, -1.0, -0.8), (1.5, 0.4, 0.6), grey),
            api.Traslate(api.ConstantMedium(api.Cube((0.0, 0.0, 0.0), (1.2, 1.2, 1.2), glass), 2.5, api.SolidTexture(0.2, 0.9, 0.2)), (1.8, -0.9, 0.8)),
            api.Sphere((3.3, 0.1, -0.3), 0.8, grey)]
    # an instanced subtree (Traslate / Rotate around a BVHNode as a child of a BVHNode): a deferred BVH item with its gate records
    sub = api.BVHNode([api.Sphere((0.0, 0.0, 0.0), 0.4, grey), api.Cube((0.5, -0.3, -0.3), (1.1, 0.3, 0.3), grey)], 0.0, 1.0)
    objs.append(api.Traslate(api.Rotate(api.AXIS_Z, sub, 20.0), (4.5, 1.0, 0.5)))
    inner = api.BVHNode([api.Sphere((-4.5, 1.2, 1.5), 0.5, grey),
                         api.ConstantMedium(api.Sphere((-4.2, 1.3, 1.4), 1.0, glass), 1.0, api.SolidTexture(0.4, 0.9, 0.6)),
                         api.Cube((-5.6, 0.2, 0.8), (-5.0, 0.9, 1.6), grey)], 0.0, 1.0)
    world = api.HittableList()
    world.push(api.Traslate(api.Rotate(api.AXIS_Y, api.BVHNode(objs, 0.0, 1.0), -30.0), (-1.0, 1.2, 3.0)))
    world.push(api.BVHNode([inner], 0.0, 1.0))
    world.push(api.BVHNode([api.ConstantMedium(api.Sphere((0.0, 2.6, -1.5), 0.8, glass), 1.2, api.SolidTexture(0.9, 0.8, 0.2))], 0.0, 1.0))
    # a nested medium (one item, the inner density behind its chain) ...
    world.push(api.ConstantMedium(api.ConstantMedium(api.Sphere((5.0, 3.0, -2.0), 0.7, glass), 0.8, api.SolidTexture(0.1, 0.1, 0.1)), 2.0,
                                  api.SolidTexture(0.9, 0.5, 0.2)))
    # ... and a HittableList WITH MEDIA as a child of a BVHNode: a group of LISTSCAN members and a terminator behind the BVH item
    inner_l = api.HittableList()
    inner_l.push(api.Traslate(api.Cube((-0.3, -0.3, -0.3), (0.3, 0.3, 0.3), grey), (6.9, -0.2, 2.1)))
    inner_l.push(api.ConstantMedium(api.Cube((6.0, 0.3, 1.2), (7.0, 0.9, 2.0), glass), 3.0, api.SolidTexture(0.9, 0.9, 0.3)))
    lst = api.HittableList()
    lst.push(api.Sphere((6.0, 0.0, 2.0), 0.4, grey))
    lst.push(api.ConstantMedium(api.Sphere((6.3, 0.1, 1.8), 0.8, glass), 2.0, api.SolidTexture(0.3, 0.9, 0.4)))
    lst.push(api.FlipNormals(inner_l))
    world.push(api.BVHNode([lst, api.Sphere((8.0, 0.0, 2.0), 0.5, grey)], 0.0, 1.0))
    world.push(api.Sphere((0.0, 9.0, 0.0), 2.0, api.DiffuseLight(api.SolidTexture(4.0, 4.0, 4.0))))
    return world


def dump(name):
    from raytracing_rust_amd import Host, scenes

    host = Host()
    if name == "compositions":
        world = compositions(host, 1)
    elif name == "media_in_bvh":
        world = media_in_bvh(host, 1)
    else:
        _, world = scenes.build(host, name, 64, 64, seed=1)
    sc = host.lower(world)
    data = flat_scene_bytes(sc.desc())
    host.free_all()
    return data


def path_of(name):
    return os.path.join(ROOT, "tests", "golden", "flat_%s.bin.gz" % name)


def main():
    check = "--check" in sys.argv
    for name in SCENES:
        data = dump(name)
        if check:
            old = gzip.decompress(open(path_of(name), "rb").read())
            print(name, "OK" if old == data else "DIFFERS", len(data), "bytes")
            if old != data:
                sys.exit(1)
        else:
            with open(path_of(name), "wb") as f:
                f.write(gzip.compress(data, 9, mtime=0))
            print(name, len(data), "bytes ->", os.path.getsize(path_of(name)), "gz")


if __name__ == "__main__":
    main()
