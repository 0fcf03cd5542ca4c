"""tests/golden/flat_*.bin.gz are byte dumps of the C++ lowering (rt_host.cpp SceneBuilder) of cornell_box,
final_scene (scene seed 1) and `compositions` (what the lowering accepts beyond the reference's own scenes: list leaves,
instanced primitives, a flipped subtree, a medium inside transforms), written by tools/dump_flat_scene.py.  They exist for the Rust shim: bindings/rust/src/lower.rs
mirrors that lowering function by function and cannot be compiled in this image; on a machine with cargo its output
(`rtmi::dump::flat_scene_bytes`) must equal these bytes.  Here: the fixtures still match the current C++ lowering, are
self-consistent, and the Rust sources reference them."""
import gzip
import hashlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dump_flat_scene as dfs  # noqa: E402


def test_fixtures_match_the_current_cpp_lowering():
    for name in dfs.SCENES:
        want = gzip.decompress(open(dfs.path_of(name), "rb").read())
        got = dfs.dump(name)
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest(), \
            name + ": regenerate with `python tools/dump_flat_scene.py` after an intended change of the lowering"


def test_dump_layout_is_self_consistent():
    data = gzip.decompress(open(dfs.path_of("final_scene"), "rb").read())
    assert data[:8] == b"RTMIFLT1"
    n_items, n_prims, n_nodes, n_alt, n_xf, n_mat, n_tex, n_perlin, n_img, depth, alt_depth = struct.unpack_from("<11I", data, 8)
    lo, hi = struct.unpack_from("<2f", data, 52)
    image_bytes, image_hash = struct.unpack_from("<2Q", data, 60)
    body = 64 * n_items + 16 * n_prims * 2 + 16 * n_prims + 32 * n_prims + 64 * n_nodes + 128 * n_alt + 16 * n_xf + 16 * n_mat + \
        32 * n_tex + 7168 * n_perlin + 16 * n_img
    assert len(data) == 76 + body
    # final_scene as the reference builds it (tests/test.rs:419-523): 400 cubes + 8 top-level primitives (5 spheres,
    # 2 medium boundaries, ...; the light rect has x0 > x1, can never be hit (rect.rs:51) and is left out of the list scan)
    # + 1000 spheres; 11 world objects in 6 items (runs of plain primitives merge)
    assert (n_items, n_prims, n_perlin, n_img, image_bytes) == (6, 1408, 1, 1, 1024 * 512 * 3)
    # bvh.rs:44-45: a slice of one element becomes a node of its own, so there are more than N - 1 nodes
    assert n_nodes >= 399 + 999 and n_alt > 0 and depth <= 24 and alt_depth <= depth
    assert n_xf == 2  # Traslate(Rotate(Y, BVH)) (:517-522)
    from raytracing_rust_amd import scenes

    assert dfs.fnv1a64(bytes(scenes.earthmap_rgb8()[0])) == image_hash
    assert lo < -3e38 and hi > 3e38  # the only MovingSphere is a top-level list primitive, not inside a BVH


def test_rust_sources_name_the_goldens():
    src = os.path.join(ROOT, "bindings", "rust", "src")
    for f in ("lower.rs", "desc.rs", "scenes.rs", "dump.rs", "philox.rs", "lib.rs", "sys.rs"):
        assert os.path.exists(os.path.join(src, f)), f
    dump = open(os.path.join(src, "dump.rs")).read()
    assert "RTMIFLT1" in dump and "flat_{}.bin" in dump and "scenes::compositions(1)" in dump
    assert "pub fn compositions(seed: u64)" in open(os.path.join(src, "scenes.rs")).read()
    lower = open(os.path.join(src, "lower.rs")).read()
    for fn in ("fn push_prim", "fn lower_bvh", "fn build_alt_tree", "fn collapse_alt", "fn lower_item", "fn lower_world",
               "fn contained", "fn true_bounds", "fn lower_leaf", "fn lower_list_leaf", "fn list_subtree", "fn plan_collapse", "fn emit_slots"):
        assert fn in lower, fn
