"""Randomly composed scenes (tests/scenes_random.py): every primitive / wrapper / material / texture
combination the lowering supports.  CPU: the C++ mirror's f64 evaluation equals the f64 oracle
bitwise and every scene lowers.  GPU: the device equals the fp32 oracle bit-for-bit (radiance,
quantised pixels and path signatures) for every kernel variant."""
import os

import numpy as np
import pytest

import scenes_random
from oracle.oracle import ARITH_DEVICE, SKY, THROUGHPUT_FORM
from raytracing_rust_amd import abi

SEEDS = list(range(1, 25))
if os.environ.get("RTMI_RANDOM_SEEDS"):  # e.g. "25-200": an ad-hoc wider sweep
    _lo, _hi = os.environ["RTMI_RANDOM_SEEDS"].split("-")
    SEEDS = list(range(int(_lo), int(_hi) + 1))
TOL = 1e-4  # north star: per-channel linear radiance


@pytest.mark.parametrize("instanced", [False, True], ids=["plain", "instanced"])
@pytest.mark.parametrize("seed", SEEDS[:12])
def test_mirror_equals_f64_oracle_and_lowers(host, orc64, seed, instanced):
    nx, ny = 16, 12
    cam, world = scenes_random.build(host, seed, nx, ny, instanced=instanced)
    camo, worldo = scenes_random.build(orc64, seed, nx, ny, instanced=instanced)
    row = 5
    ref = orc64.render(camo, worldo, nx, ny, 1, seed=42, rows=(row, row + 1))
    for i in range(nx):
        c = host.color_sample(cam, world, nx, ny, i, ny - 1 - row, 0, seed=42)
        assert np.array_equal(c, ref["mean"][row, i]), (seed, i)
    a = host.lower(world).arrays()
    assert len(a["items"]) >= 2 and len(a["prim_meta"]) >= 5 and a["max_bvh_depth"] <= abi.RTMI_MAX_BVH_DEPTH
    orc64.free_all()


@pytest.mark.gpu
@pytest.mark.parametrize("instanced", [False, True], ids=["plain", "instanced"])
@pytest.mark.parametrize("seed", SEEDS)
def test_device_equals_fp32_oracle_on_random_scenes(host, orc32, seed, instanced):
    """instanced: members of nested lists and BVH leaves wrapped in their own Traslate / Rotate / FlipNormals chains
    (reference: traslate.rs:17-24, rotate.rs:84-113 under bvh.rs:11-12)."""
    nx, ny, ns = 40, 24, 6
    cam, world = scenes_random.build(host, seed, nx, ny, instanced=instanced)
    sc = host.lower(world)
    camo, worldo = scenes_random.build(orc32, seed, nx, ny, instanced=instanced)
    sky = SKY if seed % 3 == 0 else 0
    ref = orc32.render(camo, worldo, nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM | sky)
    dsky = abi.RTMI_FLAG_SKY if sky else 0
    for label, flags in (("exact", 0), ("coop-fast", abi.RTMI_FLAG_FAST_CULL),
                         ("coop-fast-reftree", abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_REF_TREE),
                         ("blockcoop-fast", abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_BLOCK_COOP),  # ignored where it cannot run
                         ("perlane-fast", abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL),
                         ("async-fast", abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL)):
        got = sc.render(cam, nx, ny, ns, seed=42, flags=flags | dsky, sig=True)
        diff = np.abs(got["linear"].astype(np.float64) - ref["linear"].astype(np.float64))
        finite = np.isfinite(ref["linear"])
        assert int((diff[finite] > TOL).sum()) == 0, (seed, label, float(diff[finite].max()))
        assert np.array_equal(got["linear"], ref["linear"], equal_nan=True), (seed, label)
        assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"]), (seed, label)
        assert np.array_equal(got["sig"], ref["sig"]), (seed, label)
    print(seed, "mean radiance", float(np.nanmean(ref["linear"])), "items", len(sc.arrays()["items"]))
    orc32.free_all()


def _instanced_bvh_world(api):
    """The book's idiom the reference's generic wrappers allow: boxes turned and moved one by one, then handed to
    BVHNode::new (bvh.rs:17-66) — plus a translated moving sphere, a flipped translated rect and a nested list with a
    rotated member."""
    lamb = api.Lambertian(api.SolidTexture(0.6, 0.5, 0.4))
    metal = api.Metal(api.SolidTexture(0.8, 0.8, 0.9), 0.1)
    objs = []
    for i in range(12):
        cube = api.Cube((-0.4, -0.4, -0.4), (0.4, 0.3 + 0.05 * i, 0.4), lamb if i % 3 else metal)
        axis = [api.AXIS_Y, api.AXIS_X, api.AXIS_Z][i % 3]
        objs.append(api.Traslate(api.Rotate(axis, cube, 15.0 * i - 40.0), (-3.0 + 0.55 * i, 0.2 * (i % 4), -1.5 + 0.3 * (i % 5))))
    objs.append(api.Traslate(api.MovingSphere((0.0, 0.0, 0.0), (0.3, 0.2, 0.0), 0.0, 1.0, 0.35, lamb), (1.0, 1.2, 1.0)))
    objs.append(api.FlipNormals(api.Traslate(api.Rect(api.PLANE_XY, -1.0, -1.0, 1.0, 1.0, 0.0, lamb), (0.0, 0.5, -2.5))))
    objs.append(api.Rotate(api.AXIS_Y, api.Traslate(api.Sphere((0.0, 0.0, 0.0), 0.5, api.Dielectric(1.5)), (2.0, 0.6, 0.0)), 30.0))
    objs.append(api.Sphere((0.0, -100.5, 0.0), 100.0, lamb))
    inner = api.HittableList()
    inner.push(api.Rotate(api.AXIS_Z, api.Cube((-0.3, -0.3, -0.3), (0.3, 0.3, 0.3), metal), 25.0))
    inner.push(api.Traslate(api.Sphere((0.0, 0.0, 0.0), 0.3, lamb), (0.0, 0.9, 0.0)))
    w = api.HittableList()
    w.push(api.Sphere((0.0, 9.0, 2.0), 3.0, api.DiffuseLight(api.SolidTexture(4.0, 4.0, 4.0))))
    w.push(api.BVHNode(objs, 0.0, 1.0))
    w.push(api.Traslate(inner, (-1.5, 1.6, 1.0)))
    return w


def test_instanced_primitives_lower_with_their_own_chains(host):
    host.seed_scene_rng(1)
    a = host.lower(_instanced_bvh_world(host)).arrays()
    cnt = [(m.flags >> abi.RTMI_PRIMFLAG_XF_COUNT_SHIFT) & 15 for m in a["prim_meta"]]
    assert sorted(set(cnt)) == [0, 1, 2] and cnt.count(2) >= 13
    for m, c in zip(a["prim_meta"], cnt):
        if c:
            first = m.flags >> abi.RTMI_PRIMFLAG_XF_FIRST_SHIFT
            assert first + c <= len(a["xforms"])
    bvh = [it for it in a["items"] if it.kind == abi.ITEM_BVH][0]
    # Rotate::bounding_box is the whole space (rotate.rs:36-37): the root box saturates, the margins scale with the
    # geometry's true extent (the r = 100 ground sphere), and the alternative tree is still built
    assert bvh.root_max[0] == np.float32(3.4028235e38) and bvh.scale < 1000.0 and bvh.alt_first >= 0


@pytest.mark.gpu
def test_instanced_bvh_leaves_match_the_fp32_oracle(host, orc32):
    nx, ny, ns = 96, 64, 16
    worlds, cams = [], []
    for api in (host, orc32):
        api.seed_scene_rng(1)
        worlds.append(_instanced_bvh_world(api))
        cams.append(api.Camera((4.0, 3.0, 7.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.1, 8.0, 0.0, 1.0))
    sc = host.lower(worlds[0])
    ref = orc32.render(cams[1], worlds[1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    assert float(ref["linear"].mean()) > 0.02
    for flags in (0, abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_REF_TREE,
                  abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL):
        got = sc.render(cams[0], nx, ny, ns, seed=42, flags=flags, sig=True)
        assert np.array_equal(got["sig"], ref["sig"]), flags
        assert np.array_equal(got["linear"], ref["linear"]), flags
        assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"]), flags
    orc32.free_all()


def _bvh_world(api, with_zx_rect, sphere_times=(0.0, 1.0)):
    lamb = api.Lambertian(api.SolidTexture(0.5, 0.5, 0.5))
    objs = [api.Sphere((x, 0.0, z), 0.4, lamb) for x in (-2.0, 0.0, 2.0) for z in (-2.0, 0.0, 2.0)]
    objs.append(api.MovingSphere((0.0, 1.5, 0.0), (0.5, 1.8, 0.0), sphere_times[0], sphere_times[1], 0.4, lamb))
    if with_zx_rect:
        objs.append(api.Rect(api.PLANE_ZX, -3.0, -3.0, 3.0, 3.0, -0.6, lamb))  # its reference bbox is the XY one
    w = api.HittableList()
    w.push(api.Sphere((0.0, 8.0, 0.0), 3.0, api.DiffuseLight(api.SolidTexture(4.0, 4.0, 4.0))))
    w.push(api.BVHNode(objs, 0.0, 1.0))
    return w


def test_lowering_disables_pruning_when_boxes_do_not_contain_their_primitives(host):
    """Rect::bounding_box (rect.rs:71-75) is the XY-plane box whatever the plane: a BVH holding a ZX rect has
    boxes that do not contain it, so the lowering turns pruning off for that BVH (scale = 1e30)."""
    host.seed_scene_rng(1)
    a = host.lower(_bvh_world(host, False)).arrays()
    b = host.lower(_bvh_world(host, True)).arrays()
    assert a["items"][1].kind == abi.ITEM_BVH and a["items"][1].scale < 100.0  # item 0: the light sphere
    assert b["items"][1].kind == abi.ITEM_BVH and b["items"][1].scale == np.float32(1e30)
    d = host.lower(_bvh_world(host, False, sphere_times=(0.25, 0.75))).desc()
    assert (d.bvh_time_lo, d.bvh_time_hi) == (0.25, 0.75)
    d0 = host.lower(_bvh_world(host, False)).desc()
    assert (d0.bvh_time_lo, d0.bvh_time_hi) == (0.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["zx_rect_in_bvh", "shutter_beyond_sphere_times"])
def test_pruned_equals_exact_when_boxes_are_not_trustworthy(host, orc32, case):
    nx, ny, ns = 48, 32, 8
    times = (0.25, 0.75) if case == "shutter_beyond_sphere_times" else (0.0, 1.0)
    worlds, cams = [], []
    for api in (host, orc32):
        api.seed_scene_rng(1)
        worlds.append(_bvh_world(api, case == "zx_rect_in_bvh", times))
        cams.append(api.Camera((5.0, 4.0, 7.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 45.0, nx / ny, 0.0, 10.0, 0.0, 1.0))
    sc = host.lower(worlds[0])
    ref = orc32.render(cams[1], worlds[1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    for flags in (0, abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL):
        got = sc.render(cams[0], nx, ny, ns, seed=42, flags=flags, sig=True)
        assert np.array_equal(got["sig"], ref["sig"]), (case, flags)
        assert np.array_equal(got["linear"], ref["linear"]), (case, flags)
    orc32.free_all()


def _list_leaf_world(api):
    """A HittableList as a child of a BVHNode (bvh.rs:11-12 takes any Hittable), with exact ties inside the scan
    (hittable.rs:37-47): two coincident spheres (the FIRST wins: Sphere reports only below t_max, sphere.rs:44), two
    coincident rects (the LAST wins: Rect reports at t == t_max, rect.rs:47), a nested list with a cube and a flipped
    sphere.  Every member emits its own colour, so the winner of a tie is in the pixels."""
    red = api.DiffuseLight(api.SolidTexture(4.0, 0.2, 0.2))
    green = api.DiffuseLight(api.SolidTexture(0.2, 4.0, 0.2))
    blue = api.DiffuseLight(api.SolidTexture(0.2, 0.2, 4.0))
    grey = api.Lambertian(api.SolidTexture(0.7, 0.7, 0.7))
    lst = api.HittableList()
    lst.push(api.Sphere((0.0, 0.5, 0.0), 0.5, red))
    lst.push(api.Sphere((0.0, 0.5, 0.0), 0.5, green))
    lst.push(api.Rect(api.PLANE_XY, -2.0, 0.0, -1.0, 1.0, 0.0, blue))
    lst.push(api.Rect(api.PLANE_XY, -2.0, 0.0, -1.0, 1.0, 0.0, green))
    inner = api.HittableList()
    inner.push(api.Traslate(api.Cube((0.0, 0.0, -0.5), (1.0, 1.0, 0.5), grey), (1.0, 0.0, 0.0)))
    inner.push(api.FlipNormals(api.Sphere((1.5, 1.4, 0.0), 0.4, red)))
    lst.push(inner)
    api.seed_scene_rng(3)
    return api.BVHNode([lst, api.Sphere((0.0, -100.0, 0.0), 100.0, grey), api.Sphere((3.0, 0.5, 0.0), 0.5, grey),
                        api.FlipNormals(api.Rect(api.PLANE_XY, -4.0, 0.0, 4.0, 3.0, -2.0, grey))], 0.0, 1.0)


def test_list_as_bvh_leaf_lowers_in_tie_order(host):
    a = host.lower(_list_leaf_world(host)).arrays()
    meta, mats = a["prim_meta"], a["materials"]
    # every primitive hangs in the reference tree — primitive 0 too, whose sphere-leaf reference is 0x80000000 (the
    # lowering once used that value as its "nothing but media below" mark and dropped the leaf: r04)
    leaves = set()
    for n in a["nodes"]:
        for c in (n.left, n.right):
            if c < 0:
                leaves.add(c & 0x0fffffff)
    assert leaves == set(range(len(meta)))
    emit = [tuple(round(float(x), 1) for x in (a["textures"][mats[m.material].tex].f0, a["textures"][mats[m.material].tex].f1,
                                                a["textures"][mats[m.material].tex].f2)) for m in meta]
    types = [m.type for m in meta]
    # the list's members are numbered consecutively: spheres last-first (flipped red, green, red), then the rect-likes
    # first-first (blue rect, green rect, cube) — "ties -> the later leaf" then picks the scan's winner
    k = emit.index((0.2, 4.0, 0.2))  # the green sphere: second of its run
    assert types[k - 1:k + 5] == [abi.PRIM_SPHERE, abi.PRIM_SPHERE, abi.PRIM_SPHERE, abi.PRIM_RECT, abi.PRIM_RECT, abi.PRIM_CUBE]
    assert emit[k - 1][0] == 4.0 and meta[k - 1].flags & 1 and emit[k + 1][0] == 4.0 and not meta[k + 1].flags & 1
    assert emit[k + 2] == (0.2, 0.2, 4.0) and emit[k + 3] == (0.2, 4.0, 0.2)
    assert (meta[k + 4].flags >> abi.RTMI_PRIMFLAG_XF_COUNT_SHIFT) & 15 == 1
    big = np.float32(3.4028235e38)
    inner = [n for n in a["nodes"] if n.lmin[0] == -big and n.rmin[0] == -big]
    assert inner  # nodes of the list: boxes that pass every ray (the reference tests none below the holding node)
    gate = np.ctypeslib.as_array(host.lower(_list_leaf_world(host)).desc().prim_gate, shape=(len(meta), 8))
    assert all(np.array_equal(gate[k - 1], gate[j]) for j in range(k, k + 5))  # one gate: the box of the holding node


@pytest.mark.gpu
def test_list_as_bvh_leaf_matches_the_fp32_oracle(host, orc32):
    nx, ny, ns = 96, 64, 16
    worlds, cams = [], []
    for api in (host, orc32):
        worlds.append(_list_leaf_world(api))
        cams.append(api.Camera((0.5, 1.2, 6.0), (0.3, 0.6, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.05, 6.0, 0.0, 1.0))
    sc = host.lower(worlds[0])
    ref = orc32.render(cams[1], worlds[1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    lin = ref["linear"]
    # the tie winners are visible: red (first sphere) dominates green where the coincident spheres are seen, and the
    # coincident rects show green (the later one), not blue
    assert float(lin[..., 0].max()) > 1.0 and float(lin[..., 1].max()) > 1.0
    for flags in (0, abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_REF_TREE,
                  abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_BLOCK_COOP,
                  abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL):
        got = sc.render(cams[0], nx, ny, ns, seed=42, flags=flags, sig=True)
        assert np.array_equal(got["sig"], ref["sig"]), flags
        assert np.array_equal(got["linear"], ref["linear"]), flags
        assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"]), flags
    orc32.free_all()


def _medium_in_instanced_list_world(api):
    """ConstantMedium over a ONE-member HittableList whose member is an instanced sphere (medium.rs:11-15 takes any
    Hittable as boundary; traslate.rs:6-9 / rotate.rs:21-28 wrap any Hittable): the smoke must sit where the wrappers
    put the sphere, (1.5, 1.0, 0.5) here, not at the raw centre (0, 0, 0)."""
    w = api.HittableList()
    w.push(api.Rect(api.PLANE_ZX, -6.0, -6.0, 6.0, 6.0, -1.0, api.Lambertian(api.SolidTexture(0.6, 0.6, 0.6))))
    w.push(api.Sphere((0.0, 9.0, 0.0), 3.0, api.DiffuseLight(api.SolidTexture(5.0, 5.0, 5.0))))
    lst = api.HittableList()
    ball = api.Sphere((0.0, 0.0, 0.0), 1.0, api.Dielectric(1.5))
    lst.push(api.Traslate(api.Rotate(api.AXIS_Z, ball, 30.0), (1.5, 1.0, 0.5)))
    w.push(api.ConstantMedium(lst, 2.0, api.SolidTexture(0.9, 0.2, 0.2)))
    # a plain one: the fused boundary query must still serve it
    w.push(api.ConstantMedium(api.Sphere((-1.5, 0.2, 0.0), 0.9, api.Dielectric(1.5)), 1.0, api.SolidTexture(0.2, 0.9, 0.2)))
    return w


@pytest.mark.gpu
def test_medium_over_a_list_with_an_instanced_sphere_matches_the_fp32_oracle(host, orc32):
    """r03 advisor finding: rtmi_scene_create chose the fused sphere-boundary query for ANY medium over a one-sphere list
    and ignored the sphere's own transform chain; the cooperative kernels then disagreed with the per-lane ones."""
    nx, ny, ns = 96, 64, 24
    worlds, cams = [], []
    for api in (host, orc32):
        worlds.append(_medium_in_instanced_list_world(api))
        cams.append(api.Camera((0.5, 1.5, 7.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.05, 7.0, 0.0, 1.0))
    sc = host.lower(worlds[0])
    a = sc.arrays()
    med = [it for it in a["items"] if it.flags & abi.ITEMFLAG_MEDIUM]
    assert len(med) == 2 and med[0].count == 1
    assert (a["prim_meta"][med[0].first].flags >> abi.RTMI_PRIMFLAG_XF_COUNT_SHIFT) & 15 == 2  # the chain is the primitive's
    ref = orc32.render(cams[1], worlds[1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    lin = ref["linear"]
    # the red smoke is seen right of the centre (where the wrappers put it), the green one left
    assert float((lin[..., 0] - lin[..., 1])[:, nx // 2:].max()) > 0.05 and float((lin[..., 1] - lin[..., 0])[:, :nx // 2].max()) > 0.05
    for flags in (0, abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_BLOCK_COOP,
                  abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL):
        got = sc.render(cams[0], nx, ny, ns, seed=42, flags=flags, sig=True)
        assert np.array_equal(got["sig"], ref["sig"]), flags
        assert np.array_equal(got["linear"], ref["linear"]), flags
        assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"]), flags
    orc32.free_all()


def test_medium_over_a_list_with_an_instanced_sphere_mirror_equals_f64_oracle(host, orc64):
    nx, ny = 24, 16
    cams = [api.Camera((0.5, 1.5, 7.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.05, 7.0, 0.0, 1.0) for api in (host, orc64)]
    wh, wo = _medium_in_instanced_list_world(host), _medium_in_instanced_list_world(orc64)
    for row in (5, 8, 11):
        ref = orc64.render(cams[1], wo, nx, ny, 1, seed=42, rows=(row, row + 1))
        for i in range(nx):
            c = host.color_sample(cams[0], wh, nx, ny, i, ny - 1 - row, 0, seed=42)
            assert np.array_equal(c, ref["mean"][row, i]), (row, i)
    orc64.free_all()


@pytest.mark.gpu
def test_compositions_scene_matches_the_fp32_oracle(host, orc32):
    """The scene of tests/golden/flat_compositions.bin.gz (tools/dump_flat_scene.py: list leaves with ties, instanced
    primitives, a flipped subtree, a medium inside Traslate(Rotate(..))) on the device: every kernel == the fp32 oracle."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dump_flat_scene as dfs

    nx, ny, ns = 96, 64, 16
    worlds, cams = [], []
    for api in (host, orc32):
        worlds.append(dfs.compositions(api, 1))
        cams.append(api.Camera((0.5, 1.5, 7.0), (0.0, 0.8, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.05, 7.0, 0.0, 1.0))
    sc = host.lower(worlds[0])
    ref = orc32.render(cams[1], worlds[1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    assert float(ref["linear"].mean()) > 0.02
    for flags in (0, abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_REF_TREE,
                  abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL):
        got = sc.render(cams[0], nx, ny, ns, seed=42, flags=flags, sig=True)
        assert np.array_equal(got["sig"], ref["sig"]), flags
        assert np.array_equal(got["linear"], ref["linear"]), flags
        assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"]), flags
    orc32.free_all()
