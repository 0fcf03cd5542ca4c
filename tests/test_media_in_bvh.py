"""ConstantMedium as a CHILD OF A BVHNode (r03 verdict, missing 5: bvh.rs:11-12 takes any Rc<dyn Hittable>).

BVHNode::hit (bvh.rs:70-89) hands both children the query's own (t_min, t_max) and keeps the closer hit, so a medium
child is evaluated with the t_max the BVH was entered with, draws its random number whenever its clamped boundary
interval is not empty (medium.rs:30-40) at its in-order position, and is reached iff its parent's box passes; a node over
one element (bvh.rs:44-45) evaluates — and draws — twice.  The lowering keeps media out of the device's trees: each becomes
a DEFERRED item behind its BVH item (include/rtmi.h).  Here: what the lowering emits (CPU), the C++ mirror against the f64
oracle (CPU, both walk the object graph like the reference), and every device kernel against the fp32 oracle bit for bit."""
import numpy as np
import pytest

from oracle.oracle import ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import abi


def world_media_in_bvh(api):
    api.seed_scene_rng(3)
    lam = lambda r, g, b: api.Lambertian(api.SolidTexture(r, g, b))  # noqa: E731
    glass = api.Dielectric(1.5)
    w = api.HittableList()
    w.push(api.Rect(api.PLANE_ZX, -8.0, -8.0, 8.0, 8.0, -1.0, lam(0.6, 0.6, 0.6)))
    w.push(api.Sphere((0.0, 10.0, 2.0), 3.0, api.DiffuseLight(api.SolidTexture(5.0, 5.0, 5.0))))
    # 1. a BVH of primitives AND media (also a medium inside its own Traslate, and one around a transformed boundary)
    objs = [
        api.Sphere((-3.0, 0.0, 0.0), 0.9, lam(0.8, 0.3, 0.3)),
        api.ConstantMedium(api.Sphere((-1.0, 0.2, 0.5), 1.0, glass), 1.5, api.SolidTexture(0.9, 0.2, 0.2)),
        api.Cube((0.3, -1.0, -0.8), (1.5, 0.4, 0.6), lam(0.3, 0.8, 0.3)),
        api.Traslate(api.ConstantMedium(api.Cube((0.0, 0.0, 0.0), (1.2, 1.2, 1.2), glass), 2.5, api.SolidTexture(0.2, 0.9, 0.2)), (1.8, -0.9, 0.8)),
        api.Sphere((3.3, 0.1, -0.3), 0.8, api.Metal(api.SolidTexture(0.8, 0.8, 0.9), 0.1)),
        api.ConstantMedium(api.Rotate(api.AXIS_Y, api.Cube((-0.5, -0.5, -0.5), (0.5, 0.7, 0.5), glass), 25.0), 4.0, api.SolidTexture(0.2, 0.3, 0.9)),
        api.MovingSphere((-2.0, 1.6, -1.0), (-1.7, 1.9, -1.0), 0.0, 1.0, 0.4, lam(0.7, 0.7, 0.2)),
    ]
    w.push(api.BVHNode(objs, 0.0, 1.0))
    # 2. a BVHNode over ONE medium: left and right are the same object — evaluated, and drawn, twice; no primitives at all
    w.push(api.BVHNode([api.ConstantMedium(api.Sphere((0.0, 2.6, -1.5), 0.8, glass), 1.2, api.SolidTexture(0.9, 0.8, 0.2))], 0.0, 1.0))
    # 3. inside Traslate(Rotate(..)): the enclosing transforms in front of the medium's own; an inner BVH of media only
    inner = api.BVHNode([api.ConstantMedium(api.Sphere((0.0, 0.0, 0.0), 0.6, glass), 3.0, api.SolidTexture(0.9, 0.4, 0.9)),
                         api.ConstantMedium(api.Sphere((1.0, 0.3, 0.2), 0.5, glass), 0.7, api.SolidTexture(0.3, 0.9, 0.9))], 0.0, 1.0)
    outer = api.BVHNode([inner, api.Sphere((-1.2, 0.0, 0.0), 0.5, lam(0.9, 0.9, 0.9)), api.Cube((1.8, -0.4, -0.4), (2.5, 0.4, 0.4), lam(0.5, 0.5, 0.9)),
                         api.ConstantMedium(api.Traslate(api.Sphere((0.0, 0.0, 0.0), 0.45, glass), (0.4, 0.9, 0.0)), 6.0, api.SolidTexture(0.9, 0.9, 0.9))],
                        0.0, 1.0)
    w.push(api.Traslate(api.Rotate(api.AXIS_Y, outer, -30.0), (-1.0, 1.2, 3.0)))
    # 4. a BVHNode over ONE object that is itself a BVH with a medium in it: the inner BVH is evaluated on both sides
    # (bvh.rs:73-74), so its medium draws twice — found by the random scenes (seed 4), not by cases 1-3
    inner2 = api.BVHNode([api.Sphere((-4.5, 1.2, 1.5), 0.5, lam(0.9, 0.5, 0.2)),
                          api.ConstantMedium(api.Sphere((-4.2, 1.3, 1.4), 1.0, glass), 1.0, api.SolidTexture(0.4, 0.9, 0.6)),
                          api.Cube((-5.6, 0.2, 0.8), (-5.0, 0.9, 1.6), lam(0.4, 0.4, 0.9))], 0.0, 1.0)
    w.push(api.BVHNode([inner2], 0.0, 1.0))
    return w


def world_instanced_subtrees(api):
    """A BVHNode inside Traslate / Rotate as a child of a BVHNode (an instanced subtree): alone, twice under a one-element
    node, nested in another instanced subtree, flipped, with a medium and an instanced primitive inside, and the whole
    thing inside an item transform."""
    api.seed_scene_rng(7)
    lam = lambda r, g, b: api.Lambertian(api.SolidTexture(r, g, b))  # noqa: E731
    glass = api.Dielectric(1.5)
    w = api.HittableList()
    w.push(api.Rect(api.PLANE_ZX, -9.0, -9.0, 9.0, 9.0, -1.0, lam(0.6, 0.6, 0.6)))
    w.push(api.Sphere((0.0, 10.0, 2.0), 3.0, api.DiffuseLight(api.SolidTexture(5.0, 5.0, 5.0))))
    cluster = lambda col: api.BVHNode([api.Sphere((0.0, 0.0, 0.0), 0.45, lam(*col)), api.Cube((0.5, -0.4, -0.4), (1.2, 0.4, 0.4), lam(0.8, 0.8, 0.8)),  # noqa: E731
                                       api.Sphere((0.3, 0.8, 0.1), 0.35, api.Metal(api.SolidTexture(0.9, 0.9, 0.9), 0.05)),
                                       api.Rotate(api.AXIS_Z, api.Cube((-1.0, -0.3, -0.3), (-0.5, 0.3, 0.3), lam(0.3, 0.7, 0.9)), 20.0)], 0.0, 1.0)
    deep = api.BVHNode([api.Traslate(cluster((0.9, 0.9, 0.2)), (0.0, 1.6, 0.0)), api.Sphere((0.0, 0.0, 0.0), 0.4, lam(0.9, 0.3, 0.9)),
                        api.ConstantMedium(api.Sphere((0.8, 0.4, 0.0), 0.6, glass), 2.0, api.SolidTexture(0.3, 0.9, 0.5))], 0.0, 1.0)
    objs = [
        api.Sphere((-4.0, 0.0, 0.0), 0.9, lam(0.8, 0.3, 0.3)),
        api.Traslate(api.Rotate(api.AXIS_Y, cluster((0.9, 0.2, 0.2)), 35.0), (-1.5, 0.0, 0.5)),
        api.Cube((0.8, -1.0, -0.6), (1.8, 0.2, 0.6), lam(0.3, 0.8, 0.3)),
        api.FlipNormals(api.Rotate(api.AXIS_X, api.Traslate(deep, (3.6, 0.2, 0.0)), 10.0)),
        api.BVHNode([api.Traslate(cluster((0.2, 0.9, 0.9)), (-2.8, 1.9, -1.2))], 0.0, 1.0),  # a one-element node over an instanced subtree: twice
        # a medium whose boundary is itself a BVHNode (two overlapping shapes), as a child of the enclosing BVHNode
        api.ConstantMedium(api.BVHNode([api.Sphere((3.2, 1.9, 1.0), 0.7, glass), api.Cube((3.0, 1.2, 0.4), (4.2, 2.0, 1.3), glass)], 0.0, 1.0), 1.2,
                           api.SolidTexture(0.9, 0.6, 0.2)),
    ]
    w.push(api.Traslate(api.Rotate(api.AXIS_Y, api.BVHNode(objs, 0.0, 1.0), -15.0), (0.3, 0.2, 1.0)))
    return w


def world_nested_media(api):
    """A ConstantMedium whose boundary is a ConstantMedium (medium.rs:11-15 is generic over any Hittable): the inner one
    answers either boundary query of the outer one with a random distance of its own — three draws per reached pair.  In the
    world list, inside transforms, around several primitives, around a BVHNode, and as children of a BVHNode."""
    api.seed_scene_rng(11)
    lam = lambda r, g, b: api.Lambertian(api.SolidTexture(r, g, b))  # noqa: E731
    glass = api.Dielectric(1.5)
    fog = lambda boundary, d_in, d_out, r, g, b: api.ConstantMedium(api.ConstantMedium(boundary, d_in, api.SolidTexture(0.1, 0.1, 0.1)),  # noqa: E731
                                                                     d_out, api.SolidTexture(r, g, b))
    w = api.HittableList()
    w.push(api.Rect(api.PLANE_ZX, -9.0, -9.0, 9.0, 9.0, -1.0, lam(0.6, 0.6, 0.6)))
    w.push(api.Sphere((0.0, 10.0, 2.0), 3.0, api.DiffuseLight(api.SolidTexture(5.0, 5.0, 5.0))))
    w.push(fog(api.Sphere((-3.5, 0.3, 0.0), 1.2, glass), 0.8, 1.5, 0.9, 0.3, 0.2))                        # one sphere
    w.push(api.Traslate(api.Rotate(api.AXIS_Y, fog(api.Cube((-0.7, -0.7, -0.7), (0.7, 0.7, 0.7), glass), 1.5, 0.6, 0.2, 0.9, 0.3), 30.0),
                        (-0.8, 0.0, 0.5)))                                                                # both inside transforms
    two = api.HittableList()
    two.push(api.Sphere((1.6, 0.0, 0.0), 0.8, glass))
    two.push(api.Traslate(api.Sphere((0.0, 0.0, 0.0), 0.6, glass), (2.4, 0.5, 0.3)))
    w.push(api.FlipNormals(fog(two, 0.5, 2.0, 0.2, 0.4, 0.9)))                                             # a list boundary
    cluster = api.BVHNode([api.Sphere((4.0 + 0.5 * k, 0.2 * k, -0.5), 0.45, glass) for k in range(4)], 0.0, 1.0)
    w.push(fog(cluster, 0.9, 1.1, 0.9, 0.9, 0.2))                                                          # a BVHNode boundary
    objs = [api.Sphere((-3.0, 2.6, -1.0), 0.6, lam(0.8, 0.3, 0.3)),
            fog(api.Sphere((-1.2, 2.6, -1.0), 0.9, glass), 0.7, 1.8, 0.3, 0.9, 0.9),                       # children of a BVHNode
            api.Cube((0.2, 2.0, -1.6), (1.2, 3.0, -0.6), lam(0.3, 0.8, 0.3)),
            api.Traslate(fog(api.BVHNode([api.Sphere((0.0, 0.0, 0.0), 0.5, glass), api.Sphere((0.7, 0.2, 0.0), 0.5, glass)], 0.0, 1.0),
                             1.2, 0.9, 0.9, 0.5, 0.9), (2.4, 2.6, -1.0))]
    w.push(api.Traslate(api.BVHNode(objs, 0.0, 1.0), (0.0, 0.2, 0.0)))
    return w


def world_media_in_lists_in_bvh(api):
    """A HittableList WITH MEDIA among its members as a child of a BVHNode: the list's scan hands every member the closest
    hit of the members before it (from the t_max the BVH was entered with), so a medium behind a nearer member of the SAME
    list draws less often than one that sits in the tree by itself.  Primitives before and after the medium, a nested list,
    flips, members with their own transforms, two media in one list, a nested medium, the list twice under a one-element
    node, and everything once more inside Traslate(Rotate(..))."""
    api.seed_scene_rng(5)
    lam = lambda r, g, b: api.Lambertian(api.SolidTexture(r, g, b))  # noqa: E731
    glass = api.Dielectric(1.5)
    w = api.HittableList()
    w.push(api.Rect(api.PLANE_ZX, -9.0, -9.0, 9.0, 9.0, -1.0, lam(0.6, 0.6, 0.6)))
    w.push(api.Sphere((0.0, 10.0, 2.0), 3.0, api.DiffuseLight(api.SolidTexture(5.0, 5.0, 5.0))))

    def group(cx, cy, cz, nested=False):
        g = api.HittableList()
        g.push(api.Sphere((cx - 0.5, cy, cz + 0.6), 0.45, lam(0.9, 0.3, 0.3)))      # in front of the medium: shortens its interval
        med = api.ConstantMedium(api.Sphere((cx, cy, cz), 0.9, glass), 2.5, api.SolidTexture(0.3, 0.9, 0.4))
        if nested:
            med = api.ConstantMedium(api.ConstantMedium(api.Sphere((cx, cy, cz), 0.9, glass), 0.8, api.SolidTexture(0.1, 0.1, 0.1)), 2.0,
                                     api.SolidTexture(0.9, 0.5, 0.2))
        g.push(med)
        inner = api.HittableList()                                                    # a nested list: its scan continues the outer one
        inner.push(api.Traslate(api.Cube((-0.3, -0.3, -0.3), (0.3, 0.3, 0.3), lam(0.3, 0.3, 0.9)), (cx + 0.7, cy - 0.3, cz + 0.3)))
        inner.push(api.FlipNormals(api.ConstantMedium(api.Cube((cx - 0.2, cy + 0.2, cz - 0.6), (cx + 1.1, cy + 0.9, cz + 0.2), glass), 3.0,
                                                      api.SolidTexture(0.9, 0.9, 0.3))))
        g.push(api.FlipNormals(inner))
        g.push(api.Rect(api.PLANE_XY, cx - 1.0, cy - 0.8, cx + 1.0, cy + 0.8, cz - 0.95, api.Metal(api.SolidTexture(0.8, 0.8, 0.8), 0.1)))
        return g

    objs = [api.Sphere((-4.0, 0.0, 0.0), 0.8, lam(0.7, 0.7, 0.2)), group(-1.5, 0.2, 0.0), api.Cube((0.4, -1.0, -0.6), (1.4, 0.2, 0.4), lam(0.3, 0.8, 0.8)),
            api.FlipNormals(group(3.0, 0.3, -0.2, nested=True)), api.Sphere((5.2, 0.0, 0.5), 0.6, glass)]
    w.push(api.BVHNode(objs, 0.0, 1.0))
    w.push(api.BVHNode([group(-3.0, 3.0, -1.5)], 0.0, 1.0))                           # one element: the list is scanned — and draws — twice
    tilted = api.BVHNode([group(0.0, 0.0, 0.0), api.Sphere((1.8, 0.0, 0.0), 0.5, lam(0.9, 0.9, 0.9)),
                          api.ConstantMedium(api.Sphere((-1.8, 0.2, 0.0), 0.6, glass), 1.5, api.SolidTexture(0.6, 0.3, 0.9))], 0.0, 1.0)
    w.push(api.Traslate(api.Rotate(api.AXIS_Y, tilted, 25.0), (1.5, 3.0, -1.0)))
    return w


def test_lists_with_media_lower_as_scan_groups(host):
    a = host.lower(world_media_in_lists_in_bvh(host)).arrays()
    items = a["items"]
    B, M, E = abi.ITEMFLAG_LISTSCAN_BEGIN, abi.ITEMFLAG_LISTSCAN_MEMBER, abi.ITEMFLAG_LISTSCAN_END
    ends = [k for k, it in enumerate(items) if it.flags & E]
    begins = [k for k, it in enumerate(items) if it.flags & B]
    assert len(begins) == len(ends) >= 5  # 2 + 2 (the one-element node) + 1, plus whatever BVHNode::new left alone in a slice of one
    for b, e in zip(begins, ends):
        grp = items[b:e]
        assert b < e and all((it.flags & M) and (it.flags & abi.ITEMFLAG_DEFERRED) for it in grp) and not (items[e].flags & M)
        assert items[e].kind == abi.ITEM_LIST and items[e].count == 0 and items[e].first >= 0
        # sphere, medium, cube (own transform), medium (flipped twice: not flipped), rect
        assert [bool(it.flags & abi.ITEMFLAG_MEDIUM) for it in grp] == [False, True, False, True, False]
        assert [it.kind for it in grp] == [abi.ITEM_LIST] * 5 and all(it.count == 1 for it in grp)
        # every member's gate is the holder's box: one box for the group
        gates = {tuple(np.asarray(a["prim_gate"][it.first]).ravel()) for it in grp}
        assert len(gates) == 1
    assert sum(1 for it in items if (it.flags & M) and (it.flags & abi.ITEMFLAG_NESTED_MEDIUM)) >= 1
    assert not any((it.flags & M) for k, it in enumerate(items) if not any(b <= k < e for b, e in zip(begins, ends)))


def test_lists_with_media_mirror_equals_f64_oracle(host, orc64):
    nx, ny = 48, 32
    wh, wo = world_media_in_lists_in_bvh(host), world_media_in_lists_in_bvh(orc64)
    ch, co = camera(host, nx, ny), camera(orc64, nx, ny)
    for row in (8, 12, 16, 20, 24):
        ref = orc64.render(co, wo, nx, ny, 1, seed=42, rows=(row, row + 1))
        for i in range(nx):
            c = host.color_sample(ch, wh, nx, ny, i, ny - 1 - row, 0, seed=42)
            assert np.array_equal(c, ref["mean"][row, i]), (row, i)
    orc64.free_all()


def test_nested_media_lower_with_the_inner_density_behind_the_chain(host):
    a = host.lower(world_nested_media(host)).arrays()
    nested = [it for it in a["items"] if it.flags & abi.ITEMFLAG_NESTED_MEDIUM]
    assert len(nested) >= 6 and all(it.flags & abi.ITEMFLAG_MEDIUM for it in nested)
    seen = set()
    for it in nested:
        at = it.xform_first + it.xform_count + (2 if (it.flags & abi.ITEMFLAG_DEFERRED) and it.kind == abi.ITEM_BVH else 0)
        rec = a["xforms"][at]
        assert rec.kind == abi.XF_INNER_MEDIUM and rec.x < 0.0
        seen.add((bool(it.flags & abi.ITEMFLAG_DEFERRED), it.kind))
        assert np.float32(rec.x) != np.float32(it.neg_inv_density)  # the item's own density is the OUTER medium's
    assert seen == {(False, abi.ITEM_LIST), (False, abi.ITEM_BVH), (True, abi.ITEM_LIST), (True, abi.ITEM_BVH)}
    # refused: wrappers between the two media, and a third level
    glass = host.Dielectric(1.5)
    tex = host.SolidTexture(0.5, 0.5, 0.5)
    for bad in (host.ConstantMedium(host.Traslate(host.ConstantMedium(host.Sphere((0.0, 0.0, 0.0), 1.0, glass), 1.0, tex), (1.0, 0.0, 0.0)), 1.0, tex),
                host.ConstantMedium(host.ConstantMedium(host.ConstantMedium(host.Sphere((0.0, 0.0, 0.0), 1.0, glass), 1.0, tex), 1.0, tex), 1.0, tex)):
        w = host.HittableList()
        w.push(bad)
        with pytest.raises(Exception, match="not lowered"):
            host.lower(w)


def test_nested_media_mirror_equals_f64_oracle(host, orc64):
    nx, ny = 48, 32
    wh, wo = world_nested_media(host), world_nested_media(orc64)
    ch, co = camera(host, nx, ny), camera(orc64, nx, ny)
    for row in (8, 12, 16, 20, 24):
        ref = orc64.render(co, wo, nx, ny, 1, seed=42, rows=(row, row + 1))
        for i in range(nx):
            c = host.color_sample(ch, wh, nx, ny, i, ny - 1 - row, 0, seed=42)
            assert np.array_equal(c, ref["mean"][row, i]), (row, i)
    orc64.free_all()


def camera(api, nx, ny):
    return api.Camera((1.0, 3.0, 9.0), (0.0, 0.6, 0.5), (0.0, 1.0, 0.0), 42.0, nx / ny, 0.05, 9.0, 0.0, 1.0)


def test_lowering_emits_deferred_items_in_traversal_order(host):
    a = host.lower(world_media_in_bvh(host)).arrays()
    items = a["items"]
    dfr = [it for it in items if it.flags & abi.ITEMFLAG_DEFERRED]
    for it in dfr:
        assert it.flags & abi.ITEMFLAG_MEDIUM and it.kind == abi.ITEM_LIST and it.count == 1
        g = (it.flags >> abi.RTMI_ITEMFLAG_GATE_OUTER_SHIFT) & 15
        outer = (it.flags >> abi.RTMI_ITEMFLAG_MEDIUM_OUTER_SHIFT) & 15
        assert g <= outer <= it.xform_count
    bvh = [k for k, it in enumerate(items) if it.kind == abi.ITEM_BVH]
    assert len(bvh) == 3  # the BVH of media only has no BVH item
    tail = items[bvh[2] + 1:]  # case 4: the one medium of the inner BVH, once per side of the one-element node above it
    assert len(tail) % 2 == 0 and len(tail) >= 2 and all(it.flags & abi.ITEMFLAG_DEFERRED for it in tail)
    assert len({it.medium_material for it in tail}) == 1
    items = items[:bvh[2]]
    bvh = bvh[:2]
    for k in bvh:
        assert items[k].flags & abi.ITEMFLAG_SAVE_T0 and items[k + 1].flags & abi.ITEMFLAG_DEFERRED
    # first BVH: three media; BVHNode::new (bvh.rs:39-66) puts the odd one of a split into a node of its own, left and
    # right the same object (bvh.rs:44-45), which evaluates it twice: three or more deferred items, equal ones adjacent
    group1 = items[bvh[0] + 1:bvh[1]]
    n1 = 0
    while n1 < len(group1) and (group1[n1].flags & abi.ITEMFLAG_DEFERRED) and not (group1[n1].flags & abi.ITEMFLAG_SAVE_T0):
        n1 += 1
    assert n1 >= 3 and len({it.medium_material for it in group1[:n1]}) == 3
    # the one-element node: two consecutive deferred items of the same medium, the first remembers T0 itself
    k = bvh[0] + 1 + n1
    assert items[k].flags & abi.ITEMFLAG_SAVE_T0 and items[k].flags & abi.ITEMFLAG_DEFERRED
    assert items[k + 1].flags & abi.ITEMFLAG_DEFERRED and not (items[k + 1].flags & abi.ITEMFLAG_SAVE_T0)
    assert items[k].medium_material == items[k + 1].medium_material and items[k].first != items[k + 1].first  # two boundaries, two draws
    assert k + 2 == bvh[1]
    # the third group sits inside Traslate(Rotate(..)): two enclosing transforms in front of every deferred chain
    last = items[bvh[1] + 1:]
    assert len(last) >= 3 and all((it.flags & abi.ITEMFLAG_DEFERRED) and ((it.flags >> abi.RTMI_ITEMFLAG_GATE_OUTER_SHIFT) & 15) == 2 and it.xform_count >= 2
                                  for it in last)
    assert len({it.medium_material for it in last}) == 3


def test_mirror_equals_f64_oracle(host, orc64):
    nx, ny = 48, 32
    wh, wo = world_media_in_bvh(host), world_media_in_bvh(orc64)
    ch, co = camera(host, nx, ny), camera(orc64, nx, ny)
    for row in (6, 12, 16, 20, 26):
        ref = orc64.render(co, wo, nx, ny, 2, seed=42, rows=(row, row + 1))
        for i in range(nx):
            c = sum(host.color_sample(ch, wh, nx, ny, i, ny - 1 - row, s, seed=42) for s in range(2)) / 2.0
            assert np.allclose(c, ref["mean"][row, i], rtol=0, atol=1e-15), (row, i)
    orc64.free_all()


def test_instanced_subtrees_lower_as_deferred_bvh_items(host):
    a = host.lower(world_instanced_subtrees(host)).arrays()
    items = a["items"]
    bvh = [it for it in items if it.kind == abi.ITEM_BVH]
    dfr = [it for it in bvh if it.flags & abi.ITEMFLAG_DEFERRED]
    # the enclosing tree + red cluster + deep (+ its own yellow cluster) + the cyan cluster twice — and twice again whatever
    # BVHNode::new left alone in a slice of one (bvh.rs:44-45: evaluated on both sides)
    assert len(bvh) == len(dfr) + 1 and len(dfr) >= 6 and sum(1 for it in items if it.flags & abi.ITEMFLAG_SAVE_T0) == 1
    assert items[1].kind == abi.ITEM_BVH and items[1].flags & abi.ITEMFLAG_SAVE_T0 and not (items[1].flags & abi.ITEMFLAG_DEFERRED)
    assert sum(1 for it in dfr if it.flags & abi.ITEMFLAG_MEDIUM) >= 1  # the medium around a BVHNode boundary
    for it in dfr:
        g = (it.flags >> abi.RTMI_ITEMFLAG_GATE_OUTER_SHIFT) & 15
        assert 2 <= g <= it.xform_count  # the item's Traslate(Rotate(..)) in front of every chain
        kinds = [a["xforms"][it.xform_first + k].kind for k in range(it.xform_count + 2)]
        assert kinds[-2:] == [abi.XF_GATE_MIN, abi.XF_GATE_MAX] and all(k <= abi.XF_ROTATE_Z for k in kinds[:-2])
    assert sum(1 for it in dfr if it.flags & abi.ITEMFLAG_FLIP) >= 1  # the flipped one (and the subtree nested in it)
    assert sum(1 for it in items if (it.flags & abi.ITEMFLAG_DEFERRED) and (it.flags & abi.ITEMFLAG_MEDIUM) and it.kind == abi.ITEM_LIST) >= 1


def test_instanced_subtrees_mirror_equals_f64_oracle(host, orc64):
    nx, ny = 48, 32
    wh, wo = world_instanced_subtrees(host), world_instanced_subtrees(orc64)
    ch, co = camera(host, nx, ny), camera(orc64, nx, ny)
    for row in (8, 14, 18, 22):
        ref = orc64.render(co, wo, nx, ny, 1, seed=42, rows=(row, row + 1))
        for i in range(nx):
            c = host.color_sample(ch, wh, nx, ny, i, ny - 1 - row, 0, seed=42)
            assert np.array_equal(c, ref["mean"][row, i]), (row, i)
    orc64.free_all()


@pytest.mark.gpu
@pytest.mark.parametrize("build", [world_media_in_bvh, world_instanced_subtrees, world_nested_media, world_media_in_lists_in_bvh],
                         ids=["media", "instanced_subtrees", "nested_media", "media_in_lists"])
def test_every_kernel_equals_the_fp32_oracle(host, orc32, build):
    world_media_in_bvh = build  # noqa: F811 — the same comparison for both worlds
    nx, ny, ns = 120, 80, 24
    sc = host.lower(world_media_in_bvh(host))
    ref = orc32.render(camera(orc32, nx, ny), world_media_in_bvh(orc32), nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    assert float(ref["linear"].mean()) > 0.05
    cam = camera(host, nx, ny)
    for flags in (0, abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_REF_TREE, abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_BLOCK_COOP,
                  abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_FAST_CULL | (1 << 11)):
        got = sc.render(cam, nx, ny, ns, seed=42, flags=flags, sig=True)
        assert np.array_equal(got["sig"], ref["sig"]), flags
        assert np.array_equal(got["linear"], ref["linear"]), flags
        assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"]), flags
    orc32.free_all()
