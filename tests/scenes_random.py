"""Randomly composed test scenes (NOT reference scenes): every primitive, wrapper, material and texture of
the hot path in arbitrary combinations the lowering supports — list items that are single primitives,
lists or BVHs, under FlipNormals / Traslate / Rotate chains, ConstantMedium around (transformed)
boundaries, emitters, metal with fuzz > 1, dielectrics, coincident and touching surfaces.  Built with
the backend-agnostic `api` (C++ host mirror or oracle), so both sides construct the same world."""
import numpy as np


def _texture(api, rng, depth=0):
    k = rng.integers(0, 5 if depth == 0 else 2)
    if k == 0 or k == 1:
        return api.SolidTexture(*rng.uniform(0.05, 0.95, 3))
    if k == 2:
        return api.CheckerTexture(_texture(api, rng, 1), _texture(api, rng, 1))
    if k == 3:
        return api.NoiseTexture(float(rng.uniform(0.05, 4.0)))
    nx, ny = int(rng.integers(2, 17)), int(rng.integers(2, 9))
    return api.ImageTexture(rng.integers(0, 256, (ny, nx, 3), dtype=np.uint8), nx, ny)


def _material(api, rng):
    k = rng.integers(0, 10)
    if k < 4:
        return api.Lambertian(_texture(api, rng))
    if k < 6:
        return api.Metal(_texture(api, rng), float(rng.choice([0.0, 0.1, 0.5, 1.0, 3.0])))
    if k < 8:
        return api.Dielectric(float(rng.choice([1.0, 1.3, 1.5, 2.4, 0.7])))
    return api.DiffuseLight(api.SolidTexture(*rng.uniform(0.5, 6.0, 3)))


def _prim(api, rng, extent=3.0, allow_moving=True):
    k = rng.integers(0, 6 if allow_moving else 5)
    m = _material(api, rng)
    c = rng.uniform(-extent, extent, 3)
    if k <= 1:
        r = float(rng.uniform(0.2, 1.2))
        return api.Sphere(c, -r if rng.random() < 0.1 else r, m)  # negative radius: inward normals (hollow glass)
    if k == 2:
        a, b = rng.uniform(-extent, extent, 2), rng.uniform(-extent, extent, 2)
        lo, hi = np.minimum(a, b), np.maximum(a, b) + 0.3
        if rng.random() < 0.05:
            lo, hi = hi, lo  # inverted extents: can never be hit (like the reference's final_scene light)
        plane = [api.PLANE_YZ, api.PLANE_ZX, api.PLANE_XY][int(rng.integers(0, 3))]
        return api.Rect(plane, lo[0], lo[1], hi[0], hi[1], float(rng.uniform(-extent, extent)), m)
    if k == 3 or k == 4:
        a = rng.uniform(-extent, extent, 3)
        return api.Cube(a, a + rng.uniform(0.3, 1.5, 3), m)
    return api.MovingSphere(c, c + rng.uniform(-0.6, 0.6, 3), 0.0, 1.0, float(rng.uniform(0.2, 0.8)), m)


def _wrap(api, rng, h, allow_flip=True):
    for _ in range(int(rng.integers(0, 4))):
        k = rng.integers(0, 3)
        if k == 0:
            h = api.Traslate(h, rng.uniform(-1.5, 1.5, 3))
        elif k == 1:
            axis = [api.AXIS_X, api.AXIS_Y, api.AXIS_Z][int(rng.integers(0, 3))]
            h = api.Rotate(axis, h, float(rng.uniform(-80.0, 80.0)))
        elif allow_flip:
            h = api.FlipNormals(h)
    return h


def random_scene(api, seed, only=None, instanced=False):
    """only: indices of the top-level items to keep (debugging aid; construction is identical either way).
    instanced: members of nested lists and children of BVHNodes are themselves wrapped in Traslate / Rotate /
    FlipNormals chains with probability 0.4 (Traslate<H> / Rotate<H> are generic: traslate.rs:6-9, rotate.rs:21-28
    under bvh.rs:11-12); the wrap decisions draw from their own generator, so instanced=False builds the same scenes as
    before."""
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed + 77_777)

    def inst(p):
        if instanced and rng2.random() < 0.4:
            q = p
            while q is p:  # at least one wrapper
                q = _wrap(api, rng2, p)
            return q
        return p

    api.seed_scene_rng(seed)
    real = api.HittableList()

    class _Sel:  # pushes only the selected top-level items, counts all
        n = 0

        def push(self, h):
            if only is None or self.n in only:
                real.push(h)
            self.n += 1

    world = _Sel()
    # something to stand on and something to see by, so that paths are long and carry radiance
    world.push(api.Rect(api.PLANE_ZX, -6.0, -6.0, 6.0, 6.0, -3.2, api.Lambertian(_texture(api, rng))))
    world.push(api.Sphere((0.0, 7.0, 0.0), 2.5, api.DiffuseLight(api.SolidTexture(4.0, 4.0, 4.0))))
    for _ in range(int(rng.integers(3, 9))):
        k = rng.integers(0, 10)
        if k < 4:  # a single primitive, possibly wrapped
            world.push(_wrap(api, rng, _prim(api, rng)))
        elif k < 5:  # a nested list
            inner = api.HittableList()
            for _ in range(int(rng.integers(2, 5))):
                p = inst(_prim(api, rng))
                inner.push(api.FlipNormals(p) if rng.random() < 0.2 else p)
            world.push(_wrap(api, rng, inner))
        elif k < 8:  # a BVH (moving spheres make every static sphere in it a moving one on the device)
            n = int(rng.integers(1, 40))
            moving = rng.random() < 0.4
            objs = []
            for _ in range(n):
                p = inst(_prim(api, rng, extent=2.5, allow_moving=moving))
                objs.append(api.FlipNormals(p) if rng.random() < 0.1 else p)
            has_media = False
            if instanced and rng2.random() < 0.35:
                has_media = True
                # ConstantMedium as a CHILD of the BVHNode (bvh.rs:11-12 takes any Hittable; tests/test_media_in_bvh.py): one to
                # three media, around plain or transformed boundaries, sometimes themselves inside a Traslate — with n == 1
                # and the primitive replaced, a BVH of media only
                nm = int(rng2.integers(1, 4))
                meds = []
                for _ in range(nm):
                    bq = api.Sphere(rng2.uniform(-2, 2, 3), float(rng2.uniform(0.5, 1.4)), api.Dielectric(1.5)) if rng2.random() < 0.6 \
                        else api.Cube(rng2.uniform(-2.0, 0, 3), rng2.uniform(0.4, 1.8, 3), api.Dielectric(1.5))
                    if rng2.random() < 0.4:
                        bq = _wrap(api, rng2, bq, allow_flip=False)
                    elif rng2.random() < 0.3:  # a boundary that is itself a BVHNode (two overlapping shapes)
                        c2 = rng2.uniform(-1.5, 1.5, 3)
                        bq = api.BVHNode([api.Sphere(c2, float(rng2.uniform(0.5, 1.0)), api.Dielectric(1.5)),
                                          api.Cube(c2 - rng2.uniform(0.2, 0.9, 3), c2 + rng2.uniform(0.2, 0.9, 3), api.Dielectric(1.5))], 0.0, 1.0)
                    if rng2.random() < 0.25:  # a medium as the boundary of the medium (medium.rs:11-15 is generic): three draws
                        bq = api.ConstantMedium(bq, float(rng2.choice([0.3, 1.0, 3.0])), api.SolidTexture(0.5, 0.5, 0.5))
                    md = api.ConstantMedium(bq, float(rng2.choice([0.05, 0.4, 1.5, 5.0])), api.SolidTexture(*rng2.uniform(0.1, 0.95, 3)))
                    if rng2.random() < 0.3:
                        md = api.Traslate(md, rng2.uniform(-1.0, 1.0, 3))
                    meds.append(md)
                if rng2.random() < 0.35:
                    # a HittableList WITH MEDIA as one object of the BVH (tests/test_media_in_bvh.py, world_media_in_lists_in_bvh):
                    # the first medium together with up to three of the primitives, before and behind it in the scan,
                    # sometimes inside a nested (flipped) list
                    grp = api.HittableList()
                    take = [objs.pop(int(rng2.integers(0, len(objs)))) for _ in range(min(len(objs) - 1, int(rng2.integers(0, 4))))]
                    cut = int(rng2.integers(0, len(take) + 1))
                    for q in take[:cut]:
                        grp.push(q)
                    if rng2.random() < 0.3:
                        sub = api.HittableList()
                        sub.push(meds[0])
                        if take[cut:]:
                            sub.push(take[cut])
                        grp.push(api.FlipNormals(sub) if rng2.random() < 0.5 else sub)
                    else:
                        grp.push(meds[0])
                    for q in take[cut:]:
                        grp.push(q)
                    meds[0] = api.FlipNormals(grp) if rng2.random() < 0.2 else grp
                if n == 1 and rng2.random() < 0.5:
                    objs = meds  # nothing but media
                else:
                    for md in meds:
                        objs.insert(int(rng2.integers(0, len(objs) + 1)), md)
                n = len(objs)
            if instanced and n >= 3 and not has_media and rng2.random() < 0.4:
                # a HittableList as ONE object of the BVH (bvh.rs:11-12 takes any Hittable): two to five of the objects,
                # sometimes one of them twice (an exact tie inside the scan), sometimes with a nested list, sometimes flipped
                k0 = int(rng2.integers(0, n - 1))
                k1 = min(n, k0 + int(rng2.integers(2, 6)))
                grp = api.HittableList()
                for q in objs[k0:k1]:
                    grp.push(q)
                if rng2.random() < 0.5:
                    grp.push(objs[k0])
                if rng2.random() < 0.3:
                    sub = api.HittableList()
                    sub.push(objs[k1 - 1])
                    sub.push(objs[k0])
                    grp.push(api.FlipNormals(sub) if rng2.random() < 0.5 else sub)
                objs = objs[:k0] + [api.FlipNormals(grp) if rng2.random() < 0.2 else grp] + objs[k1:]
            if n > 6 and rng.random() < 0.3:  # a BVH built earlier as one of the objects of this one
                half = len(objs) // 2
                inner = api.BVHNode(objs[:half], 0.0, 1.0)
                if instanced and rng2.random() < 0.5:
                    inner = api.FlipNormals(inner)  # hittable.rs:67-88 around a subtree: every normal below is negated
                if instanced and rng2.random() < 0.4:
                    # an INSTANCED subtree: Traslate / Rotate (/ FlipNormals) around the inner BVHNode (traslate.rs:6-9 and
                    # rotate.rs:21-28 wrap any Hittable) — a deferred BVH item behind the enclosing one (tests/test_media_in_bvh.py)
                    inner = _wrap(api, rng2, api.Traslate(inner, rng2.uniform(-1.0, 1.0, 3)))
                objs = [inner] + objs[half:]
            world.push(_wrap(api, rng, api.BVHNode(objs, 0.0, 1.0)))
        else:  # a participating medium inside a (transformed) boundary; FlipNormals outside only
            b = api.Sphere(rng.uniform(-2, 2, 3), float(rng.uniform(0.8, 2.0)), api.Dielectric(1.5)) if rng.random() < 0.5 \
                else api.Cube(rng.uniform(-2.5, 0, 3), rng.uniform(0.5, 2.5, 3), api.Dielectric(1.5))
            b = _wrap(api, rng, b, allow_flip=False)
            if instanced and rng2.random() < 0.3:
                # the boundary as a ONE-member list whose member carries its own transform chain:
                # ConstantMedium(HittableList[Traslate(Rotate(Sphere))]) — the device must not take its fused
                # sphere-boundary query here (r03 advisor finding: that query read the untransformed centre)
                lst = api.HittableList()
                q = api.Sphere(rng2.uniform(-2, 2, 3), float(rng2.uniform(0.8, 2.0)), api.Dielectric(1.5))
                lst.push(_wrap(api, rng2, api.Traslate(q, rng2.uniform(-1.0, 1.0, 3)), allow_flip=False))
                b = lst
            if instanced and rng2.random() < 0.25:  # nested media (tests/test_media_in_bvh.py, world_nested_media)
                b = api.ConstantMedium(b, float(rng2.choice([0.2, 1.0, 4.0])), api.SolidTexture(0.5, 0.5, 0.5))
            med = api.ConstantMedium(b, float(rng.choice([0.0, 0.01, 0.2, 1.0, 5.0])), _texture(api, rng))
            if instanced and rng2.random() < 0.5:
                med = _wrap(api, rng2, med)  # Traslate / Rotate / FlipNormals AROUND the medium (traslate.rs:6-9 is generic)
            world.push(api.FlipNormals(med) if rng.random() < 0.1 else med)
    return real


def random_camera(api, seed, nx, ny):
    rng = np.random.default_rng(seed + 10_000)
    d = rng.normal(size=3)
    d[1] = abs(d[1]) * 0.5
    look_from = d / np.linalg.norm(d) * rng.uniform(7.0, 12.0)
    aperture = float(rng.choice([0.0, 0.0, 0.2]))
    return api.Camera(look_from, rng.uniform(-0.5, 0.5, 3), (0.0, 1.0, 0.0), float(rng.uniform(35.0, 70.0)), nx / ny,
                      aperture, float(np.linalg.norm(look_from)), 0.0, 1.0)


def build(api, seed, nx, ny, only=None, instanced=False):
    return random_camera(api, seed, nx, ny), random_scene(api, seed, only, instanced)
