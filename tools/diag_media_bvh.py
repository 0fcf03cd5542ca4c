#!/usr/bin/env python3
"""Pinpoint a media-in-BVH parity failure: small scenes, device (exact kernel) vs fp32 oracle (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle.oracle import Oracle, ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import Host, abi

host = Host(); orc = Oracle("f32")
nx, ny, ns = 60, 40, 8


def base(api):
    w = api.HittableList()
    w.push(api.Rect(api.PLANE_ZX, -8.0, -8.0, 8.0, 8.0, -1.5, api.Lambertian(api.SolidTexture(0.6, 0.6, 0.6))))
    w.push(api.Sphere((0.0, 9.0, 2.0), 3.0, api.DiffuseLight(api.SolidTexture(5.0, 5.0, 5.0))))
    return w


def S(api, c, r=0.7, col=(0.8, 0.3, 0.3)):
    return api.Sphere(c, r, api.Lambertian(api.SolidTexture(*col)))


def M(api, c, r=0.8, dens=1.5, col=(0.2, 0.9, 0.2)):
    return api.ConstantMedium(api.Sphere(c, r, api.Dielectric(1.5)), dens, api.SolidTexture(*col))


def cases(api):
    out = {}
    def add(name, fn):
        api.seed_scene_rng(5)
        w = base(api); fn(w); out[name] = w
    add("A two-element {S, M}", lambda w: w.push(api.BVHNode([S(api, (-1, 0, 0)), M(api, (1, 0, 0))], 0.0, 1.0)))
    add("B three {S, S, M}", lambda w: w.push(api.BVHNode([S(api, (-2, 0, 0)), S(api, (0, 0, 0)), M(api, (2, 0, 0))], 0.0, 1.0)))
    add("C nested {BVH{M, M}, S}", lambda w: w.push(api.BVHNode([api.BVHNode([M(api, (-1, 0, 0)), M(api, (1, 0.5, 0), col=(0.9, 0.2, 0.2))], 0.0, 1.0), S(api, (3, 0, 0))], 0.0, 1.0)))
    add("D rotated instanced prim (infinite boxes) {Rot(S), S, M}", lambda w: w.push(api.BVHNode([api.Rotate(api.AXIS_Y, S(api, (-2, 0, 0)), 30.0), S(api, (0.3, 0, 0)), M(api, (2, 0, 0))], 0.0, 1.0)))
    add("E flipped inner {Flip(BVH{S, M}), S}", lambda w: w.push(api.BVHNode([api.FlipNormals(api.BVHNode([S(api, (-1, 0, 0)), M(api, (0.6, 0, 0))], 0.0, 1.0)), S(api, (3, 0, 0))], 0.0, 1.0)))
    add("F Traslate(M) child {S, S, Tr(M)}", lambda w: w.push(api.BVHNode([S(api, (-2, 0, 0)), S(api, (0, 0, 0)), api.Traslate(M(api, (1.2, 0, 0)), (0.8, -0.6, 0.4))], 0.0, 1.0)))
    add("G moving {MS, S, M}", lambda w: w.push(api.BVHNode([api.MovingSphere((-2, 0, 0), (-1.7, 0.3, 0), 0.0, 1.0, 0.5, api.Lambertian(api.SolidTexture(0.7, 0.7, 0.2))), S(api, (0, 0, 0)), M(api, (2, 0, 0))], 0.0, 1.0)))
    add("H cube medium single-element beside internal {S,S,S,Mcube}", lambda w: w.push(api.BVHNode([S(api, (-3, 0, 0)), S(api, (-1, 0, 0)), S(api, (1, 0, 0)), api.ConstantMedium(api.Cube((2, -1, -0.5), (3.2, 0.3, 0.6), api.Dielectric(1.5)), 0.2, api.SolidTexture(0.3, 0.3, 0.9))], 0.0, 1.0)))
    add("I five {S,M,S,M,S}", lambda w: w.push(api.BVHNode([S(api, (-3, 0, 0)), M(api, (-1.5, 0, 0)), S(api, (0, 0, 0)), M(api, (1.5, 0, 0), col=(0.9, 0.2, 0.9)), S(api, (3, 0, 0))], 0.0, 1.0)))
    add("J sphere INSIDE the medium {M big, S small}", lambda w: w.push(api.BVHNode([M(api, (0.5, 0.3, 0), r=1.6, dens=0.8), S(api, (0.7, 0.2, 0.1), r=0.5)], 0.0, 1.0)))
    add("K metal sphere inside dense medium + another", lambda w: w.push(api.BVHNode([M(api, (0.5, 0.3, 0), r=1.6, dens=2.5), api.Sphere((0.7, 0.2, 0.1), 0.5, api.Metal(api.SolidTexture(0.9, 0.9, 0.9), 0.0)), S(api, (-2.5, 0, 0))], 0.0, 1.0)))
    add("L medium inside Traslate, sphere inside it", lambda w: w.push(api.BVHNode([api.Traslate(M(api, (0.0, 0.0, 0), r=1.2, dens=2.5), (0.8, -0.65, 0.44)), S(api, (0.9, -0.5, 0.3), r=0.35), S(api, (-2.5, 0, 0))], 0.0, 1.0)))
    add("M two media overlapping + sphere", lambda w: w.push(api.BVHNode([M(api, (0.0, 0.0, 0), r=1.2, dens=1.0), M(api, (0.8, 0.2, 0), r=1.0, dens=3.0, col=(0.9, 0.2, 0.2)), S(api, (0.4, 0.0, 0.2), r=0.3)], 0.0, 1.0)))
    return out


ch = cases(host); co = cases(orc)
for name in ch:
    camh = host.Camera((1.0, 2.5, 9.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.0, 9.0, 0.0, 1.0)
    camo = orc.Camera((1.0, 2.5, 9.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.0, 9.0, 0.0, 1.0)
    sc = host.lower(ch[name])
    ref = orc.render(camo, co[name], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    res = []
    for flags in (0, 1):
        got = sc.render(camh, nx, ny, ns, seed=42, flags=flags, sig=True)
        res.append(int((got["sig"] != ref["sig"]).sum()))
    items = sc.arrays()["items"]
    print("%-62s mismatches exact %4d coop %4d   items %s" % (name, res[0], res[1], [hex(it.flags) for it in items]))
