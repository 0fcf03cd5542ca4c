#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle, ARITH_DEVICE, THROUGHPUT_FORM, SKY
from raytracing_rust_amd import Host, abi
nx, ny, ns = 40, 24, 4
host = Host(); orc = Oracle("f32")
def cam(api):
    return api.Camera((6.0, 3.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, nx / ny, 0.0, 10.0, 0.0, 1.0)
def run(name, build, md=1, flags=0):
    out = []
    for api in (host, orc):
        api.seed_scene_rng(1)
        w = api.HittableList()
        for h in build(api): w.push(h)
        out.append((cam(api), w))
    sc = host.lower(out[0][1])
    a = sc.arrays()
    got = sc.render(out[0][0], nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_SKY | flags, sig=True, max_depth=md)
    ref = orc.render(out[1][0], out[1][1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM | SKY, max_depth=md)
    d = np.abs(got["linear"] - ref["linear"]).max(axis=2)
    print("%-36s pixels differing %4d (object pixels %d) max diff %.4f | items %s xforms %s" % (name, int((d > 0).sum()), int((ref["sig"] != 0).sum()), float(d.max()),
          [(it.kind, it.first, it.count, it.flags, it.xform_first, it.xform_count) for it in a["items"]], [(x.kind, round(float(x.x), 4), round(float(x.y), 4)) for x in a["xforms"]]))
    host.free_all(); orc.free_all()
chk = lambda a: a.CheckerTexture(a.SolidTexture(0.1, 0.1, 0.1), a.SolidTexture(0.9, 0.9, 0.9))
for ax in ("AXIS_X", "AXIS_Z"):
    R = lambda a, h: a.Rotate(getattr(a, ax), h, 33.0)
    run(ax + " emitter checker sphere", lambda a: [R(a, a.Sphere((0.5, 0.2, -0.3), 1.5, a.DiffuseLight(chk(a))))])
    run(ax + " metal0 sphere", lambda a: [R(a, a.Sphere((0.5, 0.2, -0.3), 1.5, a.Metal(a.SolidTexture(0.8, 0.8, 0.8), 0.0)))])
    run(ax + " lamb sphere", lambda a: [R(a, a.Sphere((0.5, 0.2, -0.3), 1.5, a.Lambertian(a.SolidTexture(0.8, 0.8, 0.8))))])
    run(ax + " lamb sphere perlane", lambda a: [R(a, a.Sphere((0.5, 0.2, -0.3), 1.5, a.Lambertian(a.SolidTexture(0.8, 0.8, 0.8))))], flags=abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL)
    run(ax + " lamb sphere async", lambda a: [R(a, a.Sphere((0.5, 0.2, -0.3), 1.5, a.Lambertian(a.SolidTexture(0.8, 0.8, 0.8))))], flags=abi.RTMI_FLAG_ASYNC)
    run(ax + " metal0 cube", lambda a: [R(a, a.Cube((-1.0, -1.0, -1.0), (1.0, 0.5, 1.2), a.Metal(a.SolidTexture(0.8, 0.8, 0.8), 0.0)))])
    run(ax + " metal0 rect ZX", lambda a: [R(a, a.Rect(a.PLANE_ZX, -2, -2, 2, 2, 0.3, a.Metal(a.SolidTexture(0.8, 0.8, 0.8), 0.0)))])
    run(ax + " two wrappers T(R(sphere))", lambda a: [a.Traslate(R(a, a.Sphere((0.5, 0.2, -0.3), 1.5, a.Metal(a.SolidTexture(0.8, 0.8, 0.8), 0.0))), (0.1, 0.2, 0.3))])
