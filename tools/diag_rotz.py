#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle, ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import Host, abi
nx, ny, ns = 40, 24, 6
host = Host(); orc = Oracle("f32")
def cam(api):
    return api.Camera((6.0, 3.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, nx / ny, 0.0, 10.0, 0.0, 1.0)
def run(name, build, max_depth=50, flags=0):
    out = []
    for api in (host, orc):
        api.seed_scene_rng(1)
        w = api.HittableList()
        for h in build(api):
            w.push(h)
        out.append((cam(api), w))
    sc = host.lower(out[0][1])
    got = sc.render(out[0][0], nx, ny, ns, seed=42, flags=flags, sig=True, max_depth=max_depth)
    ref = orc.render(out[1][0], out[1][1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM, max_depth=max_depth)
    bad = np.argwhere(got["sig"] != ref["sig"])
    print("%-40s depth %2d sig mismatches %4d radiance mismatches %4d" % (name, max_depth, len(bad), int((got["linear"] != ref["linear"]).sum())), bad[:3].tolist())
    a = sc.arrays()
    if name.endswith("!"):
        for x in a["xforms"]:
            print("   xform kind", x.kind, "x(sin)", repr(float(x.x)), "y(cos)", repr(float(x.y)), "z", float(x.z))
    host.free_all(); orc.free_all()
lamb = lambda a: a.Lambertian(a.SolidTexture(0.6, 0.5, 0.4))
light = lambda a: a.DiffuseLight(a.SolidTexture(1.0, 2.0, 3.0))
for ang in (33.0, 90.0, -10.0):
    for md in (0, 1, 50):
        run("rotZ %.0f sphere light!" % ang, lambda a: [a.Rotate(a.AXIS_Z, a.Sphere((0.5, 0.2, -0.3), 1.5, light(a)), ang)], md)
for md in (0, 1):
    run("rotZ 33 sphere lamb", lambda a: [a.Rotate(a.AXIS_Z, a.Sphere((0.5, 0.2, -0.3), 1.5, lamb(a)), 33.0)], md)
    run("rotX 33 sphere lamb", lambda a: [a.Rotate(a.AXIS_X, a.Sphere((0.5, 0.2, -0.3), 1.5, lamb(a)), 33.0)], md)
    run("rotZ 33 rectXY light", lambda a: [a.Rotate(a.AXIS_Z, a.Rect(a.PLANE_XY, -2, -2, 2, 2, 0.3, light(a)), 33.0)], md)
    run("rotZ 33 rectZX light", lambda a: [a.Rotate(a.AXIS_Z, a.Rect(a.PLANE_ZX, -2, -2, 2, 2, 0.3, light(a)), 33.0)], md)
