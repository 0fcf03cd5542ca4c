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
lamb = lambda a: a.Lambertian(a.SolidTexture(0.6, 0.5, 0.4))
def run(name, build, md=50):
    out = []
    for api in (host, orc):
        api.seed_scene_rng(1)
        w = api.HittableList()
        for h in build(api): w.push(h)
        out.append((cam(api), w))
    sc = host.lower(out[0][1])
    got = sc.render(out[0][0], nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_SKY, sig=True, max_depth=md)
    ref = orc.render(out[1][0], out[1][1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM | SKY, max_depth=md)
    d = np.abs(got["linear"] - ref["linear"]).max(axis=2)
    print("%-32s depth %2d: pixels differing %4d of %d (sphere pixels %d), max diff %.4f, sig mismatches %d" % (name, md, int((d > 0).sum()), nx * ny, int((ref["sig"] != 0).sum()), float(d.max()), int((got["sig"] != ref["sig"]).sum())))
    host.free_all(); orc.free_all()
for md in (1, 2, 50):
    run("plain sphere", lambda a: [a.Sphere((0.5, 0.2, -0.3), 1.5, lamb(a))], md)
    for ax in ("AXIS_X", "AXIS_Y", "AXIS_Z"):
        run("rot %s sphere" % ax, lambda a: [a.Rotate(getattr(a, ax), a.Sphere((0.5, 0.2, -0.3), 1.5, lamb(a)), 33.0)], md)
    run("rot Z 180", lambda a: [a.Rotate(a.AXIS_Z, a.Sphere((0.5, 0.2, -0.3), 1.5, lamb(a)), 180.0)], md)
    run("rot Z two spheres", lambda a: [a.Rotate(a.AXIS_Z, a.Sphere((0.5, 0.2, -0.3), 1.5, lamb(a)), 33.0), a.Sphere((0.0, 9.0, 0.0), 3.0, lamb(a))], md)
