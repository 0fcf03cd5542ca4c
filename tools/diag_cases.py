#!/usr/bin/env python3
"""Matrix of tiny hand-made scenes, device vs fp32 oracle (GPU box): which wrapper/material combination differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle, ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import Host, abi

nx, ny, ns = 40, 24, 6
host = Host(); orc = Oracle("f32")

def cam(api):
    return api.Camera((6.0, 3.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, nx / ny, 0.0, 10.0, 0.0, 1.0)

def run(name, build, flags=0):
    out = []
    for api in (host, orc):
        api.seed_scene_rng(1)
        w = api.HittableList()
        w.push(api.Sphere((0.0, 9.0, 0.0), 3.0, api.DiffuseLight(api.SolidTexture(4.0, 4.0, 4.0))))
        for h in build(api):
            w.push(h)
        out.append((cam(api), w))
    sc = host.lower(out[0][1])
    got = sc.render(out[0][0], nx, ny, ns, seed=42, flags=flags, sig=True)
    ref = orc.render(out[1][0], out[1][1], nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    nsig = int((got["sig"] != ref["sig"]).sum()); nlin = int((got["linear"] != ref["linear"]).sum())
    print("%-44s sig mismatches %4d  radiance mismatches %4d  mean %.4f" % (name, nsig, nlin, float(ref["linear"].mean())))
    host.free_all(); orc.free_all()

mats = {"lamb": lambda a: a.Lambertian(a.SolidTexture(0.6, 0.5, 0.4)), "metal": lambda a: a.Metal(a.SolidTexture(0.8, 0.8, 0.8), 0.1),
        "glass": lambda a: a.Dielectric(1.5), "checker": lambda a: a.Lambertian(a.CheckerTexture(a.SolidTexture(0.1, 0.1, 0.1), a.SolidTexture(0.9, 0.9, 0.9))),
        "noise": lambda a: a.Lambertian(a.NoiseTexture(1.0))}
for mn, mf in mats.items():
    run("sphere %s" % mn, lambda a: [a.Sphere((0.5, 0.2, -0.3), 1.5, mf(a))])
    run("flip sphere %s" % mn, lambda a: [a.FlipNormals(a.Sphere((0.5, 0.2, -0.3), 1.5, mf(a)))])
for ax_name in ("AXIS_X", "AXIS_Y", "AXIS_Z"):
    for mn in ("lamb", "glass"):
        mf = mats[mn]
        run("rot %s sphere %s" % (ax_name, mn), lambda a: [a.Rotate(getattr(a, ax_name), a.Sphere((0.5, 0.2, -0.3), 1.5, mf(a)), 33.0)])
        run("rot %s cube %s" % (ax_name, mn), lambda a: [a.Rotate(getattr(a, ax_name), a.Cube((-1.0, -1.0, -1.0), (1.0, 0.5, 1.2), mf(a)), 33.0)])
run("rotZ rotY flip sphere lamb", lambda a: [a.FlipNormals(a.Rotate(a.AXIS_Z, a.Rotate(a.AXIS_Y, a.Sphere((0.5, 0.2, -0.3), 1.5, mats["lamb"](a)), 20.0), -40.0))])
run("traslate rotX cube glass", lambda a: [a.Traslate(a.Rotate(a.AXIS_X, a.Cube((-1.0, -1.0, -1.0), (1.0, 0.5, 1.2), mats["glass"](a)), 33.0), (0.3, 0.2, 0.1))])
run("rect planes", lambda a: [a.Rect(a.PLANE_YZ, -1, -1, 1, 1, 0.5, mats["lamb"](a)), a.Rect(a.PLANE_ZX, -2, -2, 2, 2, -1.0, mats["checker"](a)), a.Rect(a.PLANE_XY, -1, -1, 1, 1, -0.5, mats["metal"](a))])
run("flip rect", lambda a: [a.FlipNormals(a.Rect(a.PLANE_ZX, -2, -2, 2, 2, -1.0, mats["lamb"](a)))])
run("moving sphere", lambda a: [a.MovingSphere((0.0, 0.0, 0.0), (0.4, 0.3, -0.2), 0.0, 1.0, 1.0, mats["lamb"](a))])
run("medium sphere", lambda a: [a.ConstantMedium(a.Sphere((0.0, 0.0, 0.0), 1.5, a.Dielectric(1.5)), 1.0, a.SolidTexture(0.8, 0.8, 0.8))])
run("medium rot cube", lambda a: [a.ConstantMedium(a.Rotate(a.AXIS_Z, a.Cube((-1.0, -1.0, -1.0), (1.0, 0.5, 1.2), a.Dielectric(1.5)), 25.0), 1.0, a.SolidTexture(0.8, 0.8, 0.8))])
run("list in item", lambda a: [(lambda l: (l.push(a.Sphere((0, 0, 0), 1.0, mats["lamb"](a))), l.push(a.Cube((1, -1, -1), (2, 0, 0), mats["metal"](a))), l)[-1])(a.HittableList())])
run("bvh mixed", lambda a: [a.BVHNode([a.Sphere((0, 0, 0), 1.0, mats["lamb"](a)), a.Cube((1, -1, -1), (2, 0, 0), mats["metal"](a)), a.Rect(a.PLANE_XY, -1, -1, 1, 1, -1.5, mats["lamb"](a)), a.Sphere((-2, 0, 0), 0.7, mats["glass"](a))], 0.0, 1.0)])
run("rot bvh", lambda a: [a.Rotate(a.AXIS_X, a.BVHNode([a.Sphere((0, 0, 0), 1.0, mats["lamb"](a)), a.Cube((1, -1, -1), (2, 0, 0), mats["metal"](a)), a.Sphere((-2, 0, 0), 0.7, mats["glass"](a))], 0.0, 1.0), 30.0)])
