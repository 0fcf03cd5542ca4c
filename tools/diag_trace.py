#!/usr/bin/env python3
"""Recover the hit distance of every bounce of one (pixel, sample) from path signatures, device vs fp32 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.oracle import Oracle, ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import Host, abi
nx, ny, NS = 40, 24, 6
host = Host(); orc = Oracle("f32")
M32 = 0xffffffff
def unmix(h, k):
    x = h & M32
    x ^= x >> 16
    x = (x * pow(0x846CA68B, -1, 1 << 32)) & M32
    x ^= (x >> 15) ^ (x >> 30)
    x = (x * pow(0x7FEB352D, -1, 1 << 32)) & M32
    x ^= x >> 16
    x ^= ((k + 1) * 0x9E3779B9) & M32
    return np.array([x], np.uint32).view(np.float32)[0]
def cam(api):
    return api.Camera((6.0, 3.0, 7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, nx / ny, 0.0, 10.0, 0.0, 1.0)
lamb = lambda a: a.Lambertian(a.SolidTexture(0.6, 0.5, 0.4))
def build(a):
    w = a.HittableList()
    w.push(a.Sphere((0.0, 9.0, 0.0), 3.0, a.DiffuseLight(a.SolidTexture(4.0, 4.0, 4.0))))
    w.push(a.Rotate(getattr(a, sys.argv[1] if len(sys.argv) > 1 else "AXIS_Z"), a.Sphere((0.5, 0.2, -0.3), 1.5, lamb(a)), 33.0))
    return w
host.seed_scene_rng(1); orc.seed_scene_rng(1)
ch, wh = cam(host), build(host); co, wo = cam(orc), build(orc)
sc = host.lower(wh)
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def sig_dev(ns, md): return sc.render(ch, nx, ny, ns, seed=42, flags=FLAGS, sig=True, max_depth=md)["sig"]
def sig_orc(ns, md): return orc.render(co, wo, nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM, max_depth=md)["sig"]
full_d, full_o = sig_dev(NS, 50), sig_orc(NS, 50)
bad = np.argwhere(full_d != full_o)
print("mismatching pixels", len(bad), bad[:5].tolist())
def contrib(fn, r, c, s, k):
    def one(ns, md): return int(fn(ns, md)[r, c]) if ns > 0 and md >= 0 else 0
    hi = one(s + 1, k) - one(s, k)
    lo = (one(s + 1, k - 1) - one(s, k - 1)) if k > 0 else 0
    return (hi - lo) % (1 << 64)
cache = {}
def cached(fn):
    def g(ns, md):
        key = (fn.__name__, ns, md)
        if key not in cache: cache[key] = fn(ns, md)
        return cache[key]
    g.__name__ = fn.__name__
    return g
sd, so = cached(sig_dev), cached(sig_orc)
for (r, c) in bad[:4]:
    for s in range(NS):
        line = []
        for k in range(3):
            cd, co_ = contrib(sd, r, c, s, k), contrib(so, r, c, s, k)
            td = unmix(cd, k) if cd else None; to = unmix(co_, k) if co_ else None
            line.append("k%d dev %s orc %s%s" % (k, None if td is None else "%.6f" % td, None if to is None else "%.6f" % to, "" if cd == co_ else " <<<"))
        print("px", r, c, "s", s, " | ".join(line))
