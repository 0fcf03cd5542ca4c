#!/usr/bin/env python3
"""Per-bounce hit distances of individual camera samples, recovered from path signatures (GPU box).

usage: python tools/diag_trace.py <random-scene seed> [device flags=0]

The path signature of a pixel is the wrapping sum of mix(bits(t), bounce) over every hit of every sample;
mix is invertible, so rendering with ns = s, s+1 and max_depth = k-1, k isolates the hit distance of sample s
at bounce k.  Prints them for the first mismatching pixels, device (with `flags`) next to the fp32 oracle:
shows at which bounce two implementations part and whether by rounding or by a different hit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes_random
from oracle.oracle import ARITH_DEVICE, THROUGHPUT_FORM, Oracle
from raytracing_rust_amd import Host

M32 = 0xFFFFFFFF


def unmix(h, k):
    x = h & M32
    x ^= x >> 16
    x = (x * pow(0x846CA68B, -1, 1 << 32)) & M32
    x ^= (x >> 15) ^ (x >> 30)
    x = (x * pow(0x7FEB352D, -1, 1 << 32)) & M32
    x ^= x >> 16
    x ^= ((k + 1) * 0x9E3779B9) & M32
    return float(np.array([x], np.uint32).view(np.float32)[0])


def main():
    seed = int(sys.argv[1])
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    nx, ny, ns = 40, 24, 6
    host, orc = Host(), Oracle("f32")
    cam, world = scenes_random.build(host, seed, nx, ny)
    camo, worldo = scenes_random.build(orc, seed, nx, ny)
    sc = host.lower(world)
    cache = {}

    def sig(which, n, md):
        if n <= 0 or md < 0:
            return np.zeros((ny, nx), np.uint64)
        key = (which, n, md)
        if key not in cache:
            cache[key] = (sc.render(cam, nx, ny, n, seed=42, flags=flags, sig=True, max_depth=md)["sig"] if which == "dev"
                          else orc.render(camo, worldo, nx, ny, n, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM, max_depth=md)["sig"])
        return cache[key]

    def contrib(which, r, c, s, k):
        one = lambda n, md: int(sig(which, n, md)[r, c])
        return ((one(s + 1, k) - one(s, k)) - (one(s + 1, k - 1) - one(s, k - 1))) % (1 << 64)

    bad = np.argwhere(sig("dev", ns, 50) != sig("orc", ns, 50))
    print("seed", seed, "flags", flags, "mismatching pixels", len(bad), bad[:6].tolist())
    for (r, c) in bad[:3]:
        for s in range(ns):
            parts = []
            for k in range(6):
                a, b = contrib("dev", r, c, s, k), contrib("orc", r, c, s, k)
                parts.append("k%d %s|%s%s" % (k, "-" if not a else "%.6g" % unmix(a, k), "-" if not b else "%.6g" % unmix(b, k), "" if a == b else " <<<"))
            print(" px", r, c, "sample", s, "  ".join(parts))


if __name__ == "__main__":
    main()
