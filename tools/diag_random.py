#!/usr/bin/env python3
"""Diagnose a random-scene parity failure: which top-level items are needed to reproduce it (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes_random
from oracle.oracle import Oracle, ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import Host, abi

seed = int(sys.argv[1]); flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nx, ny, ns = 40, 24, 6
host = Host(); orc = Oracle("f32")

def mism(only):
    cam, world = scenes_random.build(host, seed, nx, ny, only)
    sc = host.lower(world)
    camo, worldo = scenes_random.build(orc, seed, nx, ny, only)
    ref = sc.render(cam, nx, ny, ns, seed=42, flags=0, sig=True)  # exact traversal on the device (== oracle)
    got = sc.render(cam, nx, ny, ns, seed=42, flags=flags, sig=True)
    bad = np.argwhere(got["sig"] != ref["sig"])
    a = sc.arrays()
    return bad, a, got, ref

bad, a, got, ref = mism(None)
n = len(a["items"])
print("seed", seed, "items", n, "sig mismatches", len(bad), bad[:6].tolist())
for k, it in enumerate(a["items"]):
    print(" item", k, "kind", it.kind, "first", it.first, "count", it.count, "flags", it.flags, "xforms", it.xform_count,
          [ (a["xforms"][it.xform_first + q].kind) for q in range(it.xform_count)],
          "types", [a["prim_meta"][it.first + q].type for q in range(max(0, min(it.count, 6)))] if it.kind == abi.ITEM_LIST else "bvh")
keep = list(range(n))
for k in range(n):
    trial = [x for x in keep if x != k]
    try:
        b, *_ = mism(trial)
    except Exception as e:
        print(" drop", k, "->", type(e).__name__, e); continue
    print(" drop", k, "-> mismatches", len(b))
    if len(b) > 0:
        keep = trial
print("minimal set", keep)
b, a, got, ref = mism(keep)
print("mismatches", len(b), b[:4].tolist())
for (r, c) in b[:3]:
    print(" px", r, c, "fast", got["linear"][r, c], "exact", ref["linear"][r, c], hex(int(got["sig"][r, c])), hex(int(ref["sig"][r, c])))
it = a["items"]
for k, I in enumerate(it):
    if I.kind == abi.ITEM_BVH:
        nodes = a["nodes"]
        print(" bvh item", k, "root", I.first, "scale", I.scale, "root box", list(I.root_min), list(I.root_max), "nodes", len(nodes), "max depth", a["max_bvh_depth"])
        prim_types = [m.type for m in a["prim_meta"]]
        print("  prim types", prim_types, "prim flags", [m.flags for m in a["prim_meta"]])
