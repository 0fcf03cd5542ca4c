#!/usr/bin/env python3
"""Diagnose a random-scene parity failure against the fp32 ORACLE: which top-level objects are needed to reproduce it
(GPU box).  usage: python tools/diag_random_oracle.py SEED [instanced 0|1] [flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes_random
from oracle.oracle import Oracle, ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import Host, abi

seed = int(sys.argv[1]); instanced = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nx, ny, ns = 60, 48, 6
host = Host(); orc = Oracle("f32")


def mism(only):
    cam, world = scenes_random.build(host, seed, nx, ny, only, instanced)
    sc = host.lower(world)
    camo, worldo = scenes_random.build(orc, seed, nx, ny, only, instanced)
    ref = orc.render(camo, worldo, nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    got = sc.render(cam, nx, ny, ns, seed=42, flags=flags, sig=True)
    bad = np.argwhere(got["sig"] != ref["sig"])
    orc.free_all()
    return bad, sc.arrays(), got, ref


bad, a, got, ref = mism(None)
print("seed", seed, "instanced", instanced, "lowered items", len(a["items"]), "sig mismatches", len(bad), bad[:6].tolist())
for k, it in enumerate(a["items"]):
    print(" item", k, "kind", it.kind, "first", it.first, "count", it.count, "flags", hex(it.flags), "xforms",
          [(a["xforms"][it.xform_first + q].kind) for q in range(it.xform_count)],
          "types", [a["prim_meta"][it.first + q].type for q in range(max(0, min(it.count, 6)))] if it.kind == abi.ITEM_LIST else "bvh")
n_top = 12
keep = list(range(n_top))
for k in range(2, n_top):
    trial = [x for x in keep if x != k]
    try:
        b, *_ = mism(trial)
    except Exception as e:
        print(" drop", k, "->", type(e).__name__, e); continue
    print(" drop top-level", k, "-> mismatches", len(b))
    if len(b) > 0:
        keep = trial
print("minimal set of top-level objects", keep)
b, a, got, ref = mism(keep)
print("mismatches", len(b), b[:4].tolist())
for k, it in enumerate(a["items"]):
    print(" item", k, "kind", it.kind, "first", it.first, "count", it.count, "flags", hex(it.flags), "xforms",
          [(a["xforms"][it.xform_first + q].kind, round(a["xforms"][it.xform_first + q].x, 3), round(a["xforms"][it.xform_first + q].y, 3), round(a["xforms"][it.xform_first + q].z, 3)) for q in range(it.xform_count)],
          "prims", [(a["prim_meta"][it.first + q].type, hex(a["prim_meta"][it.first + q].flags)) for q in range(max(0, min(it.count, 6)))] if it.kind == abi.ITEM_LIST else ("bvh", it.first))
for (r, c) in b[:3]:
    print(" px", r, c, "device", got["linear"][r, c], "oracle", ref["linear"][r, c], hex(int(got["sig"][r, c])), hex(int(ref["sig"][r, c])))
