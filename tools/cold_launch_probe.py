#!/usr/bin/env python3
"""Why is the first render of a fresh process slower than the steady state (bench.py cold_start: kernel 1354 ms
against 820 ms)?  One fresh process per mode, kernel_ms from HIP events inside librtmi:
  plain   : full render x3
  tiny    : a 64x64x8 render first, then full x2      (code object loaded, queue and scratch set up, GPU still idle-clocked)
  spin N  : N ms of hipMemset traffic first, then full x2  (clocks up, render kernel never run)
Usage: python tools/cold_launch_probe.py plain|tiny|spin [ms]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    from raytracing_rust_amd import Host, scenes

    host = Host()
    nx, ny, ns = 1920, 1080, 1000
    cam, world = scenes.build(host, "final_scene", nx, ny, seed=1)
    sc = host.lower(world)
    out = {"mode": mode}
    if mode == "tiny":
        cs, ws = scenes.build(host, "final_scene", 64, 64, seed=1)
        s2 = host.lower(ws).upload(0)
        out["tiny_kernel_ms"] = round(s2.render(cs, 64, 64, 8, seed=42, flags=1)["stats"]["kernel_ms"], 3)
    if mode == "spin":
        ms = float(sys.argv[2])
        hip = C.CDLL("libamdhip64.so")
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        hip.hipFree.argtypes = [C.c_void_p]
        p = C.c_void_p()
        hip.hipMalloc(C.byref(p), 8 << 30)
        t0 = time.perf_counter()
        n = 0
        while (time.perf_counter() - t0) * 1e3 < ms:
            hip.hipMemset(p, 0, 8 << 30)
            hip.hipDeviceSynchronize()
            n += 1
        out["spin_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        out["memsets"] = n
        hip.hipFree(p)
    sc.upload(0)
    ks = []
    for _ in range(3 if mode == "plain" else 2):
        t = time.perf_counter()
        r = sc.render(cam, nx, ny, ns, seed=42, flags=1)
        ks.append((round(r["stats"]["kernel_ms"], 1), round((time.perf_counter() - t) * 1e3, 1)))
    out["full_kernel_ms_and_call_ms"] = ks
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
