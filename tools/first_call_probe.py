#!/usr/bin/env python3
"""Where does the first render of a handle spend its time?  (r03 verdict: the driver's box showed first_call_s 2.15 s
against a steady 0.83 s; the builder's runs 0.004-0.009 s more.)  Times, in ONE fresh process:
  * raw hipMalloc / hipMemset / hipFree of the per-sample buffer sizes (25, 16, 8, 4 GB), twice each;
  * rtmi_multi_create -> prepare -> render x3 -> destroy at the headline size with the default budget and with
    budgets of 8 and 4 GB, each twice (the second round finds the process's allocator warm);
  * the same after `--burn` steady renders on another handle that is destroyed just before (what bench.py does).
Usage: python tools/first_call_probe.py [--burn N] > log"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--burn", type=int, default=0)
    ap.add_argument("--spp", type=int, default=1000)
    args = ap.parse_args()
    t_start = time.perf_counter()
    from raytracing_rust_amd import Host, abi, scenes

    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]

    def now():
        return time.perf_counter()

    out = {"import_s": round(now() - t_start, 3)}
    t = now()
    assert abi.load_rtmi().rtmi_device_count() >= 1
    hip.hipDeviceSynchronize()
    out["context_s"] = round(now() - t, 3)
    raw = []
    for gb in (25, 16, 8, 4, 25):
        for rep in range(2):
            p = C.c_void_p()
            t0 = now(); rc = hip.hipMalloc(C.byref(p), gb << 30); t1 = now()
            hip.hipMemset(p, 0, gb << 30); hip.hipDeviceSynchronize(); t2 = now()
            hip.hipMemset(p, 0, gb << 30); hip.hipDeviceSynchronize(); t3 = now()
            hip.hipFree(p); t4 = now()
            raw.append({"gb": gb, "rep": rep, "rc": rc, "malloc_ms": round((t1 - t0) * 1e3, 2), "memset1_ms": round((t2 - t1) * 1e3, 2),
                        "memset2_ms": round((t3 - t2) * 1e3, 2), "free_ms": round((t4 - t3) * 1e3, 2)})
    out["raw_hip"] = raw
    print(json.dumps(out), flush=True)

    nx, ny, ns = 1920, 1080, args.spp
    host = Host()
    cam, world = scenes.build(host, "final_scene", nx, ny, seed=1)
    sc = host.lower(world)
    if args.burn:
        sc.upload(0)
        for _ in range(args.burn):
            sc.render(cam, nx, ny, ns, seed=42, flags=1)
        host.free_all()
        cam, world = scenes.build(host, "final_scene", nx, ny, seed=1)
        sc = host.lower(world)
    rows = []
    for budget_gb in (0, 0, 8, 8, 4, 4, 0):
        kw = dict(seed=42, flags=1, sample_buffer_bytes=budget_gb << 30)
        t0 = now(); sc.upload_multi([0]); t1 = now()
        sc.prepare_resident(nx, ny, ns, **kw); t2 = now()
        calls = []
        for _ in range(3):
            t = now(); r = sc.render_resident(cam, nx, ny, ns, **kw); calls.append(round(now() - t, 4))
        t3 = now(); sc.free_multi(); t4 = now()
        rows.append({"budget_gb": budget_gb, "create_s": round(t1 - t0, 4), "prepare_s": round(t2 - t1, 4), "calls_s": calls,
                     "kernel_ms": round(r["stats"]["kernel_ms"], 2), "destroy_s": round(t4 - t3, 4)})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
