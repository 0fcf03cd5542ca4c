#!/usr/bin/env python3
"""Lane-activity profile of the render kernel (diagnostics build, RTMI_FLAG_PROFILE).
usage: python tools/lane_profile.py [scene] [nx ny spp]  — run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raytracing_rust_amd import Host, abi, dist as rdist, scenes

NAMES = {0: "bounce loop (lanes with work)", 13: "list prim tests", 14: "traversal: at node", 15: "traversal: at leaf",
         16: "shade", 25: "TIME camera", 26: "TIME list items", 27: "TIME BVH item 0", 28: "TIME other BVH items",
         29: "TIME media", 30: "TIME shading", 31: "TIME loop overhead", 20: "async ST_ITEM", 21: "async ST_NODE", 22: "async ST_PRIM", 23: "async ST_SHADE", 24: "async ST_NEW"}


def run(name, nx, ny, ns, flags, threshold=0):
    host = Host()
    cam, world = scenes.build(host, name, nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    dev = torch.device("cuda", 0)
    prof = torch.zeros(64, dtype=torch.int64, device=dev)
    p = rdist.rank_params(nx, ny, ns, 0, 1, seed=42, flags=flags | abi.RTMI_FLAG_PROFILE, shade_threshold=threshold, spp_chunks=1)
    p.prof = prof.data_ptr()
    local = rdist.new_local_framebuffer(p, dev)
    st = sc.render_device(cam, p, local.data_ptr(), torch.cuda.current_stream().cuda_stream, want_stats=True)
    torch.cuda.synchronize()
    c = prof.cpu().numpy()
    print("== %s %dx%dx%d flags=%d  kernel %.1f ms (profiling build)" % (name, nx, ny, ns, flags, st["render_ms"]))
    samples = nx * ny * ns
    tot_wave = 0
    print("  deepest cooperative pool: %d entries (capacity 64 x (max BVH depth + 2) = %d)" % (int(c[2 * 18]), 64 * (sc.arrays()["max_bvh_depth"] + 2)))
    c[2 * 18] = 0
    for s in range(32):
        act, wav = int(c[2 * s]), int(c[2 * s + 1])
        if wav == 0:
            continue
        label = NAMES.get(s, "traversal of item %d" % (s - 1) if 1 <= s <= 12 else "slot %d" % s)
        if label.startswith("TIME"):
            tt = sum(int(c[2 * k]) for k in range(25, 32))
            print("  %-32s cycles %16d  share %5.1f%%  stamps %12d  cycles/stamp %8.1f" % (label, act, 100.0 * act / max(tt, 1), wav, act / max(wav, 1)))
            continue
        print("  %-32s lane-iters %14d  wave-iters*64 %14d  util %5.1f%%  wave-iters/sample-wave %8.2f"
              % (label, act, wav, 100.0 * act / wav, wav / 64.0 / (samples / 64.0)))
    host.free_all()


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "final_scene"
    nx, ny, ns = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (640, 360, 16)
    thr = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    run(name, nx, ny, ns, abi.RTMI_FLAG_FAST_CULL, thr)  # cooperative
