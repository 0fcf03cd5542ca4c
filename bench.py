#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X render path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch: one whole render of the headline
workload of BASELINE.json — final_scene 1920x1080 at 1000 spp per GPU — with the scene
already resident in HBM.  With N GPUs the image is tile-partitioned (tile % N == rank),
every rank renders its tiles and ONE gather (RCCL) brings the framebuffer to rank 0; the
per-GPU work is held fixed (spp = 1000 * N), i.e. weak scaling.  `value` is whole-job
Msamples/s = nx*ny*spp*K / max-over-ranks wall time.

Printed by rank 0 as ONE JSON line, with
  roofline     : algorithmic bytes per launch (oracle operation counts x SURVEY §8(d) cost
                 table) / average launch duration from HIP events on the launch stream,
                 against HBM3E 8 TB/s; the VALU-side fraction is reported next to it
  cpu_baseline : the f64 CPU oracle (a port of the reference's single-threaded loop) timed
                 on this box on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "Msamples/s (W×H×spp/s) + wall-clock to PPM, final_scene 1920×1080×1000spp"


def cpu_sample(scene, nx, ny, spp, nrows, seed_scene=1):
    """Times the f64 oracle (literal restatement, recursive color) on `nrows` evenly spaced
    rows of the workload at `spp` samples per pixel.  Returns (samples, seconds, counters)."""
    from oracle.oracle import Oracle
    from raytracing_rust_amd import scenes

    orc = Oracle("f64")
    cam, world = scenes.build(orc, scene, nx, ny, seed=seed_scene)
    rows = [int((k + 0.5) * ny / nrows) for k in range(nrows)]
    orc.reset_counters()
    t0 = time.perf_counter()
    for r in rows:
        orc.render(cam, world, nx, ny, spp, seed=42, flags=0, rows=(r, r + 1))
    dt = time.perf_counter() - t0
    counters = orc.counters()
    orc.free_all()
    return nrows * nx * spp, dt, counters


_ALLCORE = {}


def _allcore_init(scene, nx, ny):
    from oracle.oracle import Oracle
    from raytracing_rust_amd import scenes

    orc = Oracle("f64")
    cam, world = scenes.build(orc, scene, nx, ny, seed=1)
    _ALLCORE.update(orc=orc, cam=cam, world=world, nx=nx, ny=ny)


def _allcore_worker(job):
    spp, rows = job
    a = _ALLCORE
    for r in rows:
        a["orc"].render(a["cam"], a["world"], a["nx"], a["ny"], spp, seed=42, flags=0, rows=(r, r + 1))
    return len(rows) * a["nx"] * spp


def cpu_sample_allcore(scene, nx, ny, spp, rows_per_worker, nproc):
    """The same oracle on all host cores: `nproc` forked workers (scene built once per worker, not
    timed), each rendering its own rows (the reference itself is single-threaded; this is the generous
    baseline of BASELINE.md §3).  Must run BEFORE the process touches the GPU (fork)."""
    import multiprocessing as mp

    total_rows = rows_per_worker * nproc
    rows = [int((k + 0.5) * ny / total_rows) for k in range(total_rows)]
    ctx = mp.get_context("fork")
    with ctx.Pool(nproc, initializer=_allcore_init, initargs=(scene, nx, ny)) as pool:
        pool.map(_allcore_worker, [(1, [0])] * nproc)  # every worker has built its scene
        t0 = time.perf_counter()
        n = sum(pool.map(_allcore_worker, [(spp, rows[w::nproc]) for w in range(nproc)], chunksize=1))
        dt = time.perf_counter() - t0
    return n, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="final_scene")
    ap.add_argument("--nx", type=int, default=1920)
    ap.add_argument("--ny", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000, help="samples per pixel PER GPU (total spp = spp * gpus)")
    ap.add_argument("--flags", type=int, default=1,
                    help="rtmi flags; default 1 = RTMI_FLAG_FAST_CULL (two-phase kernel with wave-cooperative, "
                         "pruned BVH traversal: verified bit-identical to exact mode and to the fp32 oracle)")
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--shade-threshold", type=int, default=0)
    ap.add_argument("--cpu-rows", type=int, default=16)
    ap.add_argument("--cpu-spp", type=int, default=144)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) for real runs; gloo only to rehearse N > 1 on a box with "
                         "fewer GPUs than ranks (ranks share GPUs, the gather goes through host memory)")
    ap.add_argument("--ppm-out", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "rtmi_bench.ppm"))
    ap.add_argument("--sample-buffer-mb", type=int, default=0, help="per-sample buffer budget (0 = library default)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--cpu-allcore-procs", type=int, default=min(16, os.cpu_count() or 1),
                    help="workers of the all-core CPU sample (0 = skip)")
    args = ap.parse_args()

    allcore = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and args.cpu_allcore_procs > 1:
        # before anything initialises the GPU: the workers are forked
        spp_all, rows_all = max(1, args.cpu_spp // 4), 8  # many thin rows per worker: balanced
        n_all, dt_all = cpu_sample_allcore(args.scene, args.nx, args.ny, spp_all, rows_all, args.cpu_allcore_procs)
        allcore = {"value": round(n_all / dt_all / 1e6, 5), "unit": "Msamples/s", "cores": args.cpu_allcore_procs,
                   "sample": "%d rows x %d px x %d spp in %d forked workers (%.1f s)"
                             % (rows_all * args.cpu_allcore_procs, args.nx, spp_all, args.cpu_allcore_procs, dt_all)}

    import numpy as np
    import torch
    import torch.distributed as dist

    from raytracing_rust_amd import Host, abi, dist as rdist, ppm_p3, roofline, scenes

    if not torch.cuda.is_available() or abi.load_rtmi().rtmi_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the rtmi render path has no CPU fallback")
    rank, world, local_rank = rdist.init_process_group(args.backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    nx, ny = args.nx, args.ny
    ns = args.spp * world
    host = Host()
    t_b = time.perf_counter()
    cam, world_obj = scenes.build(host, args.scene, nx, ny, seed=1)
    scene = host.lower(world_obj)
    t_build = time.perf_counter() - t_b
    t_u = time.perf_counter()
    scene.upload(local_rank)
    t_upload = time.perf_counter() - t_u
    params = rdist.rank_params(nx, ny, ns, rank, world, seed=42, flags=args.flags, spp_chunks=args.chunks,
                               shade_threshold=args.shade_threshold, sample_buffer_bytes=args.sample_buffer_mb << 20)
    local = rdist.new_local_framebuffer(params, device)
    scene.prepare(params)  # buffers (33 GB per-sample radiance buffer at the headline size) allocated before any timed step
    stream = torch.cuda.current_stream(device)

    render_ms = []  # the render kernel alone: HIP events recorded inside librtmi on the launch stream

    def step(events=None):
        if events is not None:
            events[0].record(stream)
        st = scene.render_device(cam, params, local.data_ptr(), stream.cuda_stream, want_stats=events is not None)
        if events is not None:
            events[1].record(stream)
            render_ms.append(st["render_ms"])
        if args.backend == "gloo" and world > 1:
            torch.cuda.synchronize(device)
            return rdist.gather_framebuffer(local.cpu(), rank, world)
        return rdist.gather_framebuffer(local, rank, world)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        gathered = step(evs[k])
    fence()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    kernel_ms_avg = float(np.mean(kernel_ms))      # render + resolve kernels (torch events on the launch stream)
    render_ms_avg = float(np.mean(render_ms))      # the dominant kernel alone

    # ---- wall-clock to PPM: one more pass, now including D2H, un-tiling, P3 text and the file write
    fence()
    t1 = time.perf_counter()
    gathered = step()
    wall_ppm = None
    checks = {}
    if rank == 0:
        g = gathered.cpu().numpy()
        lin, rgb = rdist.untile(params, g)
        txt = ppm_p3(rgb)
        with open(args.ppm_out, "wb") as f:
            f.write(txt)
        wall_ppm = time.perf_counter() - t1
        checks = {"ppm_bytes": len(txt), "rgb_max": int(rgb.max()), "linear_mean": float(lin.mean())}
    fence()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_samples = float(nx) * ny * ns
    value = total_samples * args.steps / elapsed / 1e6
    out = {
        "metric": METRIC,
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%s %dx%dx%dspp" % (args.scene, nx, ny, ns),
            "per_gpu": "%dx%dx%dspp-equivalent (tiles t %% %d == rank)" % (nx, ny, args.spp, world),
            "scene_seed": 1, "render_seed": 42, "max_depth": 50, "t_min": 0.001,
            "parallelism": "tile-interleave x%d + one gather" % world,
            "flags": args.flags,
        },
        "kernel_ms_avg": round(kernel_ms_avg, 3),
        "render_kernel_ms_avg": round(render_ms_avg, 3),
        "wall_clock_to_ppm_s": None if wall_ppm is None else round(wall_ppm, 4),
        "scene_build_s": round(t_build, 3),
        "scene_upload_s": round(t_upload, 3),
        "checks": checks,
    }

    # ---- roofline + cpu baseline (rank 0).  Counts come from the oracle; at N > 1 only a small
    # counting pass runs (no timing claim), at N = 1 the timed bounded sample provides both.
    want_cpu = (world == 1) and not args.no_cpu_baseline
    rows, cspp = (args.cpu_rows, args.cpu_spp) if want_cpu else (4, 4)
    n_cpu, dt_cpu, counters = cpu_sample(args.scene, nx, ny, cspp, rows)
    work = roofline.per_sample(counters, ns)
    samples_per_launch = total_samples / world  # this rank's launch
    achieved_gbs = work["bytes"] * samples_per_launch / (render_ms_avg * 1e-3) / 1e9
    achieved_tflops = work["flops"] * samples_per_launch / (render_ms_avg * 1e-3) / 1e12
    traffic = None
    if os.path.exists(args.traffic_json):
        try:
            tj = json.load(open(args.traffic_json))
            if tj.get("workload") == out["config"]["workload"]:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out["roofline"] = {
        "bound": "hbm",
        "achieved": round(achieved_gbs, 3),
        "peak": roofline.HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved_gbs / roofline.HBM_PEAK_GBS, 6),
        "traffic": traffic,
        "kernel": "rtmi_render_coop" if (args.flags & 1) and not (args.flags & 24) else ("rtmi_render_async" if args.flags & 16 else "rtmi_render_kernel"),
        "bytes_per_sample": round(work["bytes"], 2),
        "flops_per_sample": round(work["flops"], 2),
        "samples_per_launch": samples_per_launch,
        "launch_ms": round(render_ms_avg, 3),
        "valu": {"achieved": round(achieved_tflops, 4), "peak": roofline.FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(achieved_tflops / roofline.FP32_VALU_PEAK_TFLOPS, 6)},
        "note": "working set is L2/Infinity-Cache resident: the binding limits are VALU issue, divergence and latency",
    }
    if want_cpu:
        cpu_ms = n_cpu / dt_cpu / 1e6
        out["cpu_baseline"] = {
            "value": round(cpu_ms, 5),
            "unit": "Msamples/s",
            "cores": 1,
            "kind": "port",
            "sample": "%d evenly spaced rows x %d px x %d spp of %s %dx%d (%.1f s, f64 oracle, recursive color)"
                      % (rows, nx, cspp, args.scene, nx, ny, dt_cpu),
            "gpu_over_cpu": round(value / cpu_ms, 1),
            "all_cores": allcore,
        }
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
