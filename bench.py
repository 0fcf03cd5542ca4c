#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X render path.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (WORLD_SIZE unset): this process starts the N rank processes itself, as fresh
children, BEFORE anything touches the GPU, waits for them and relays rank 0's JSON line.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK/LOCAL_RANK/WORLD_SIZE in
the environment) it is one of the ranks.  One rank per GPU, backend "nccl" (= RCCL over xGMI).

A "step" is one pass of the hot path over one batch: one whole render of the headline workload of
BASELINE.json — final_scene 1920x1080 at 1000 spp per GPU — with the scene already resident in HBM.  With N
GPUs the image is tile-partitioned (tile % N == rank), every rank renders its tiles and ONE gather brings the
framebuffer to rank 0; the per-GPU work is held fixed (spp = 1000 * N), i.e. weak scaling.  `value` is
whole-job Msamples/s = nx*ny*spp*K / max-over-ranks wall time.  At N > 1 the BASELINE config C5 itself
(final_scene 1920x1080x5000spp, strong-scaled: 5000/N spp-equivalent per GPU) is timed as well and reported
next to it as `baseline_config`.

Printed by rank 0 as ONE JSON line, with
  roofline     : what bounds the dominant kernel.  The scene (< 2 MB) is L2 resident, so HBM is not the bound
                 (measured traffic is < 1 % of peak): the kernel is VALU bound.  `frac` = algorithmic flops
                 (operation counts of the CPU oracle x SURVEY §8(d) cost table) / launch time (HIP events on the
                 launch stream) / 157.3 TFLOP/s; beside it the measured SQ counter fractions of the committed
                 rocprofv3 PMC pass (issue_frac, lane_util, wait_frac) and the measured HBM fraction.  The §8(d)
                 algorithmic-bytes figure is kept as a labelled secondary field (it prices L2 hits as HBM fetches
                 and is not a bound).
  cpu_baseline : the uninstrumented f64 CPU oracle (-O3 -march=native, compiled on this host; a port of the
                 reference's single-threaded loop) timed on a bounded sample of the same workload (N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "Msamples/s (W×H×spp/s) + wall-clock to PPM, final_scene 1920×1080×1000spp"
BASELINE_C5 = ("final_scene", 1920, 1080, 5000)  # BASELINE.json configs[4]


# ------------------------------------------------------------------------------------------------
# CPU baseline legs (oracle = test infrastructure; it is only ever the thing compared against)
# ------------------------------------------------------------------------------------------------
def cpu_sample(scene, nx, ny, spp, nrows, native, seed_scene=1):
    """Renders `nrows` evenly spaced rows of the workload at `spp` samples per pixel with the f64 oracle
    (literal restatement, recursive color).  native=True: the uninstrumented -march=native build (timing);
    native=False: the instrumented build (operation counts).  Returns (samples, seconds, counters)."""
    from oracle.oracle import Oracle
    from raytracing_rust_amd import scenes

    orc = Oracle("f64", native=native)
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

    orc = Oracle("f64", native=True)
    cam, world = scenes.build(orc, scene, nx, ny, seed=1)
    _ALLCORE.update(orc=orc, cam=cam, world=world, nx=nx, ny=ny)


def _allcore_worker(job):
    spp, rows = job
    a = _ALLCORE
    for r in rows:
        a["orc"].render(a["cam"], a["world"], a["nx"], a["ny"], spp, seed=42, flags=0, rows=(r, r + 1))
    return len(rows) * a["nx"] * spp


def cpu_sample_allcore(scene, nx, ny, spp, rows_per_worker, nproc):
    """The same oracle on all host cores: `nproc` forked workers (scene built once per worker, not timed), each
    rendering its own rows (the reference itself is single-threaded; this is the generous baseline of
    BASELINE.md §3).  Must run BEFORE the process touches the GPU (fork)."""
    import multiprocessing as mp

    from oracle import oracle as orc_mod

    orc_mod.build_native()  # once, in the parent: the workers inherit the "built" flag
    total_rows = rows_per_worker * nproc
    rows = [int((k + 0.5) * ny / total_rows) for k in range(total_rows)]
    ctx = mp.get_context("fork")
    with ctx.Pool(nproc, initializer=_allcore_init, initargs=(scene, nx, ny)) as pool:
        pool.map(_allcore_worker, [(1, [0])] * nproc)  # every worker has built its scene
        t0 = time.perf_counter()
        n = sum(pool.map(_allcore_worker, [(spp, rows[w::nproc]) for w in range(nproc)], chunksize=1))
        dt = time.perf_counter() - t0
    return n, dt


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as fresh children (no GPU call has happened in this process)
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, timeout_s):
    """Starts the n rank processes (one per GPU) as fresh children, relays rank 0's stdout (the JSON line) and returns
    the exit code.  A rank that dies takes the others down at once (they would otherwise sit in a collective until its
    time-out); only the exact PIDs started here are ever signalled."""
    import tempfile

    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None))
    deadline = time.time() + timeout_s
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes) or time.time() > deadline:
                rc = 124 if time.time() > deadline else 1
                break
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:  # exactly the children started above, by PID
                p.kill()
                p.wait()
    for r, p in enumerate(procs):
        if p.returncode != 0:
            if rc in (0, 1) and p.returncode and p.returncode > 0:
                rc = p.returncode
            elif rc == 0:
                rc = 1
            sys.stderr.write("bench.py: rank %d exited with code %s\n" % (r, p.returncode))
    out0.seek(0)
    sys.stdout.write(out0.read().decode(errors="replace"))
    sys.stdout.flush()
    return rc


def load_profile_summary(path, workload):
    """profiles/pmc_summary.json (tools/pmc_summary.py): per workload the HBM bytes and the SQ counter sums of
    one launch of the dominant kernel, from rocprofv3 --pmc passes committed under profiles/."""
    try:
        d = json.load(open(path))
    except Exception:
        return None
    return d.get("workloads", {}).get(workload)


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
    ap.add_argument("--profile-json", default=os.path.join(ROOT, "profiles", "pmc_summary.json"))
    ap.add_argument("--cpu-allcore-procs", type=int, default=min(16, os.cpu_count() or 1),
                    help="workers of the all-core CPU sample (0 = skip)")
    ap.add_argument("--no-baseline-config", action="store_true",
                    help="N > 1: skip the extra strong-scaled run of BASELINE config C5 (final_scene x5000spp)")
    ap.add_argument("--launch-timeout", type=int, default=3000, help="seconds the self-launcher waits for its ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing in this process has initialised the GPU (torch is not even imported).
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))

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

    from raytracing_rust_amd import Host, abi, dist as rdist, roofline, scenes, write_ppm

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, env_world))
    if args.backend == "nccl" and torch.cuda.device_count() < env_world:
        raise SystemExit("--gpus %d needs %d GPUs, this host shows %d (use --backend gloo to rehearse with ranks sharing "
                         "GPUs)" % (args.gpus, env_world, torch.cuda.device_count()))
    if not torch.cuda.is_available() or abi.load_rtmi().rtmi_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the rtmi render path has no CPU fallback")
    rank, world, local_rank = rdist.init_process_group(args.backend)
    if world > 1:
        world = dist.get_world_size()  # the ranks the backend actually connected
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    nx, ny = args.nx, args.ny
    host = Host()
    t_b = time.perf_counter()
    cam, world_obj = scenes.build(host, args.scene, nx, ny, seed=1)
    scene = host.lower(world_obj)
    t_build = time.perf_counter() - t_b
    t_u = time.perf_counter()
    scene.upload(local_rank)
    t_upload = time.perf_counter() - t_u
    stream = torch.cuda.current_stream(device)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    def timed_run(ns, steps, warmup):
        """`warmup` untimed + `steps` timed whole-image renders at ns samples per pixel (tiles of this rank + the one
        gather), bracketed by barrier + synchronize; returns max-over-ranks seconds and the per-launch kernel times."""
        params = rdist.rank_params(nx, ny, ns, rank, world, seed=42, flags=args.flags, spp_chunks=args.chunks,
                                   shade_threshold=args.shade_threshold, sample_buffer_bytes=args.sample_buffer_mb << 20)
        local = rdist.new_local_framebuffer(params, device)
        scene.prepare(params)  # per-sample radiance buffer (33 GB at the headline size) allocated before any timed step
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

        for _ in range(warmup):
            step()
        scene.check_status()  # the asynchronous warm-up calls report a traversal-pool overflow here
        fence()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            step(evs[k])
        fence()
        elapsed = time.perf_counter() - t0
        el = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        kernel_ms = [a.elapsed_time(b) for a, b in evs]
        return {"elapsed": float(el.item()), "kernel_ms": float(np.mean(kernel_ms)), "render_ms": float(np.mean(render_ms)),
                "params": params, "step": step}

    ns = args.spp * world
    run = timed_run(ns, args.steps, args.warmup)
    elapsed, kernel_ms_avg, render_ms_avg, params, step = (run["elapsed"], run["kernel_ms"], run["render_ms"],
                                                           run["params"], run["step"])

    # ---- wall-clock to PPM: one more pass, now including D2H, un-tiling, P3 text and the file write
    fence()
    t1 = time.perf_counter()
    gathered = step()
    wall_ppm = None
    checks = {}
    if rank == 0:
        g = gathered.cpu().numpy()
        lin, rgb = rdist.untile(params, g)
        write_ppm(args.ppm_out, rgb, 3)  # the P3 text of create_image, streamed (rtmi_write_ppm)
        wall_ppm = time.perf_counter() - t1
        checks = {"ppm_bytes": os.path.getsize(args.ppm_out), "rgb_max": int(rgb.max()), "linear_mean": float(lin.mean())}
    scene.check_status()
    fence()

    # ---- N > 1: BASELINE config C5 by name (strong scaling: the fixed 5000 spp image split over the ranks)
    baseline_cfg = None
    if world > 1 and not args.no_baseline_config and (args.scene, nx, ny) == BASELINE_C5[:3]:
        b = timed_run(BASELINE_C5[3], 2, 1)
        baseline_cfg = {"workload": "%s %dx%dx%dspp" % BASELINE_C5, "n_gpus": world, "scaling": "strong",
                        "per_gpu": "%.0f spp-equivalent" % (BASELINE_C5[3] / world), "steps": 2, "warmup": 1,
                        "value": round(float(nx) * ny * BASELINE_C5[3] * 2 / b["elapsed"] / 1e6, 3), "unit": "Msamples/s",
                        "ms_per_step": round(b["elapsed"] / 2 * 1e3, 3), "render_kernel_ms_avg": round(b["render_ms"], 3)}
        scene.check_status()
        fence()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_samples = float(nx) * ny * ns
    value = total_samples * args.steps / elapsed / 1e6
    workload = "%s %dx%dx%dspp" % (args.scene, nx, ny, ns)
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
            "workload": workload,
            "per_gpu": "%dx%dx%dspp-equivalent (tiles t %% %d == rank)" % (nx, ny, args.spp, world),
            "scene_seed": 1, "render_seed": 42, "max_depth": 50, "t_min": 0.001,
            "parallelism": "tile-interleave x%d + one gather (%s)" % (world, "RCCL" if args.backend == "nccl" else "gloo rehearsal"),
            "flags": args.flags,
        },
        "kernel_ms_avg": round(kernel_ms_avg, 3),
        "render_kernel_ms_avg": round(render_ms_avg, 3),
        "wall_clock_to_ppm_s": None if wall_ppm is None else round(wall_ppm, 4),
        "scene_build_s": round(t_build, 3),
        "scene_upload_s": round(t_upload, 3),
        "checks": checks,
    }
    if baseline_cfg is not None:
        out["baseline_config"] = baseline_cfg

    # ---- roofline + cpu baseline (rank 0).  Operation counts come from a small pass of the instrumented oracle;
    # the TIMED baseline is the uninstrumented -march=native build (N = 1 only).
    _, _, counters = cpu_sample(args.scene, nx, ny, 4, 4, native=False)
    work = roofline.per_sample(counters, ns)
    samples_per_launch = total_samples / world  # this rank's launch
    launch_s = render_ms_avg * 1e-3
    achieved_tflops = work["flops"] * samples_per_launch / launch_s / 1e12
    algo_gbs = work["bytes"] * samples_per_launch / launch_s / 1e9
    kernel = "rtmi_render_coop" if (args.flags & 1) and not (args.flags & 24) else (
        "rtmi_render_async" if args.flags & 16 else "rtmi_render_kernel")
    prof = load_profile_summary(args.profile_json, "%s %dx%dx%dspp" % (args.scene, nx, ny, args.spp)) if world == 1 else None
    rl = {
        "bound": "valu",
        "achieved": round(achieved_tflops, 4),
        "peak": roofline.FP32_VALU_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved_tflops / roofline.FP32_VALU_PEAK_TFLOPS, 6),
        "traffic": None,
        "kernel": kernel,
        "flops_per_sample": round(work["flops"], 2),
        "samples_per_launch": samples_per_launch,
        "launch_ms": round(render_ms_avg, 3),
        "why": "scene < 2 MB is L2 resident; measured HBM traffic is < 1 % of peak: VALU issue under divergence binds",
    }
    if prof:
        rl["traffic"] = prof.get("hbm_bytes_per_launch")
        if rl["traffic"]:
            rl["hbm_frac"] = round(rl["traffic"] / launch_s / 1e9 / roofline.HBM_PEAK_GBS, 6)
        sq = roofline.sq_fractions(prof)
        if sq:
            rl.update(sq)  # issue_frac, lane_util, wait_frac of the profiled launch
        rl["counters_from"] = prof.get("source")
    rl["algorithmic_bytes"] = {
        "bytes_per_sample": round(work["bytes"], 2), "gb_per_s": round(algo_gbs, 3),
        "over_hbm_peak": round(algo_gbs / roofline.HBM_PEAK_GBS, 4),
        "note": "SURVEY §8(d) cost table: prices every node/primitive test as an HBM fetch although the scene is L2 "
                "resident; secondary figure, not a bound (may exceed 1)",
    }
    out["roofline"] = rl
    if world == 1 and not args.no_cpu_baseline:
        n_cpu, dt_cpu, _ = cpu_sample(args.scene, nx, ny, args.cpu_spp, args.cpu_rows, native=True)
        cpu_ms = n_cpu / dt_cpu / 1e6
        out["cpu_baseline"] = {
            "value": round(cpu_ms, 5),
            "unit": "Msamples/s",
            "cores": 1,
            "kind": "port",
            "sample": "%d evenly spaced rows x %d px x %d spp of %s %dx%d (%.1f s; f64 oracle, recursive color, no "
                      "operation counters, gcc -O3 -march=native on this host)"
                      % (args.cpu_rows, nx, args.cpu_spp, args.scene, nx, ny, dt_cpu),
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
