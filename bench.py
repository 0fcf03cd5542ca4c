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


def host_core_counts():
    """(cores the host reports, cores this process may actually use): os.cpu_count() next to the scheduler affinity
    and the cgroup CPU quota (a container on a big host sees all its cores but is throttled to its share)."""
    host = os.cpu_count() or 1
    usable = host
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    usable = min(usable, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    usable = min(usable, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return host, usable


def cpu_sample_allcore(scene, nx, ny, rows_per_worker, nproc, target_s):
    """The same oracle on all usable host cores: `nproc` forked workers (scene built once per worker, not timed), each
    rendering its own rows (the reference itself is single-threaded; this is the generous baseline of
    BASELINE.md §3).  A short calibration batch sizes the timed batch for about `target_s` seconds of wall clock.
    Must run BEFORE the process touches the GPU (fork).  Returns (samples, seconds, spp of the timed batch)."""
    import multiprocessing as mp

    from oracle import oracle as orc_mod

    orc_mod.build_native()  # once, in the parent: the workers inherit the "built" flag
    total_rows = rows_per_worker * nproc
    rows = [int((k + 0.5) * ny / total_rows) for k in range(total_rows)]
    ctx = mp.get_context("fork")
    with ctx.Pool(nproc, initializer=_allcore_init, initargs=(scene, nx, ny)) as pool:
        pool.map(_allcore_worker, [(1, [0])] * nproc)  # every worker has built its scene
        t0 = time.perf_counter()
        n_cal = sum(pool.map(_allcore_worker, [(4, rows[w::nproc]) for w in range(nproc)], chunksize=1))
        rate = n_cal / (time.perf_counter() - t0)
        spp = max(8, min(4096, int(target_s * rate / (total_rows * nx) + 0.5)))
        t0 = time.perf_counter()
        n = sum(pool.map(_allcore_worker, [(spp, rows[w::nproc]) for w in range(nproc)], chunksize=1))
        dt = time.perf_counter() - t0
    return n, dt, spp


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


def abi_multi_leg(devices, scene_name, nx, ny, ns, flags, steps, ppm_path, compare_single=False, force_rccl=True):
    """The path a host's Camera::render binds for create_image (tests/test.rs:55-85) with a device list: ONE persistent
    handle (rtmi_multi_create), then whole-image renders from it.  Reports, for the same handle, what creating it
    costs, the first call (which allocates the per-sample buffers) and the steady state — each as wall clock from the
    call to the finished PPM file (render + the one gather + D2H + un-tiling + P3 text + write) — and destroy.
    compare_single: also times rtmi_render (the single-device blocking call) on its own handle the same way."""
    import numpy as np

    from raytracing_rust_amd import Host, scenes, write_ppm

    host = Host()
    t = time.perf_counter()
    cam, world = scenes.build(host, scene_name, nx, ny, seed=1)
    sc = host.lower(world)
    t_lower = time.perf_counter() - t
    out = (np.zeros((ny, nx, 3), np.float32), np.zeros((ny, nx, 3), np.uint8))
    kw = dict(seed=42, flags=flags)

    def to_ppm(fn):
        t0 = time.perf_counter()
        r = fn()
        t1 = time.perf_counter()
        write_ppm(ppm_path, r["rgb8"], 3)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t0, r

    # cold: the first create of this process also pays the HIP context of every device and, for distinct devices, the
    # RCCL communicator set (dlopen + ncclCommInitAll; the set goes back to a per-device-list pool at destroy and the
    # next handle takes it over); warm: what a host pays per handle.  A one-entry list gets a one-rank communicator
    # too here (RTMI_FORCE_RCCL), so that the leg runs the same grouped ncclGather at N = 1 as at N = 8.
    if force_rccl:
        os.environ["RTMI_FORCE_RCCL"] = "1"
    t = time.perf_counter()
    sc.upload_multi(devices)
    t_create_cold = time.perf_counter() - t
    sc.free_multi()
    t = time.perf_counter()
    sc.upload_multi(devices)
    t_create = time.perf_counter() - t
    collective = sc.multi_collective()
    t = time.perf_counter()
    sc.prepare_resident(nx, ny, ns, **kw)  # the per-sample radiance buffers (12 B x pixels x spp per device) alone
    t_prepare = time.perf_counter() - t
    call1, first, r = to_ppm(lambda: sc.render_resident(cam, nx, ny, ns, out=out, **kw))
    calls, walls, kern = [], [], []
    for _ in range(steps):
        c, w, r = to_ppm(lambda: sc.render_resident(cam, nx, ny, ns, out=out, **kw))
        calls.append(c); walls.append(w); kern.append(r["stats"]["kernel_ms"])
    checks = {"rgb_max": int(r["rgb8"].max()), "linear_mean": float(r["linear"].mean()), "samples": r["stats"]["samples"]}
    t = time.perf_counter()
    sc.free_multi()
    t_destroy = time.perf_counter() - t
    res = {
        "entry_points": "rtmi_multi_create / rtmi_multi_render / rtmi_multi_destroy", "devices": list(devices),
        "workload": "%s %dx%dx%dspp" % (scene_name, nx, ny, ns), "steps": steps, "collective": collective,
        "prepare_s": round(t_prepare, 4),
        "lower_s": round(t_lower, 4), "create_first_in_process_s": round(t_create_cold, 4), "create_s": round(t_create, 4),
        "destroy_s": round(t_destroy, 4),
        "first_call_s": round(call1, 4), "first_call_to_ppm_s": round(first, 4),
        "steady_call_s": round(float(np.mean(calls)), 4), "steady_to_ppm_s": round(float(np.mean(walls)), 4),
        "steady_to_ppm_min_s": round(float(np.min(walls)), 4),
        "steady_kernel_ms_slowest_device": round(float(np.mean(kern)), 3),
        "steady_msamples_per_s": round(float(nx) * ny * ns / float(np.mean(calls)) / 1e6, 2),
        "one_shot_fixed_cost_s": round(t_create + t_prepare + (call1 - float(np.mean(calls))) + t_destroy, 4),
        "checks": checks,
    }
    if compare_single:
        t = time.perf_counter()
        sc.upload(devices[0])
        t_up = time.perf_counter() - t
        to_ppm(lambda: sc.render(cam, nx, ny, ns, out=out, **kw))
        c1, w1 = [], []
        for _ in range(steps):
            c, w, r1 = to_ppm(lambda: sc.render(cam, nx, ny, ns, out=out, **kw))
            c1.append(c); w1.append(w)
        res["rtmi_render_same_device"] = {"upload_s": round(t_up, 4), "steady_call_s": round(float(np.mean(c1)), 4),
                                          "steady_to_ppm_s": round(float(np.mean(w1)), 4),
                                          "multi_over_single": round(float(np.mean(calls)) / float(np.mean(c1)), 4),
                                          "image_equal": bool(np.array_equal(r1["rgb8"], r["rgb8"]))}
    host.free_all()
    return res


def cold_one_shot(scene_name, nx, ny, ns, flags, ppm_path, spawned_at):
    """The reference's usage model — one render per process (tests/test.rs:802-838: set_camera, final_scene(),
    create_image, write) — through the one-shot entry point: a FRESH process builds the scene, lowers it, calls
    rtmi_render_multi on [0] once (upload + the 25 GB per-sample buffer + kernels + gather + D2H + un-tiling inside) and
    writes the P3 file.  `spawned_at` = the parent's time.time() just before it started this process, so that
    cold_wall_clock_to_ppm_s includes the interpreter start and the imports (numpy, the two libraries; no torch)."""
    t_main = time.time()
    import numpy as np  # noqa: F401

    from raytracing_rust_amd import Host, scenes, write_ppm

    t_imp = time.time()
    host = Host()
    cam, world = scenes.build(host, scene_name, nx, ny, seed=1)
    sc = host.lower(world)
    t_scene = time.time()
    # rtmi_render_multi = create + (reserve +) render + destroy; called piecewise here so that the line can say what the
    # upload, the allocation of the per-sample buffer (a hipMalloc of tens of GB: usually 0.3 ms, now and then much more,
    # DESIGN.md §7) and the render itself took
    sc.upload_multi([0])
    t_create = time.time()
    sc.prepare_resident(nx, ny, ns, seed=42, flags=flags)
    t_alloc = time.time()
    r = sc.render_resident(cam, nx, ny, ns, seed=42, flags=flags)
    t_render = time.time()
    sc.free_multi()
    t_free = time.time()
    write_ppm(ppm_path, r["rgb8"], 3)
    t_ppm = time.time()
    return {
        "entry_points": "rtmi_multi_create + _prepare + _render + _destroy on [0] (= the one-shot rtmi_render_multi, piecewise), "
                        "fresh process, its first GPU calls",
        "workload": "%s %dx%dx%dspp" % (scene_name, nx, ny, ns),
        "cold_wall_clock_to_ppm_s": round(t_ppm - spawned_at, 4),
        "process_start_and_imports_s": round(t_imp - spawned_at, 4), "of_which_interpreter_start_s": round(t_main - spawned_at, 4),
        "scene_build_and_lower_s": round(t_scene - t_imp, 4),
        "create_s": round(t_create - t_scene, 4), "create_note": "HIP context + code object + scene upload",
        "sample_buffer_alloc_s": round(t_alloc - t_create, 4),
        "render_call_s": round(t_render - t_alloc, 4), "kernel_ms": round(r["stats"]["kernel_ms"], 3),
        "destroy_s": round(t_free - t_render, 4), "ppm_write_s": round(t_ppm - t_free, 4),
        "checks": {"ppm_bytes": os.path.getsize(ppm_path), "samples": r["stats"]["samples"]},
    }


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
    ap.add_argument("--cpu-spp", type=int, default=352)  # x 16 rows x 1920 px = 10.8 M samples: a little over 30 s of one core
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) for real runs; gloo only to rehearse N > 1 on a box with "
                         "fewer GPUs than ranks (ranks share GPUs, the gather goes through host memory)")
    ap.add_argument("--no-collective-at-1", action="store_true",
                    help="N = 1: do not make a one-rank process group (by default the single rank initialises the backend and "
                         "every step's gather is a real torch.distributed.gather on it, so the N = 1 line has crossed RCCL)")
    ap.add_argument("--ppm-out", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "rtmi_bench.ppm"))
    ap.add_argument("--sample-buffer-mb", type=int, default=0, help="per-sample buffer budget (0 = library default)")
    ap.add_argument("--profile-json", default=os.path.join(ROOT, "profiles", "pmc_summary.json"))
    ap.add_argument("--cpu-allcore-procs", type=int, default=-1,
                    help="workers of the all-core CPU sample (-1 = every core this process may use: os.cpu_count() "
                         "unless the scheduler affinity or the cgroup quota is smaller; 0 = skip)")
    ap.add_argument("--cpu-allcore-seconds", type=float, default=20.0, help="wall-clock target of the all-core sample")
    ap.add_argument("--abi-multi", default="auto",
                    help="devices of the persistent C-ABI handle leg (rtmi_multi_*): 'auto' = [0] at N = 1 and all N "
                         "devices (in a child process of rank 0, after the ranks have finished) at N > 1; 'off'; or a "
                         "comma-separated device list")
    ap.add_argument("--abi-multi-child", default="", help=argparse.SUPPRESS)  # internal: run only the handle leg
    ap.add_argument("--cold-child", type=float, default=0.0, help=argparse.SUPPRESS)  # internal: the cold one-shot leg
    ap.add_argument("--no-cold-start", action="store_true",
                    help="N = 1: skip the cold one-shot leg (a fresh child process: start -> scene -> rtmi_render_multi -> PPM)")
    ap.add_argument("--abi-multi-steps", type=int, default=3)
    ap.add_argument("--abi-multi-timeout", type=int, default=240, help="seconds the N > 1 child may take")
    ap.add_argument("--no-baseline-config", action="store_true",
                    help="N > 1: skip the extra strong-scaled run of BASELINE config C5 (final_scene x5000spp)")
    ap.add_argument("--launch-timeout", type=int, default=3000, help="seconds the self-launcher waits for its ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing in this process has initialised the GPU (torch is not even imported).
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))

    if args.abi_multi_child:
        # internal: the persistent-handle leg alone, in a fresh process (N > 1: rank 0 starts it once the ranks are done)
        devs = [int(x) for x in args.abi_multi_child.split(",")]
        print(json.dumps(abi_multi_leg(devs, args.scene, args.nx, args.ny, args.spp, args.flags, args.abi_multi_steps,
                                       args.ppm_out + ".multi")), flush=True)
        return

    if args.cold_child:
        print(json.dumps(cold_one_shot(args.scene, args.nx, args.ny, args.spp, args.flags, args.ppm_out + ".cold", args.cold_child)),
              flush=True)
        return

    # ---- N = 1: the cold one-shot, in a fresh child BEFORE this process touches the GPU (and before the CPU legs load the host)
    cold = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_cold_start:
        cmd = [sys.executable, os.path.abspath(__file__), "--scene", args.scene, "--nx", str(args.nx), "--ny", str(args.ny),
               "--spp", str(args.spp), "--flags", str(args.flags), "--ppm-out", args.ppm_out]
        try:
            cp = subprocess.run(cmd + ["--cold-child", repr(time.time())], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            cold = (json.loads(cp.stdout.decode().strip().splitlines()[-1]) if cp.returncode == 0 else
                    {"error": "child exited with code %d" % cp.returncode, "stderr_tail": cp.stderr.decode(errors="replace")[-600:]})
        except subprocess.TimeoutExpired:
            cold = {"error": "child exceeded 240 s and was killed"}
        except Exception as e:
            cold = {"error": repr(e)}

    allcore = None
    host_cores, usable_cores = host_core_counts()
    nproc_all = args.cpu_allcore_procs if args.cpu_allcore_procs >= 0 else min(usable_cores, 256)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and nproc_all > 1:
        # before anything initialises the GPU: the workers are forked
        rows_all = 8  # many thin rows per worker: balanced
        n_all, dt_all, spp_all = cpu_sample_allcore(args.scene, args.nx, args.ny, rows_all, nproc_all, args.cpu_allcore_seconds)
        allcore = {"value": round(n_all / dt_all / 1e6, 5), "unit": "Msamples/s", "cores": nproc_all,
                   "host_cores": host_cores, "usable_cores": usable_cores, "workers": nproc_all,
                   "sample": "%d rows x %d px x %d spp in %d forked workers (%.1f s); the host reports %d cores, this "
                             "process may use %d"
                             % (rows_all * nproc_all, args.nx, spp_all, nproc_all, dt_all, host_cores, usable_cores)}

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
    collective_note = None
    try:
        rank, world, local_rank = rdist.init_process_group(args.backend, collective_at_1=not args.no_collective_at_1)
    except Exception as e:  # a one-rank group that cannot be made must not cost the headline line
        if env_world > 1:
            raise
        collective_note = "one-rank process group failed (%r): no collective ran" % (e,)
        rank, world, local_rank = rdist.env_rank_world()
    if world > 1:
        world = dist.get_world_size()  # the ranks the backend actually connected
    # what the gather of a step really is in this process
    if dist.is_initialized():
        collective = "%s: torch.distributed.gather on a %d-rank process group" % (
            "RCCL" if args.backend == "nccl" else "gloo rehearsal", dist.get_world_size())
    else:
        collective = "none (single rank, no process group%s)" % ("" if collective_note is None else "; " + collective_note)
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
        if dist.is_initialized():
            dist.barrier()
            torch.cuda.synchronize(device)

    def timed_run(ns, steps, warmup):
        """`warmup` untimed + `steps` timed whole-image renders at ns samples per pixel (tiles of this rank + the one
        gather), bracketed by barrier + synchronize; returns max-over-ranks seconds and the per-launch kernel times."""
        params = rdist.rank_params(nx, ny, ns, rank, world, seed=42, flags=args.flags, spp_chunks=args.chunks,
                                   shade_threshold=args.shade_threshold, sample_buffer_bytes=args.sample_buffer_mb << 20)
        local = rdist.new_local_framebuffer(params, device)
        scene.prepare(params)  # per-sample radiance buffer (25 GB at the headline size) allocated before any timed step
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

    if dist.is_initialized():
        dist.destroy_process_group()  # the ranks are done; rank 0 goes on alone
    if rank != 0:
        return

    # ---- the persistent C-ABI handle (rtmi_multi_*): N = 1 in this process on [0]; N > 1 in a fresh child of rank 0
    # over all N devices, after the other ranks have finished (their processes exit; nothing else runs on the GPUs)
    abi_multi = None
    if args.abi_multi != "off":
        devs = list(range(world)) if args.abi_multi == "auto" else [int(x) for x in args.abi_multi.split(",")]
        try:
            host.free_all()  # release this rank's scene and its per-sample buffer first
            torch.cuda.empty_cache()
            if world == 1:
                abi_multi = abi_multi_leg(devs, args.scene, nx, ny, ns, args.flags, args.abi_multi_steps,
                                          args.ppm_out + ".multi", compare_single=True)
            else:
                cmd = [sys.executable, os.path.abspath(__file__), "--abi-multi-child", ",".join(map(str, devs)),
                       "--scene", args.scene, "--nx", str(nx), "--ny", str(ny), "--spp", str(ns), "--flags", str(args.flags),
                       "--abi-multi-steps", str(args.abi_multi_steps), "--ppm-out", args.ppm_out]
                env = {k: v for k, v in os.environ.items()
                       if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.abi_multi_timeout)
                if cp.returncode == 0:
                    abi_multi = json.loads(cp.stdout.decode().strip().splitlines()[-1])
                else:
                    abi_multi = {"error": "child exited with code %d" % cp.returncode, "stderr_tail": cp.stderr.decode(errors="replace")[-600:]}
        except subprocess.TimeoutExpired:
            abi_multi = {"error": "child exceeded %d s and was killed" % args.abi_multi_timeout}
        except Exception as e:  # the headline line must still be printed
            abi_multi = {"error": repr(e)}

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
            "parallelism": "tile-interleave x%d + one gather (%s)" % (world, collective),
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
    if abi_multi is not None:
        out["abi_multi"] = abi_multi
    if cold is not None:
        out["cold_start"] = cold
        out["cold_wall_clock_to_ppm_s"] = cold.get("cold_wall_clock_to_ppm_s")

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
        # Everything below comes from the rocprofv3 PMC passes COMMITTED under profiles/ (one launch of the same kernel
        # and workload on an MI355X), not from this run: kept apart under `profiled`; only `traffic` (the contract's
        # field) is repeated at the top level.
        lib_hash = abi.load_rtmi().rtmi_build_hash().decode()
        pf = {"source": prof.get("source"), "hbm_bytes_per_launch": prof.get("hbm_bytes_per_launch"),
              # the committed counters belong to the build whose hash they carry; another library -> stale
              "profiled_build": prof.get("build_hash"), "running_build": lib_hash,
              "stale": prof.get("build_hash") != lib_hash}
        rl["traffic"] = prof.get("hbm_bytes_per_launch")
        sq = roofline.sq_fractions(prof)
        if sq:
            pf.update(sq)  # issue_frac, lane_util, wait_frac, profiled_launch_ms
            if pf.get("profiled_launch_ms") and pf.get("hbm_bytes_per_launch"):
                pf["hbm_frac"] = round(pf["hbm_bytes_per_launch"] / (pf["profiled_launch_ms"] * 1e-3) / 1e9 / roofline.HBM_PEAK_GBS, 6)
            if "issue_frac" in pf and "lane_util" in pf:
                # the EXECUTED-work figure: share of the chip's VALU lane-cycles that did something
                pf["lane_cycle_frac"] = round(pf["issue_frac"] * pf["lane_util"], 4)
        rl["profiled"] = pf
        rl["frac_note"] = ("frac prices the REFERENCE algorithm's operation count (both children, unshrunk interval) at "
                           "this launch time; the executed share of VALU lane-cycles is profiled.lane_cycle_frac")
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
            "host_cores": host_cores,
            "usable_cores": usable_cores,
            "all_cores": allcore,
        }
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
