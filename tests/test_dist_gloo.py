"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercise tile ownership, the
padded local framebuffers, the ONE gather and the un-tiling.  The render itself needs a GPU (no CPU
fallback), so every rank fills its local tile-packed framebuffer with a deterministic function of
the pixel it owns; rank 0 must recover the raster image exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _expected(nx, ny):
    rows, cols = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    lin = np.stack([rows * 0.5, cols * 0.25, rows + cols], -1).astype(np.float32)
    rgb = np.stack([rows & 255, cols & 255, (rows * 7 + cols) & 255], -1).astype(np.uint8)
    return lin, rgb


def _worker(rank, world, port, nx, ny, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from raytracing_rust_amd import dist as rdist

    r, w, _ = rdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    p = rdist.rank_params(nx, ny, 4, rank, world, seed=42)
    local = rdist.new_local_framebuffer(p, torch.device("cpu"))
    # fill the tiles this rank owns exactly like the kernel lays them out
    lin, rgb = _expected(nx, ny)
    txn = (nx + 7) // 8
    tiles = txn * ((ny + 7) // 8)
    buf = local.numpy()
    for t in range(rank, tiles, world):
        lt = t // world
        ty, tx = divmod(t, txn)
        for ly in range(8):
            for lx in range(8):
                row, px = ty * 8 + ly, tx * 8 + lx
                if row < ny and px < nx:
                    k = lt * 64 + ly * 8 + lx
                    buf[k, :3] = lin[row, px]
                    packed = int(rgb[row, px, 0]) | (int(rgb[row, px, 1]) << 8) | (int(rgb[row, px, 2]) << 16)
                    buf[k, 3] = np.array([packed], np.uint32).view(np.float32)[0]
    gathered = rdist.gather_framebuffer(local, rank, world)
    if rank == 0:
        got_lin, got_rgb = rdist.untile(p, gathered.numpy())
        np.savez(out_path, lin=got_lin, rgb=got_rgb)
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nx,ny", [(2, 40, 24), (3, 37, 21), (2, 8, 8)])
def test_tile_shard_gather_untile(tmp_path, world, nx, ny):
    out = str(tmp_path / "img.npz")
    mp.spawn(_worker, args=(world, _free_port(), nx, ny, out), nprocs=world, join=True)
    got = np.load(out)
    lin, rgb = _expected(nx, ny)
    assert np.array_equal(got["lin"], lin)
    assert np.array_equal(got["rgb"], rgb)


def test_every_pixel_has_exactly_one_owner():
    sys.path.insert(0, ROOT)
    import ctypes as C

    from raytracing_rust_amd import abi, dist as rdist

    lib = abi.load_rtmi()
    for nx, ny, world in [(1920, 1080, 8), (1200, 800, 4), (400, 225, 2), (7, 3, 8)]:
        tiles = ((nx + 7) // 8) * ((ny + 7) // 8)
        per = [lib.rtmi_local_tiles(C.byref(rdist.rank_params(nx, ny, 1, r, world))) for r in range(world)]
        assert sum(per) == tiles
        assert max(per) - min(per) <= 1  # interleaved: balanced to within one tile


def test_bench_self_launches_its_ranks_without_a_launcher():
    """`python bench.py --gpus N` with WORLD_SIZE unset must itself start N rank processes as fresh
    children (before any GPU call in the parent) instead of dying on the world-size check.  Without a GPU the
    ranks stop at their own "needs a GPU" check — which proves they were started with the rank environment."""
    import subprocess

    if __import__("raytracing_rust_amd").abi.load_rtmi().rtmi_device_count() > 0:
        pytest.skip("CPU-only check (on a GPU box the ranks would render)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline", "--nx", "64", "--ny", "40", "--spp", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE" not in r.stderr, r.stderr  # the old failure: "--gpus 2 but WORLD_SIZE=1"
    # a rank got as far as its own device check (the launcher stops the other one as soon as the first has failed)
    assert 1 <= r.stderr.count("bench.py needs a GPU") <= 2, r.stderr
    assert "rank 0 exited" in r.stderr or "rank 1 exited" in r.stderr


@pytest.mark.gpu
def test_two_ranks_render_and_gather_on_the_gpu(tmp_path):
    """N = 2 for real: two processes (gloo, sharing this box's one GPU) each RENDER their tiles on the device, one
    gather, un-tile on rank 0 — equals the single-process image bit for bit (the CPU test above
    only fills the framebuffers synthetically)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    ppm = str(tmp_path / "n2.ppm")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline", "--no-baseline-config", "--scene", "cornell_box", "--nx", "200",
                        "--ny", "120", "--spp", "8", "--ppm-out", ppm], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["workload"] == "cornell_box 200x120x16spp"
    sys.path.insert(0, ROOT)
    from raytracing_rust_amd import Host, abi, ppm_p3, scenes

    host = Host()
    cam, world = scenes.build(host, "cornell_box", 200, 120, seed=1)
    one = host.lower(world).render(cam, 200, 120, 16, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
    assert open(ppm, "rb").read() == ppm_p3(one["rgb8"])
    assert one["rgb8"].max() > 0
