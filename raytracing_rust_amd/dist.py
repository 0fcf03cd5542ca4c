"""Tile sharding of one image across GPUs: one process per GPU, one gather.

Pixels are independent given the counter RNG keyed by (pixel, sample) and the scene is
small, so every rank holds the whole scene and renders the 8x8 tiles t with
t % world == rank (interleaved: neighbouring tiles have similar cost, so the split is
balanced).  The only exchange is ONE gather of the tile-packed framebuffers to rank 0
(`torch.distributed.gather`; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU) —
no all-reduce, no ring.  The result is identical for any world size.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .host import default_params


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_process_group(backend=None, collective_at_1=False):
    """torch.distributed from the torchrun environment (MASTER_ADDR must be 127.0.0.1 on one node).
    collective_at_1: also a single rank makes its (one-rank) process group, so that gather_framebuffer() really
    crosses the backend — under "nccl" that is RCCL's communicator set-up and one ncclGather-shaped exchange per
    image on the one GPU of a one-GPU box, the same code the N > 1 ranks run.  Costs microseconds per image."""
    import torch
    import torch.distributed as dist

    rank, world, local_rank = env_rank_world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if (world > 1 or collective_at_1) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511" if world > 1 else str(_free_port()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def padded_local_tiles(params):
    """Tiles in rank 0's local buffer = the largest of all ranks (gather needs equal sizes)."""
    p0 = abi.RenderParams.from_buffer_copy(params)
    p0.tile_rank = 0
    return abi.load_rtmi().rtmi_local_tiles(C.byref(p0))


def new_local_framebuffer(params, device):
    """[padded_tiles*64, 4] float32: r, g, b, bits(rgb8) — the rtmi_texel layout."""
    import torch

    return torch.zeros((padded_local_tiles(params) * 64, 4), dtype=torch.float32, device=device)


def gather_framebuffer(local, rank, world, group=None):
    """The single collective of the path.  Returns the [world, padded*64, 4] tensor on rank 0, else None."""
    import torch
    import torch.distributed as dist

    if world == 1 and not dist.is_initialized():
        return local.unsqueeze(0)  # no process group: nothing to cross (init_process_group(collective_at_1=True) makes one)
    if rank == 0:
        out = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=out, dst=0, group=group)
        return torch.stack(out, 0)
    dist.gather(local, gather_list=None, dst=0, group=group)
    return None


def untile(params, gathered_host):
    """gathered_host: numpy float32 [world, padded*64, 4] (rtmi_texel bits) -> (linear f32 [ny,nx,3], rgb8)."""
    lib = abi.load_rtmi()
    g = np.ascontiguousarray(gathered_host, dtype=np.float32)
    nx, ny = params.nx, params.ny
    lin = np.zeros((ny, nx, 3), np.float32)
    rgb = np.zeros((ny, nx, 3), np.uint8)
    rc = lib.rtmi_untile(C.byref(params), g.ctypes.data, lin.ctypes.data, rgb.ctypes.data)
    if rc != 0:
        raise RuntimeError("rtmi_untile: " + (lib.rtmi_last_error() or b"").decode())
    return lin, rgb


def rank_params(nx, ny, ns, rank, world, **kw):
    return default_params(nx, ny, ns, tile_rank=rank, tile_world=world, **kw)
