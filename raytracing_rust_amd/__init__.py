"""raytracing_rust_amd — MI355X-native (gfx950) per-pixel render loop of DrStiev/raytracing_rust.

The package holds only what the hot path needs:
  csrc/   hand-written HIP kernels + the C ABI of include/rtmi.h (librtmi.so)
  host/   C++ mirror of the reference's Hittable/Material/Texture/Camera surface, lowering
  host.py / abi.py   ctypes face of both libraries
  scenes.py          the reference's eight test scenes (tests/test.rs:89-523)
  dist.py            tile sharding across GPUs (one process per GPU) + framebuffer gather

Importing the package does not need a GPU; rendering does, and fails loudly without one.
"""
from . import abi, scenes  # noqa: F401
from .host import Host, HostError, Panic, Unsupported, Scene, default_params, ppm_p3, release_cached, write_ppm  # noqa: F401

__all__ = ["Host", "HostError", "Panic", "Unsupported", "Scene", "default_params", "ppm_p3", "release_cached", "write_ppm", "abi", "scenes"]
