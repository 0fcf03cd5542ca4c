"""The C-ABI library loads on a CPU-only box and exports every symbol include/rtmi.h declares;
struct layouts match the header; the host-side helpers (untile, P3 writer) work without a GPU;
entry points that need a device fail loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracing_rust_amd import abi, default_params, dist as rdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = abi.load_rtmi()
    names = _declared_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), "librtmi.so does not export " + n
    assert sorted(abi.RTMI_SYMBOLS) == names


def test_struct_sizes_match_header_comments():
    assert C.sizeof(abi.Texture) == 32
    assert C.sizeof(abi.Material) == 16
    assert C.sizeof(abi.PrimMeta) == 16
    assert C.sizeof(abi.BvhNode) == 64
    assert C.sizeof(abi.Xform) == 16
    assert C.sizeof(abi.Item) == 64
    assert C.sizeof(abi.Texel) == 16
    assert C.sizeof(abi.Perlin) == 256 * 4 * 4 + 3 * 256 * 4
    assert C.sizeof(abi.Camera) == 21 * 4


def test_local_tiles_and_untile_roundtrip():
    lib = abi.load_rtmi()
    nx, ny, world = 37, 21, 3  # ragged: 5 x 3 tiles, partial on both edges
    tiles = ((nx + 7) // 8) * ((ny + 7) // 8)
    counts = [lib.rtmi_local_tiles(C.byref(default_params(nx, ny, 1, tile_rank=r, tile_world=world))) for r in range(world)]
    assert sum(counts) == tiles and counts[0] == max(counts)
    p0 = default_params(nx, ny, 1, tile_rank=0, tile_world=world)
    g = np.zeros((world, counts[0] * 64, 4), np.float32)
    txn = (nx + 7) // 8
    for t in range(tiles):
        r, lt = t % world, t // world
        ty, tx = divmod(t, txn)
        for ly in range(8):
            for lx in range(8):
                row, px = ty * 8 + ly, tx * 8 + lx
                g[r, lt * 64 + ly * 8 + lx, :3] = (row, px, t)
                g[r, lt * 64 + ly * 8 + lx, 3] = np.array([(row & 255) | ((px & 255) << 8) | ((t & 255) << 16)], np.uint32).view(np.float32)[0]
    lin, rgb = rdist.untile(p0, g)
    rows, cols = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    assert np.array_equal(lin[..., 0], rows) and np.array_equal(lin[..., 1], cols)
    assert np.array_equal(rgb[..., 0], rows & 255) and np.array_equal(rgb[..., 1], cols & 255)


def test_bad_parameters_are_rejected():
    lib = abi.load_rtmi()
    p = default_params(0, 10, 1)
    assert lib.rtmi_untile(C.byref(p), None, None, None) == 1  # RTMI_ERR_INVALID
    assert b"positive" in lib.rtmi_last_error()
    d = abi.SceneDesc()
    h = C.c_void_p()
    assert lib.rtmi_scene_create(C.byref(d), 0, C.byref(h)) == 1
    assert not h.value


@pytest.mark.skipif(abi.load_rtmi().rtmi_device_count() > 0, reason="CPU-only check")
def test_no_cpu_fallback_for_rendering(host):
    """Without a GPU the render path must fail loudly (no silent CPU route)."""
    from raytracing_rust_amd import HostError, scenes

    cam, world = scenes.build(host, "two_spheres", 16, 16)
    with pytest.raises(HostError) as e:
        host.lower(world).render(cam, 16, 16, 1)
    assert "no HIP device" in str(e.value)
    with pytest.raises(HostError):
        cam.render(world, 16, 16, 1)
    out = np.zeros(4, np.float32)
    assert abi.load_rtmi().rtmi_probe_math(0, out.ctypes.data, None, out.ctypes.data, 4) == 3  # RTMI_ERR_DEVICE
