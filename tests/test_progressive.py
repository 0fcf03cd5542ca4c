"""Opt-in progressive output (RTMI_FLAG_PROGRESSIVE; the reference's stand-in is the sleeping bar of
src/progressbar.rs:6-58): after every pass of the sample range the framebuffer holds the image of the samples so far,
and rtmi_partial_image() fetches it from inside the progress callback of the running rtmi_render call.
The mean over the first k samples in sample order IS the image of a render with ns = k (same streams, same sums,
tests/test.rs:65-71), which is what the test holds a stable snapshot against."""
import ctypes as C

import numpy as np
import pytest

import scenes_extra
from raytracing_rust_amd import abi
from raytracing_rust_amd.host import default_params


def test_partial_image_validates_without_a_device():
    lib = abi.load_rtmi()
    p = default_params(16, 16, 4)
    spp = C.c_uint32(7)
    assert lib.rtmi_partial_image(None, C.byref(p), None, None, C.byref(spp)) == 1
    assert abi.RTMI_FLAG_PROGRESSIVE == 16384


@pytest.mark.gpu
def test_partial_images_are_prefix_renders(host):
    nx, ny, ns = 1280, 720, 384
    cam, world = scenes_extra.build(host, "lit_final_scene", nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    per_sample = ((nx + 7) // 8) * ((ny + 7) // 8) * 64 * abi.RTMI_SAMPLE_SLOT_BYTES
    pass_spp = 32  # 12 passes
    seen = []

    def progress(done, total):
        spp, lin, rgb = sc.partial_image(nx, ny, ns)
        seen.append((spp, lin.copy(), rgb.copy(), done, total))
        return False

    flags = abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_PROGRESSIVE
    final = sc.render(cam, nx, ny, ns, seed=42, flags=flags, sample_buffer_bytes=per_sample * pass_spp, progress=progress)
    plain = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
    assert np.array_equal(final["linear"], plain["linear"]) and np.array_equal(final["rgb8"], plain["rgb8"])
    spps = [s[0] for s in seen]
    print("polls", len(seen), "spp seen", sorted(set(spps)), "kernel ms", final["stats"]["kernel_ms"])
    assert spps == sorted(spps) and all(s % pass_spp == 0 for s in spps) and spps[-1] == ns
    # the last call of the callback (done == total) sees the finished image
    assert seen[-1][3] == seen[-1][4] and np.array_equal(seen[-1][1], final["linear"]) and np.array_equal(seen[-1][2], final["rgb8"])
    assert any(0 < s < ns for s in spps), "no intermediate image was observed (render too short for the 50 ms poll?)"
    # a snapshot is the image of a render with ns = spp: check the stable ones (two consecutive polls at the same spp, i.e.
    # no pass ended during either copy) and at least one
    checked = 0
    for a, b in zip(seen, seen[1:]):
        if 0 < a[0] == b[0] < ns and np.array_equal(a[1], b[1]):
            ref = sc.render(cam, nx, ny, a[0], seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
            assert np.array_equal(a[1], ref["linear"]) and np.array_equal(a[2], ref["rgb8"]), a[0]
            checked += 1
            if checked == 2:
                break
    k = next(s for s in spps if 0 < s < ns)
    snap = next(s for s in seen if s[0] == k)
    ref = sc.render(cam, nx, ny, k, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
    nb = int((snap[1] != ref["linear"]).any(axis=2).sum())
    print("stable snapshots checked:", checked, "; first intermediate spp", k, "pixels differing from the prefix render:", nb)
    assert nb < 0.5 * nx * ny  # (a pass may end during an unstable copy: then part of the pixels belong to the next prefix)
