"""Opt-in extensions of SURVEY §8(f) n4 beyond the sky (tests/test_sky_extension.py), all OFF by default:
  * RTMI_FLAG_FACE_FORWARD  opaque materials scatter about the normal turned against the ray
                            (the reference never turns it: src/sphere.rs:50, src/rect.rs:58-59)
  * RTMI_FLAG_UV_BOOK       get_sphere_uv with pi/2 instead of FRAC_2_PI (src/sphere.rs:13)
  * "<scene>_corrected"     the three scene slips repaired: final_scene's light rect (tests/test.rs:444-452),
                            cornell_smoke's back wall (:369-377), cornell_box's ceiling (:268-285)
  * progress callback       rtmi_render_params.progress_fn, driven by the device's unit counter
                            (replaces src/progressbar.rs:6-58)
and the error path of the asynchronous entry point (traversal-pool overflow -> poisoned texels + rtmi_scene_status).
Each flag exists in the fp32/f64 oracle and in the C++ mirror too; the device is compared bit-for-bit."""
import ctypes as C

import numpy as np
import pytest

import scenes_extra
from oracle.oracle import ARITH_DEVICE, FACE_FORWARD, SKY, THROUGHPUT_FORM, UV_BOOK
from raytracing_rust_amd import HostError, abi, default_params, scenes


# ---------------------------------------------------------------- CPU: what the flags mean (oracle, mirror)
def test_uv_book_known_answer(orc64):
    """v = (asin(n.y) + pi/2) / pi spans [0, 1] over the sphere; the reference's FRAC_2_PI gives [-0.297, 0.703]."""
    mat = orc64.Lambertian(orc64.SolidTexture(1, 1, 1))
    sph = orc64.Sphere((0, 0, 0), 1.0, mat)
    for y, v_ref, v_book in ((1.0, (np.pi / 2 + 2 / np.pi) / np.pi, 1.0), (-1.0, (-np.pi / 2 + 2 / np.pi) / np.pi, 0.0),
                             (0.0, (2 / np.pi) / np.pi, 0.5)):
        o = (0.0, 5.0 if y > 0 else (-5.0 if y < 0 else 0.0), 0.0) if y else (5.0, 0.0, 0.0)
        d = tuple(-c for c in o)
        assert abs(orc64.hit(sph, o, d)["v"] - v_ref) < 1e-12
        assert abs(orc64.hit(sph, o, d, flags=UV_BOOK)["v"] - v_book) < 1e-12
    orc64.free_all()


def test_face_forward_known_answer(orc64):
    """A Lambertian rect hit from its back side (rect.rs:58-59: normal always +e_k): by default the scattered ray
    leaves about +e_k (through the surface), with the extension about -e_k (back towards the origin side)."""
    mat = orc64.Lambertian(orc64.SolidTexture(1, 1, 1))
    rect = orc64.Rect(orc64.PLANE_XY, -1, -1, 1, 1, 0.0, mat)
    o, d = (0.0, 0.0, -3.0), (0.0, 0.0, 1.0)  # travels along +z: d.n > 0, the back side
    rec = orc64.hit(rect, o, d)
    assert np.array_equal(rec["normal"], [0, 0, 1])
    for seed in range(8):
        plain = orc64.scatter(mat, o, d, 0.0, rec, seed=seed)
        turned = orc64.scatter(mat, o, d, 0.0, rec, flags=FACE_FORWARD, seed=seed)
        rs = plain["d"] - np.array([0, 0, 1.0])  # the same draw in both
        assert np.allclose(turned["d"], np.array([0, 0, -1.0]) + rs)
    # Dielectric is untouched: it resolves the side itself (material.rs:106-114)
    glass = orc64.Dielectric(1.5)
    a = orc64.scatter(glass, o, d, 0.0, rec, seed=3)
    b = orc64.scatter(glass, o, d, 0.0, rec, flags=FACE_FORWARD, seed=3)
    assert np.array_equal(a["d"], b["d"])
    orc64.free_all()


@pytest.mark.parametrize("name", ["cornell_box_corrected", "earth"])
def test_mirror_extensions_equal_f64_oracle(host, orc64, name):
    """C++ mirror (f64, recursive color) with both switches == f64 oracle with both flags, sample by sample."""
    nx, ny = 24, 16
    cam, world = scenes_extra.build(host, name, nx, ny, seed=1)
    camo, worldo = scenes_extra.build(orc64, name, nx, ny, seed=1)
    row = 9
    ref = orc64.render(camo, worldo, nx, ny, 2, seed=42, flags=FACE_FORWARD | UV_BOOK | SKY, rows=(row, row + 1))
    for i in range(0, nx, 5):
        c = sum(host.color_sample(cam, world, nx, ny, i, ny - 1 - row, s, seed=42, sky=True, face_forward=True, uv_book=True)
                for s in range(2))
        assert np.allclose(c / 2.0, ref["mean"][row, i], rtol=1e-12, atol=1e-15)


def test_corrected_scenes_are_lit_and_default_scenes_unchanged(orc64):
    for name in ("final_scene", "cornell_smoke"):
        cam, w = scenes.build(orc64, name, 32, 20, seed=1)
        assert orc64.render(cam, w, 32, 20, 4, seed=42)["rgb"].max() == 0  # the reference's golden property
        cam, w = scenes.build(orc64, name + "_corrected", 32, 20, seed=1)
        assert orc64.render(cam, w, 32, 20, 4, seed=42)["mean"].mean() > 0.05
        orc64.free_all()
    with pytest.raises(KeyError):
        scenes.build(orc64, "two_spheres_corrected", 8, 8)


# ---------------------------------------------------------------- GPU: bit parity with the fp32 oracle
EXT_CASES = [
    ("final_scene_corrected", 48, 32, 8, abi.RTMI_FLAG_FACE_FORWARD | abi.RTMI_FLAG_UV_BOOK, FACE_FORWARD | UV_BOOK),
    ("final_scene_corrected", 48, 32, 8, 0, 0),
    ("cornell_box_corrected", 40, 40, 16, abi.RTMI_FLAG_FACE_FORWARD, FACE_FORWARD),
    ("cornell_smoke_corrected", 40, 40, 8, abi.RTMI_FLAG_FACE_FORWARD, FACE_FORWARD),
    ("cornell_box", 40, 40, 16, abi.RTMI_FLAG_FACE_FORWARD, FACE_FORWARD),
    ("earth", 40, 24, 8, abi.RTMI_FLAG_UV_BOOK | abi.RTMI_FLAG_SKY, UV_BOOK | SKY),
    ("lit_final_scene", 48, 32, 8, abi.RTMI_FLAG_FACE_FORWARD | abi.RTMI_FLAG_UV_BOOK, FACE_FORWARD | UV_BOOK),
    ("lit_random_spheres", 29, 19, 5, abi.RTMI_FLAG_FACE_FORWARD, FACE_FORWARD),  # ragged tiles
]


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, abi.RTMI_FLAG_FAST_CULL, abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL,
                                    abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL],
                         ids=["exact", "coop-fast", "perlane-fast", "async-fast"])
@pytest.mark.parametrize("name,nx,ny,ns,dflags,oflags", EXT_CASES)
def test_device_extensions_match_fp32_oracle(host, orc32, name, nx, ny, ns, dflags, oflags, kernel):
    cam, world = scenes_extra.build(host, name, nx, ny, seed=1)
    got = host.lower(world).render(cam, nx, ny, ns, seed=42, flags=kernel | dflags, sig=True)
    camo, worldo = scenes_extra.build(orc32, name, nx, ny, seed=1)
    ref = orc32.render(camo, worldo, nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM | oflags)
    diff = np.abs(got["linear"].astype(np.float64) - ref["linear"].astype(np.float64))
    print(name, "max abs diff", diff.max(), "mean radiance", float(ref["linear"].mean()))
    assert int((diff > 1e-4).sum()) == 0  # the tolerance asked for
    assert np.array_equal(got["linear"], ref["linear"])  # the level actually held
    assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"])
    assert np.array_equal(got["sig"], ref["sig"])
    assert ref["linear"].mean() > 0.01


@pytest.mark.gpu
def test_extensions_are_off_by_default_and_change_the_image(host):
    nx, ny, ns = 40, 40, 16
    cam, world = scenes.build(host, "cornell_box", nx, ny, seed=1)
    sc = host.lower(world)
    a = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
    b = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_FACE_FORWARD)
    assert not np.array_equal(a["linear"], b["linear"])  # the cubes' min-side faces stop leaking
    cam, world = scenes.build(host, "earth", nx, ny, seed=1)
    sc = host.lower(world)
    a = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_SKY)
    b = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_SKY | abi.RTMI_FLAG_UV_BOOK)
    assert not np.array_equal(a["linear"], b["linear"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,nx,ny,ns", [("hollow_glass", 48, 32, 16)])
def test_negative_radius_leaves_stay_reachable(host, orc32, name, nx, ny, ns):
    """the pruned kernels' padded leaf boxes come from the primitive's true extent (|r|), so a concentric
    Sphere(-r) inside a BVH is tested by every kernel variant exactly as the reference does."""
    cam, world = scenes_extra.build(host, name, nx, ny, seed=1)
    sc = host.lower(world)
    camo, worldo = scenes_extra.build(orc32, name, nx, ny, seed=1)
    ref = orc32.render(camo, worldo, nx, ny, ns, seed=42, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    assert ref["linear"].mean() > 0.01
    for label, flags in (("exact", 0), ("coop-fast", abi.RTMI_FLAG_FAST_CULL),
                         ("coop-fast-reftree", abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_REF_TREE),
                         ("perlane-fast", abi.RTMI_FLAG_SYNC | abi.RTMI_FLAG_FAST_CULL),
                         ("async-fast", abi.RTMI_FLAG_ASYNC | abi.RTMI_FLAG_FAST_CULL)):
        got = sc.render(cam, nx, ny, ns, seed=42, flags=flags, sig=True)
        assert np.array_equal(got["sig"], ref["sig"]), label
        assert np.array_equal(got["linear"], ref["linear"]), label


# ---------------------------------------------------------------- GPU: progress callback, overflow reporting
@pytest.mark.gpu
def test_progress_callback_counts_units_and_can_cancel(host):
    nx, ny, ns = 640, 360, 256
    cam, world = scenes.build(host, "final_scene", nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    seen = []
    img = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, progress=lambda d, t: seen.append((d, t)) and False)
    assert len(seen) >= 2  # ~140 ms of kernel: at least one poll and the final report
    total = seen[-1][1]
    assert seen[-1] == (total, total) and total == 80 * 45 * 16  # tiles x sample chunks of 16
    assert all(a[0] <= b[0] for a, b in zip(seen, seen[1:])) and all(t == total for _, t in seen)
    assert 0 < seen[0][0] <= total
    plain = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
    assert np.array_equal(img["linear"], plain["linear"])  # the callback changes nothing
    # several passes (small sample buffer): the count still runs to the same total
    seen.clear()
    sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, sample_buffer_bytes=80 * 45 * 64 * abi.RTMI_SAMPLE_SLOT_BYTES * 48,
              progress=lambda d, t: seen.append((d, t)) and False)
    assert seen[-1][0] == seen[-1][1] and all(a[0] <= b[0] for a, b in zip(seen, seen[1:]))
    with pytest.raises(HostError) as e:
        sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, progress=lambda d, t: True)
    assert "cancelled" in str(e.value)


@pytest.mark.gpu
def test_traversal_pool_overflow_is_reported_on_every_path(host):
    """ADVICE r1 / VERDICT weak 3: the overflow word was only looked at by calls that asked for stats.  The test knob
    RTMI_FLAG_TEST_OVERFLOW makes the cooperative kernel report one."""
    import torch

    from raytracing_rust_amd import dist as rdist

    nx, ny, ns = 64, 40, 4
    cam, world = scenes.build(host, "final_scene", nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    bad = abi.RTMI_FLAG_FAST_CULL | abi.RTMI_FLAG_TEST_OVERFLOW
    with pytest.raises(HostError) as e:  # blocking call: error code
        sc.render(cam, nx, ny, ns, seed=42, flags=bad)
    assert "overflow" in str(e.value)
    sc.check_status()  # ... reported once, not again
    good = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)  # and the next call is clean
    # asynchronous call without stats: texels poisoned, rtmi_untile refuses them, rtmi_scene_status reports and clears
    p = rdist.rank_params(nx, ny, ns, 0, 1, seed=42, flags=bad)
    local = rdist.new_local_framebuffer(p, torch.device("cuda", 0))
    sc.render_device(cam, p, local.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    tex = local.cpu().numpy()
    assert np.isnan(tex[:, :3]).all() and (tex.view(np.uint32)[:, 3] == abi.RTMI_TEXEL_POISON).all()
    with pytest.raises(RuntimeError):
        rdist.untile(p, tex[None])
    with pytest.raises(HostError) as e:
        sc.check_status()
    assert "overflow" in str(e.value)
    sc.check_status()
    p = rdist.rank_params(nx, ny, ns, 0, 1, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
    sc.render_device(cam, p, local.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lin, _ = rdist.untile(p, local.cpu().numpy()[None])
    assert np.array_equal(lin, good["linear"])


# ---------------------------------------------------------------- GPU: several devices inside the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0, 0], [0, 0]])
def test_render_multi_equals_single_device(host, devices):
    """rtmi_render_multi (multi-GPU inside the C ABI): scene on every listed device, tiles t % n, one
    gather on devices[0].  On one GPU: [0] takes the RCCL-free single path, [0,0,0] rehearses three ranks with
    device-to-device copies (RCCL cannot put two ranks on one device).  Image == rtmi_render bit-for-bit."""
    nx, ny, ns = 100, 60, 12  # ragged tiles, tile count not divisible by 3
    cam, world = scenes_extra.build(host, "lit_final_scene", nx, ny, seed=1)
    sc = host.lower(world)
    one = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL)
    seen = []
    multi = sc.render_multi(cam, nx, ny, ns, devices, seed=42, flags=abi.RTMI_FLAG_FAST_CULL,
                            progress=lambda d, t: seen.append((d, t)) and False)
    assert np.array_equal(multi["linear"], one["linear"]) and np.array_equal(multi["rgb8"], one["rgb8"])
    assert multi["stats"]["samples"] == nx * ny * ns
    assert seen and seen[-1][0] == seen[-1][1]
    assert one["linear"].mean() > 0.01
    with pytest.raises(HostError):
        sc.render_multi(cam, nx, ny, ns, [0, 99], seed=42)
