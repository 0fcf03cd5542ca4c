"""Device vs fp32 oracle at ~2 x 10^6 camera paths per BASELINE scene (VERDICT r02 "weak" 2: every comparison above
25 k paths used to be device-vs-device).  Default flags (pruned, wave-cooperative traversal on the alternative
trees), radiance, quantised image and per-pixel path signatures compared BIT FOR BIT — this is what checks the code
all kernels share (shade_hit, camera_sample, work_take, the RNG hand-out, the resolve kernel, pass splitting) against
an independent implementation at a size where every unit-queue and pass boundary case occurs.  The oracle renders
bands of rows in spawned worker processes (oracle/parallel.py).
Reference loop: tests/test.rs:62-79; color: src/color.rs:6-23."""
import time

import numpy as np
import pytest

from oracle.oracle import ARITH_DEVICE, THROUGHPUT_FORM
from oracle.parallel import render_parallel
from raytracing_rust_amd import abi

import scenes_extra

pytestmark = pytest.mark.gpu

NX, NY, NS = 256, 144, 56  # 2.06 M paths; 576 tiles x 4 chunks of the unit queue


@pytest.mark.parametrize("name,budget", [
    ("two_spheres", 0),          # C1
    ("random_spheres", 0),       # C2
    ("cornell_box", 8 << 20),    # C3, per-sample buffer budget 8 MiB of the 25 MB needed: passes, sums carried in f64
    ("cornell_smoke", 0),        # C4
    ("final_scene", 12 << 20),   # C5, 3 passes
    ("lit_final_scene", 0),      # C5's object graph with reachable emitters: every material reaches the pixels
])
def test_two_million_paths_match_the_fp32_oracle_bit_for_bit(host, name, budget):
    cam, world = scenes_extra.build(host, name, NX, NY, seed=1)
    sc = host.lower(world).upload(0)
    got = sc.render(cam, NX, NY, NS, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, sig=True, sample_buffer_bytes=budget)
    t0 = time.perf_counter()
    ref = render_parallel("scenes_extra", name, NX, NY, NS, 42, ARITH_DEVICE | THROUGHPUT_FORM)
    dt = time.perf_counter() - t0
    nbits = int((got["linear"] != ref["linear"]).sum())
    nsig = int((got["sig"] != ref["sig"]).sum())
    print("%s: %d paths, oracle %.1f s, device %.1f ms, non-identical channels %d, signature mismatches %d, mean radiance %.5f"
          % (name, NX * NY * NS, dt, got["stats"]["kernel_ms"], nbits, nsig, float(ref["linear"].mean())))
    assert nsig == 0, "%d pixels have a different path signature" % nsig
    assert nbits == 0, "%d radiance channels differ" % nbits
    assert np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"])
    assert int(np.count_nonzero(ref["sig"])) > 0.3 * NX * NY  # the paths are pinned even where the image is black
    if name in ("cornell_box", "lit_final_scene"):
        assert float(ref["linear"].mean()) > 0.01


def test_baseline_c1_two_spheres_at_its_literal_size(host):
    """BASELINE.json configs[0] as written — two_spheres 400x225x100 spp, depth 50 (9.0 M camera paths; the reference's
    own CPU-runnable case, tests/test.rs:165-182 with the camera of :582-593): device == fp32 oracle bit for bit
    (radiance — identically zero, F4/F5 of SURVEY.md —, quantised image, and the path signatures that pin the
    geometry of every bounce)."""
    nx, ny, ns = 400, 225, 100
    cam, world = scenes_extra.build(host, "two_spheres", nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    got = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL, sig=True)
    t0 = time.perf_counter()
    ref = render_parallel("scenes_extra", "two_spheres", nx, ny, ns, 42, ARITH_DEVICE | THROUGHPUT_FORM, band=3)
    print("two_spheres 400x225x100: oracle %.1f s, device %.1f ms" % (time.perf_counter() - t0, got["stats"]["kernel_ms"]))
    assert np.array_equal(got["sig"], ref["sig"]) and int(np.count_nonzero(ref["sig"])) > 0.3 * nx * ny
    assert np.array_equal(got["linear"], ref["linear"]) and np.array_equal(got["rgb8"].astype(np.int32), ref["rgb"])
    assert int(got["rgb8"].max()) == 0  # the reference's image of this scene is black (no emitter, black background)
