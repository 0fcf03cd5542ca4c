"""T0: the oracle against the only golden vectors the reference holds for this path:
output/final_scene.ppm and output/cornell_smoke.ppm (identical bytes: all-black 800x800 P3).
The fixture keeps the sha256 (tests/golden/make_black_800.py re-creates and checks it)."""
import hashlib
import os

import pytest

from oracle.oracle import ARITH_DEVICE, THROUGHPUT_FORM
from raytracing_rust_amd import scenes

GOLD = open(os.path.join(os.path.dirname(__file__), "golden", "black_800.sha256")).read().split()[0]


@pytest.mark.parametrize("name", ["final_scene", "cornell_smoke"])
def test_f64_oracle_reproduces_reference_ppm(orc64, name):
    cam, world = scenes.build(orc64, name, 800, 800, seed=1)
    r = orc64.render(cam, world, 800, 800, 1, seed=42)
    txt = orc64.ppm_text(r["rgb"])
    assert len(txt) == 3840015
    assert hashlib.sha256(txt).hexdigest() == GOLD
    orc64.free_all()


def test_f32_contract_oracle_reproduces_reference_ppm(orc32):
    cam, world = scenes.build(orc32, "cornell_smoke", 800, 800, seed=1)
    r = orc32.render(cam, world, 800, 800, 1, seed=7, flags=ARITH_DEVICE | THROUGHPUT_FORM)
    assert hashlib.sha256(orc32.ppm_text(r["rgb"])).hexdigest() == GOLD
    orc32.free_all()


def test_black_for_any_seed_and_spp(orc64):
    """The all-black result is deterministic (no emitter reachable), not a property of one seed."""
    for name in ("two_spheres", "random_spheres", "final_scene"):
        cam, world = scenes.build(orc64, name, 48, 32, seed=3)
        for seed in (1, 99):
            assert orc64.render(cam, world, 48, 32, 3, seed=seed)["rgb"].max() == 0
    orc64.free_all()


def test_native_ppm_writer_matches_create_image_format():
    import numpy as np

    from raytracing_rust_amd import ppm_p3

    rgb = np.zeros((800, 800, 3), np.uint8)
    assert hashlib.sha256(ppm_p3(rgb)).hexdigest() == GOLD
    rgb = np.array([[[0, 9, 10], [99, 100, 255]]], np.uint8)
    assert ppm_p3(rgb) == b"P3\n2 1\n255\n0 9 10\n99 100 255\n"


def test_streaming_ppm_writer_p3_and_p6(tmp_path):
    """rtmi_write_ppm: the P3 file is the create_image text byte for byte (golden sha on the
    all-black 800x800 image; a random image spanning several 1 MiB pieces), P6 = header + the raw bytes."""
    import numpy as np
    import pytest

    from raytracing_rust_amd import HostError, ppm_p3, write_ppm

    p = str(tmp_path / "a.ppm")
    write_ppm(p, np.zeros((800, 800, 3), np.uint8))
    assert hashlib.sha256(open(p, "rb").read()).hexdigest() == GOLD
    rgb = np.random.default_rng(3).integers(0, 256, (700, 901, 3), dtype=np.uint8)  # 1.9 M values, ~7 MB of text
    write_ppm(p, rgb, 3)
    assert open(p, "rb").read() == ppm_p3(rgb)
    write_ppm(p, rgb, 6)
    assert open(p, "rb").read() == b"P6\n901 700\n255\n" + rgb.tobytes()
    with pytest.raises(HostError):
        write_ppm(p, rgb, 5)
    with pytest.raises(HostError):
        write_ppm(str(tmp_path / "no_such_dir" / "a.ppm"), rgb)
