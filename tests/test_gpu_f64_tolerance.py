"""The north star's "stated fp32 radiance tolerance", stated against the REFERENCE'S OWN ARITHMETIC at BASELINE size.

The device computes in fp32 under the contract of DESIGN.md §4 and is bit-identical to the fp32 oracle
(test_gpu_oracle_million.py).  What the reference computes is f64 (`Vector3<f64>`, src/color.rs:6-23): its literal
restatement is the f64 oracle with flags 0 (recursive color, divisions where the reference divides).  Both sides take
the same Philox streams (24-bit uniforms are exact in either precision), so a camera path differs only where fp32
rounding moves a decision — a hit/miss at a grazing edge, a rejection-sampler trial at the unit sphere's surface, a
dielectric's reflect/refract draw — after which that ONE path of the pixel's ns goes elsewhere and the pixel mean moves
by (radiance of the path) / ns.  This test measures and bounds that on
  * BASELINE C3 cornell_box 800x800x1000 spp (the one lit BASELINE config), 64 evenly spaced rows = 51.2 M paths,
  * lit_final_scene 480x270x1000 spp (C5's object graph with the light the right way round), every 2nd row = 64.8 M paths,
  * lit_smoke 800x800x1000 spp (C4's object graph and size with the back wall where the light can reach it), 64 rows = 51.2 M paths,
  * BASELINE C2 random_spheres 1200x800x500 spp under the opt-in sky (the scene has no emitter), 64 rows = 38.4 M paths,
in forked oracle workers (oracle/parallel.py).  The measured figures are printed, asserted with some slack, and quoted
in DESIGN.md §6.  Reference loop: tests/test.rs:62-78."""
import json
import os
import time

import numpy as np
import pytest

from oracle.oracle import SKY
from oracle.parallel import render_parallel
from raytracing_rust_amd import abi

import scenes_extra

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compare(host, name, nx, ny, ns, rows, sky=False):
    cam, world = scenes_extra.build(host, name, nx, ny, seed=1)
    sc = host.lower(world).upload(0)
    got = sc.render(cam, nx, ny, ns, seed=42, flags=abi.RTMI_FLAG_FAST_CULL | (abi.RTMI_FLAG_SKY if sky else 0))
    t0 = time.perf_counter()
    ref = render_parallel("scenes_extra", name, nx, ny, ns, 42, SKY if sky else 0, precision="f64", rows=rows, timeout=1500)
    dt = time.perf_counter() - t0
    d = np.abs(got["linear"][rows].astype(np.float64) - ref["mean"][rows])  # linear radiance, per channel
    lev = np.abs(got["rgb8"][rows].astype(np.int32) - ref["rgb"][rows])     # PPM values (0..255)
    mean_ref = float(ref["mean"][rows].mean())
    res = {
        "scene": "%s %dx%dx%d" % (name, nx, ny, ns), "rows": len(rows), "paths": len(rows) * nx * ns, "oracle_s": round(dt, 1),
        "mean_radiance_f64": mean_ref,
        "image_mean_rel_err": abs(float(got["linear"][rows].astype(np.float64).mean()) - mean_ref) / mean_ref,
        "share_within_1e-4": float((d <= 1e-4).mean()), "share_within_1e-3": float((d <= 1e-3).mean()),
        "share_within_1e-2": float((d <= 1e-2).mean()), "share_identical_to_f32_rounding": float((d <= 1e-6 * np.maximum(1.0, ref["mean"][rows])).mean()),
        "max_abs": float(d.max()), "mean_abs": float(d.mean()), "p99_abs": float(np.quantile(d, 0.99)),
        "ppm_values_differing": float((lev > 0).mean()), "ppm_values_differing_by_more_than_1": float((lev > 1).mean()),
        "ppm_max_level_diff": int(lev.max()),
    }
    print(json.dumps(res))
    out = os.path.join(ROOT, "gpurun_out", "f64_tolerance_%s.json" % name)
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(res, open(out, "w"), indent=1)
    except OSError:
        pass
    return res


def test_c3_cornell_box_fullsize_against_the_f64_literal(host):
    nx, ny, ns = 800, 800, 1000
    rows = [int((k + 0.5) * ny / 64) for k in range(64)]
    r = compare(host, "cornell_box", nx, ny, ns, rows)
    assert r["mean_radiance_f64"] > 0.05
    # measured (MI355X, r04): see DESIGN.md §6 "Stated tolerance against the reference's f64 arithmetic"
    # (r04: 98.5 % of the channels within 1e-4, 99.3 % within 1e-3, mean |d| 4.7e-5, max 0.045; 0.78 % of the PPM values
    # differ, 0.38 % by more than one level, at most 10 levels; image mean 2.1e-5 relative)
    assert r["image_mean_rel_err"] <= 2e-4
    assert r["share_within_1e-4"] >= 0.97 and r["share_within_1e-3"] >= 0.985 and r["share_within_1e-2"] >= 0.995
    assert r["mean_abs"] <= 2e-4 and r["max_abs"] <= 0.15
    assert r["ppm_values_differing"] <= 0.02 and r["ppm_values_differing_by_more_than_1"] <= 0.01 and r["ppm_max_level_diff"] <= 20


def test_lit_final_scene_against_the_f64_literal(host):
    nx, ny, ns = 480, 270, 1000
    rows = list(range(0, ny, 2))
    r = compare(host, "lit_final_scene", nx, ny, ns, rows)
    assert r["mean_radiance_f64"] > 0.01
    # (r04, with the sphere discriminant of contract substitution 5: image mean 4.3e-5 relative, 87.7 % of the channels within
    # 1e-4, 95.2 % within 1e-3, 99.95 % within 1e-2, mean |d| 1.8e-4, max 0.019; 5.0 % of the PPM values differ, 1.0 % by more
    # than one level, at most 19.  With the literal b*b - a*c the same comparison gave: image mean 0.55 % LOW, the small
    # far spheres 10 % too dark, 17 % of the PPM values different — which is what this test exists to catch)
    assert r["image_mean_rel_err"] <= 5e-4
    assert r["share_within_1e-4"] >= 0.80 and r["share_within_1e-3"] >= 0.92 and r["share_within_1e-2"] >= 0.995
    assert r["mean_abs"] <= 5e-4 and r["max_abs"] <= 0.1
    assert r["ppm_values_differing"] <= 0.10 and r["ppm_values_differing_by_more_than_1"] <= 0.03 and r["ppm_max_level_diff"] <= 40


def test_lit_smoke_fullsize_against_the_f64_literal(host):
    """BASELINE C4's object graph and size (cornell_smoke 800x800x1000; the reference's own scene renders black: its back wall
    sits at z = 0, tests/test.rs:369-377 — here at z = 555): the two ConstantMedium boxes put the fp32 logarithm and the
    distance arithmetic of medium.rs:38-44 on most paths.  64 evenly spaced rows = 51.2 M paths."""
    nx, ny, ns = 800, 800, 1000
    rows = [int((k + 0.5) * ny / 64) for k in range(64)]
    r = compare(host, "lit_smoke", nx, ny, ns, rows)
    assert r["mean_radiance_f64"] > 0.3
    # measured (MI355X, r04): image mean 1.8e-5 relative; 92.2 % of the channels within 1e-4, 96.5 % within 1e-3, 99.996 %
    # within 1e-2; mean |d| 1.2e-4, max 0.0102; 2.6 % of the PPM values differ, 0.18 % by more than one level, at most 6
    assert r["image_mean_rel_err"] <= 2e-4
    assert r["share_within_1e-4"] >= 0.88 and r["share_within_1e-3"] >= 0.94 and r["share_within_1e-2"] >= 0.999
    assert r["mean_abs"] <= 4e-4 and r["max_abs"] <= 0.05
    assert r["ppm_values_differing"] <= 0.06 and r["ppm_values_differing_by_more_than_1"] <= 0.01 and r["ppm_max_level_diff"] <= 16


def test_c2_random_spheres_under_the_sky_fullsize_against_the_f64_literal(host):
    """BASELINE C2 random_spheres 1200x800x500 spp — the scene has no emitter (the reference renders it black), so under the
    opt-in sky gradient (color.rs:18-20, commented out in the reference; the same gradient in the oracle): the BVH of 485
    spheres, glass and metal on most paths.  64 evenly spaced rows = 38.4 M paths."""
    nx, ny, ns = 1200, 800, 500
    rows = [int((k + 0.5) * ny / 64) for k in range(64)]
    r = compare(host, "random_spheres", nx, ny, ns, rows, sky=True)
    assert r["mean_radiance_f64"] > 0.1
    # measured (MI355X, r04): image mean 7.6e-5 relative; 75.0 % of the channels within 1e-4, 95.1 % within 1e-3, 99.94 % within
    # 1e-2; mean |d| 1.9e-4, 99th percentile 1.9e-3, max 0.11; 4.1 % of the PPM values differ, 0.08 % by more than one level, at
    # most 33 (500 spp: one flipped path moves a pixel twice as far as at 1000).  The image-mean figure is systematic, not noise:
    # the checker ground within 0.3 of the origin (radius-1000 sphere: its fp32 hit point's y is wrong by more than its size
    # there, the checker's sin(10 y) flips) — DESIGN.md §6
    assert r["image_mean_rel_err"] <= 4e-4
    assert r["share_within_1e-4"] >= 0.68 and r["share_within_1e-3"] >= 0.92 and r["share_within_1e-2"] >= 0.995
    assert r["mean_abs"] <= 6e-4 and r["max_abs"] <= 0.5
    assert r["ppm_values_differing"] <= 0.09 and r["ppm_values_differing_by_more_than_1"] <= 0.005 and r["ppm_max_level_diff"] <= 80
