#!/usr/bin/env python3
"""Generates tests/golden/math_tables.npz: an INDEPENDENT check of include/rtmi_math.h (device and fp32 oracle share
that header, so a common-mode error in sin / ln / atan2 / asin would pass every device-vs-oracle test).

Per function 10 240 fp32 inputs on the argument ranges the scenes produce, with the result computed ONCE in f64 by
the C library's functions (numpy -> glibc libm, in the build container) and committed:
  sin   : CheckerTexture sin(10*p) with |p| <= 1000 (texture.rs:41: two_spheres' r = 1000 ground, final_scene's
          coordinates up to ~5000 scaled down), NoiseTexture sin(0.1*x + 5*turb) (texture.rs:68), small arguments
  ln    : ConstantMedium ln(U), U = k * 2^-24 (medium.rs:40) incl. the smallest and largest uniforms
  atan2 : get_sphere_uv atan2(n.z, n.x) of unit normals (sphere.rs:10) incl. the axes
  asin  : get_sphere_uv asin(n.y) (sphere.rs:11) incl. +-1 and tiny values
The test (tests/test_math_tables.py) asserts the host build of rtmi_math.h against these within the stated bound and,
on the GPU box, host == device bit for bit on the same inputs.  Run from the repo root: python tests/golden/make_math_tables.py"""
import os

import numpy as np

N = 10240
rng = np.random.default_rng(20261004)


def f32(a):
    return np.asarray(a, np.float64).astype(np.float32)


def main():
    out = {}
    p = rng.uniform(-1000.0, 1000.0, 4096)
    sin_x = np.concatenate([
        f32(10.0 * f32(p)),                                       # Checker: 10 * coordinate, fp32 product
        f32(0.1 * rng.uniform(-600, 600, 3072) + 5.0 * rng.uniform(0, 2.5, 3072)),  # Noise: 0.1 x + 5 turb
        f32(rng.uniform(-8, 8, 2048)),
        f32(rng.uniform(-50000, 50000, 1000)),                    # final_scene-sized coordinates x 10
        f32([0.0, -0.0, 1e-30, -1e-30, 3.14159274, -3.14159274, 1.57079637, 6.28318548, 1e-4, -1e-4, 0.5, -0.5,
             10000.0, -10000.0, 9999.999, 31415.926, 1.0, -1.0, 2.0, 3.0, 100.0, 1000.0, -1000.0, 12345.678]),
    ])
    assert len(sin_x) == N
    out["sin_x"] = sin_x
    out["sin_ref"] = np.sin(sin_x.astype(np.float64))

    k = np.concatenate([rng.integers(1, 1 << 24, N - 16), [1, 2, 3, (1 << 24) - 1, (1 << 24) - 2, 1 << 23, 1 << 12, 5,
                                                            (1 << 23) + 1, (1 << 23) - 1, 7, 11, 13, 1 << 20, 3 << 22, 12345]])
    ln_x = (k.astype(np.float64) * 2.0 ** -24).astype(np.float32)  # exact
    out["ln_x"] = ln_x
    out["ln_ref"] = np.log(ln_x.astype(np.float64))

    n = rng.normal(size=(N - 8, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n = np.concatenate([n, [[1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1], [0, 1, 0], [0, -1, 0], [0.6, 0, 0.8], [-0.6, 0, -0.8]]])
    n32 = n.astype(np.float32)
    out["atan2_y"], out["atan2_x"] = n32[:, 2].copy(), n32[:, 0].copy()
    out["atan2_ref"] = np.arctan2(n32[:, 2].astype(np.float64), n32[:, 0].astype(np.float64))
    asin_x = np.concatenate([n32[: N - 1024, 1], f32(rng.uniform(-1, 1, 1000)),
                             f32([1.0, -1.0, 0.0, -0.0, 1e-20, -1e-20, 0.5, -0.5, 0.99999994, -0.99999994, 0.70710677,
                                  0.86602540, 1e-4, -1e-4, 0.999, -0.999, 0.25, 0.75, 0.9, -0.9, 0.1, -0.1, 0.3, 0.6])])
    assert len(asin_x) == N
    out["asin_x"] = asin_x
    out["asin_ref"] = np.arcsin(asin_x.astype(np.float64))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "math_tables.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
