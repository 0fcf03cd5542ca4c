"""An independent check of include/rtmi_math.h.  Device and fp32 oracle share that header, so a
common-mode error in sin / ln / atan2 / asin would pass every device-vs-oracle comparison; here both are held against
tests/golden/math_tables.npz: 10 240 fp32 inputs per function on the argument ranges the scenes produce, with results
computed once in f64 by the C library (tests/golden/make_math_tables.py, committed with the table).

Reference call sites: src/texture.rs:41 (Checker, sin(10 p)), :68 (Noise, sin(scale x + 5 turb)), src/medium.rs:40
(ln U), src/sphere.rs:10-11 (atan2, asin of the unit normal).

Bounds asserted, in units of the last place of the correctly rounded fp32 result (measured maxima in brackets):
sin 1.5 (1.43), ln 1.0 (0.73), atan2 3.0 (2.64), asin 3.0 (2.23) — for sin / atan2 / asin that is |error| <= 3e-7 absolute,
for ln <= 1 ulp of a value up to 16.6 (1e-6): orders of magnitude inside the 1e-4 radiance tolerance of the north star.  On the GPU box: host == device BIT FOR BIT on the
same inputs (so the bounds hold for the device as well, which is then asserted directly, too)."""
import os

import numpy as np
import pytest

from raytracing_rust_amd import abi

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "math_tables.npz")
FUNCS = [  # name, probe op, oracle symbol, argument keys, ulp bound
    ("sin", 0, "orc_rtmi_sinf", ("sin_x",), 1.5),
    ("ln", 1, "orc_rtmi_logf", ("ln_x",), 1.0),
    ("atan2", 2, "orc_rtmi_atan2f", ("atan2_y", "atan2_x"), 3.0),
    ("asin", 3, "orc_rtmi_asinf", ("asin_x",), 3.0),
]


def _host(orc32, sym, args):
    fn = getattr(orc32.lib, sym)
    return np.array([fn(*[float(a[i]) for a in args]) for i in range(len(args[0]))], np.float32)


def _ulp_error(got, ref64):
    ulp = np.maximum(np.spacing(np.abs(ref64.astype(np.float32))).astype(np.float64), 2.0 ** -149)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_table_is_what_the_generator_says():
    z = np.load(TABLE)
    for name, _, _, keys, _ in FUNCS:
        for k in keys:
            assert z[k].dtype == np.float32 and z[k].shape == (10240,)
        assert z[name + "_ref"].dtype == np.float64 and np.isfinite(z[name + "_ref"]).all()
    # ranges the scenes use are present: Checker's 10 * p up to |p| = 1000, the smallest and the largest uniform
    assert np.abs(z["sin_x"]).max() >= 10000.0 and (z["ln_x"] == np.float32(2.0 ** -24)).any()
    assert (z["ln_x"] == np.float32(1.0 - 2.0 ** -24)).any() and (np.abs(z["asin_x"]) == 1.0).any()
    # the committed references are libm's (recomputed here; glibc's f64 functions are correctly rounded or within 1 ulp
    # of f64, i.e. exact for this purpose): a corrupted table fails here, not in the accuracy test
    assert np.allclose(z["sin_ref"], np.sin(z["sin_x"].astype(np.float64)), rtol=0, atol=1e-15)
    assert np.allclose(z["ln_ref"], np.log(z["ln_x"].astype(np.float64)), rtol=1e-15, atol=0)


@pytest.mark.parametrize("name,op,sym,keys,bound", FUNCS, ids=[f[0] for f in FUNCS])
def test_host_build_of_rtmi_math_against_libm_table(orc32, name, op, sym, keys, bound):
    z = np.load(TABLE)
    got = _host(orc32, sym, [z[k] for k in keys])
    err = _ulp_error(got, z[name + "_ref"])
    worst = int(np.argmax(err))
    print("%s: max %.3f ulp at input %s, max abs error %.3g" % (name, err.max(), [float(z[k][worst]) for k in keys],
                                                                 np.abs(got.astype(np.float64) - z[name + "_ref"]).max()))
    assert err.max() <= bound, "%s: %.3f ulp at %s" % (name, err.max(), [float(z[k][worst]) for k in keys])
    if name != "ln":
        assert np.abs(got.astype(np.float64) - z[name + "_ref"]).max() <= 3e-7


@pytest.mark.gpu
@pytest.mark.parametrize("name,op,sym,keys,bound", FUNCS, ids=[f[0] for f in FUNCS])
def test_device_equals_host_on_the_table_and_meets_the_same_bound(orc32, name, op, sym, keys, bound):
    lib = abi.load_rtmi()
    z = np.load(TABLE)
    args = [np.ascontiguousarray(z[k]) for k in keys]
    out = np.zeros(10240, np.float32)
    rc = lib.rtmi_probe_math(op, args[0].ctypes.data, args[1].ctypes.data if len(args) > 1 else None, out.ctypes.data, 10240)
    assert rc == 0, lib.rtmi_last_error()
    host = _host(orc32, sym, args)
    assert np.array_equal(out.view(np.uint32), host.view(np.uint32)), "%s: %d inputs differ" % (
        name, int((out.view(np.uint32) != host.view(np.uint32)).sum()))
    assert _ulp_error(out, z[name + "_ref"]).max() <= bound
