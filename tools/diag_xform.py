#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_rust_amd import abi
lib = abi.load_rtmi()
rng = np.random.default_rng(1)
n = 1 << 16
x = rng.normal(size=n).astype(np.float32) * 5; y = rng.normal(size=n).astype(np.float32) * 5
s, c = np.float32(0.5446390509605408), np.float32(0.838670551776886)
exp = {0: c * x + s * y, 1: (-s) * x + c * y, 2: c * x - s * y, 3: s * x + c * y}
for BASE in (16, 28):
 for axis in range(3):
  for which in (range(4) if BASE == 16 else (2, 3)):
    out = np.zeros(n, np.float32)
    rc = lib.rtmi_probe_math(BASE + 4 * axis + which, x.ctypes.data, y.ctypes.data, out.ctypes.data, n)
    bad = int((out != exp[which]).sum())
    print(BASE, "axis", "XYZ"[axis], "which", which, "rc", rc, "mismatches", bad, "max ulp-ish", float(np.abs(out - exp[which]).max()))
