import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from raytracing_rust_amd import Host, abi
import test_random_scenes as t
host = Host()
nx, ny, ns = 96, 64, 16
w = t._list_leaf_world(host)
cam = host.Camera((0.5, 1.2, 6.0), (0.3, 0.6, 0.0), (0.0, 1.0, 0.0), 40.0, nx / ny, 0.05, 6.0, 0.0, 1.0)
sc = host.lower(w)
for flags in (0, 1, 1 | 64, 9, 17):
    got = sc.render(cam, nx, ny, ns, seed=42, flags=flags, sig=True)
    print(flags, "sig nonzero", int(np.count_nonzero(got["sig"])), "mean", float(got["linear"].mean()), got["stats"])
