"""fp32 contract oracle vs f64 literal oracle on BASELINE C2 random_spheres under the opt-in sky: the signed image-mean
offset per seed (diag_f64_bias_c2.py) and the first-hit albedo over the footprint of the biased patch (…_albedo.py).
CPU only; DESIGN.md §6 quotes the result."""
import sys, numpy as np, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from oracle.parallel import render_parallel
from oracle.oracle import SKY, ARITH_DEVICE, THROUGHPUT_FORM
nx,ny,ns=1200,800,500
rows=[int((k+0.5)*ny/64) for k in range(64)]
for seed in (42,43,44):
    t=time.time()
    a=render_parallel("scenes_extra","random_spheres",nx,ny,ns,seed,ARITH_DEVICE|THROUGHPUT_FORM|SKY,precision="f32",rows=rows,workers=8,timeout=3000)
    b=render_parallel("scenes_extra","random_spheres",nx,ny,ns,seed,SKY,precision="f64",rows=rows,workers=8,timeout=3000)
    d=a["linear"][rows].astype(np.float64)-b["mean"][rows]
    m=b["mean"][rows].mean()
    print(seed, "rel signed mean diff %.3e"%(d.mean()/m), "upper half %.3e lower half %.3e"%(d[:32].mean()/b["mean"][rows][:32].mean(), d[32:].mean()/b["mean"][rows][32:].mean()), "t=%.0fs"%(time.time()-t), flush=True)
