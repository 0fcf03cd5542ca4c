"""Signed image-mean offset of the fp32-contract oracle against the f64 literal oracle, three seeds, for the lit scenes of
tests/test_gpu_f64_tolerance.py (CPU only, about ten minutes on 8 cores); DESIGN.md §6 quotes the result."""
import sys, numpy as np, time
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from oracle.parallel import render_parallel
from oracle.oracle import SKY, ARITH_DEVICE, THROUGHPUT_FORM
for name,nx,ny,ns,rows in (("cornell_box",800,800,1000,[int((k+0.5)*800/64) for k in range(64)]),
                            ("lit_smoke",800,800,1000,[int((k+0.5)*800/64) for k in range(64)]),
                            ("lit_final_scene",480,270,1000,list(range(0,270,2)))):
    for seed in (42,43,44):
        t=time.time()
        a=render_parallel("scenes_extra",name,nx,ny,ns,seed,ARITH_DEVICE|THROUGHPUT_FORM,precision="f32",rows=rows,workers=8,timeout=3000)
        b=render_parallel("scenes_extra",name,nx,ny,ns,seed,0,precision="f64",rows=rows,workers=8,timeout=3000)
        d=a["linear"][rows].astype(np.float64)-b["mean"][rows]
        m=b["mean"][rows].mean()
        flat=np.abs(d).sum(axis=2).ravel(); top=np.argsort(flat)[::-1][:200]
        print(name,seed,"rel signed mean diff %.3e"%(d.mean()/m),"share of signed sum in top-200 pixels %.2f"%(d.sum(axis=2).ravel()[top].sum()/d.sum()),"t=%.0fs"%(time.time()-t),flush=True)
