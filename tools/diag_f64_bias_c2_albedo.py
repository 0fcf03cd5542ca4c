"""fp32 contract oracle vs f64 literal oracle on BASELINE C2 random_spheres under the opt-in sky: the signed image-mean
offset per seed (diag_f64_bias_c2.py) and the first-hit albedo over the footprint of the biased patch (…_albedo.py).
CPU only; DESIGN.md §6 quotes the result."""
import sys, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from oracle.oracle import Oracle, SKY, ARITH_DEVICE, THROUGHPUT_FORM
import scenes_extra
nx,ny=1200,800
S={}
for prec,fl in (("f64",0),("f32",ARITH_DEVICE)):
    orc=Oracle(prec)
    mats=[];texs=[]
    class Rec:
        def __getattr__(s,n):
            f=getattr(orc,n)
            if n=="CheckerTexture":
                def g(*a,**k):
                    t=f(*a,**k); texs.append(t); return t
                return g
            return f
    cam,world=scenes_extra.build(Rec(),"random_spheres",nx,ny,seed=1)
    S[prec]=(orc,cam,world,texs[0],fl)
rng=np.random.default_rng(1)
nd=0
for k in range(3000):
    s=(585+rng.random()*10)/nx; t=((ny-1-406)+rng.random())/ny
    out={}
    for prec in ("f64","f32"):
        orc,cam,world,tex,fl=S[prec]
        ray=orc.get_ray(cam,s,t,seed=k)
        h=orc.hit(world,ray[:3],ray[3:6],time=ray[6],flags=fl)
        col=orc.tex_value(tex,h["u"],h["v"],h["p"],flags=fl)
        out[prec]=(h,col,ray)
    if abs(out["f64"][1][2]-out["f32"][1][2])>0.1:
        nd+=1
        if nd<=6:
            for prec in ("f64","f32"):
                h,col,ray=out[prec]; print(prec,"t",h["t"],"p",h["p"],"col",col)
print("differ",nd,"of 3000")
