import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.synthetic_random_su3(lo, spread=0.3); q.rephase(lo,g)
ctx=q.Context(lat)
fl=np.zeros_like(g); ll=np.zeros_like(g)
for name,fn in (("hisq",lambda: q.HisqCoefs().smear(ctx,g,fl,ll)),("nhyp",lambda: q.HypCoefs().smear(ctx,g,fl))):
    fn(); ctx.timers_enable(1); ctx.timers_reset(); t=time.time(); fn(); dt=time.time()-t
    n,ms=ctx.timer("smear"); ctx.timers_enable(0)
    print(name,"wall incl PCIe",round(dt*1e3,1),"ms; staple kernels",n,"calls",round(ms,2),"ms total",round(1e3*ms/n,1),"us each",flush=True)
