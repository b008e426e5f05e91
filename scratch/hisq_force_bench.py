import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.synthetic_random_su3(lo, spread=0.3); q.rephase(lo,g)
ctx=q.Context(lat)
dfl=q.RngField(lat,q.RngMilc6,5).randomTAH(); dll=0.1*dfl
hc=q.HisqCoefs().init()
f=hc.force(ctx,g,dfl,dll)
ctx.timers_enable(1); ctx.timers_reset()
t=time.time(); f=hc.force(ctx,g,dfl,dll); dt=time.time()-t
print("hisq force wall incl PCIe %.1f ms; staple-deriv kernels %s; staple kernels %s; finite %s"%(dt*1e3, ctx.timer("smear_deriv"), ctx.timer("smear"), np.isfinite(f).all()))
