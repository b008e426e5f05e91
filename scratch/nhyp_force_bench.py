import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.RngField(lat,q.RngMilc6,987654321).warm(0.5)   # QEX g.warm(0.5)
ctx=q.Context(lat)
fl=np.zeros_like(g); f=np.zeros_like(g)
hc=q.HypCoefs(0.4,0.5,0.5)
t=time.time(); sf=hc.smearGetForce(ctx,g,fl); print("prepare (cold, allocs)",round((time.time()-t)*1e3,1),"ms",flush=True)
ctx.timers_enable(1); ctx.timers_reset()
t=time.time(); sf=hc.smearGetForce(ctx,g,fl); print("prepare wall incl PCIe",round((time.time()-t)*1e3,1),"ms; staple kernels",ctx.timer("smear"),flush=True)
ctx.timers_enable(0)
psis=[q.synthetic_gaussian_vector(lo,seed=5+k) for k in range(2)]
for rep in range(2):
    ctx.timers_enable(1); ctx.timers_reset()
    t=time.time(); sf.gforce(f, plaq=1.0); dt=time.time()-t
    n,ms=ctx.timer("nhyp_force"); n2,ms2=ctx.timer("staple")
    print("gforce wall",round(dt*1e3,1),"ms; chain",round(ms,2),"ms; deriv",round(ms2,3),"ms",flush=True)
    ctx.timers_reset()
    t=time.time(); sf.fforce(f, psis, [1.0,0.5]); dt=time.time()-t
    n,ms=ctx.timer("nhyp_force"); n2,ms2=ctx.timer("outer")
    print("fforce(2 fields) wall",round(dt*1e3,1),"ms; chain",round(ms,2),"ms; outer",n2,round(ms2,3),"ms",flush=True)
    ctx.timers_enable(0)
print("finite:", np.isfinite(f).all())
