import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.synthetic_random_su3(lo, spread=0.3); q.rephase(lo,g)
ctx=q.Context(lat); s=q.newStag(ctx,g); print("links",s.links_info(),flush=True)
ms=[0.1,0.2,0.4,0.05]
bs=[q.synthetic_gaussian_vector(lo,seed=11+k) for k in range(4)]
for b in bs: b[lo.vol//2:]=0
K=200
for n in (1,2,3,4):
    xs=[np.zeros_like(b) for b in bs[:n]]
    s.solveXX_batch(xs,bs[:n],ms[:n],0.0,10,True)           # warm up / allocations
    ctx.timers_enable(1); ctx.timers_reset()
    t=time.time(); its,_=s.solveXX_batch(xs,bs[:n],ms[:n],0.0,K,True); dt=time.time()-t
    nd,msd=ctx.timer("dslash_batch"); nb,msb=ctx.timer("blas"); ctx.timers_enable(0)
    print("n=%d: %.1f us per iteration of all systems (%.1f us per system-iteration); sweep %.1f us; blas %.1f us/iter; wall incl PCIe %.1f ms"%(n,1e3*(msd+msb)/K,1e3*(msd+msb)/K/n,1e3*msd/nd,1e3*msb/K,dt*1e3),flush=True)
sp=q.SolverParams(r2req=0.0,maxits=K,verbosity=0); x=np.zeros_like(bs[0])
s.solveXX(x,bs[0],0.1,sp,True); t=time.time(); s.solveXX(x,bs[0],0.1,sp,True); print("single solveXX wall %.1f ms for %d its"%((time.time()-t)*1e3,K))
