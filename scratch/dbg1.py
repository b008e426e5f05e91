import sys, time; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from oracle import oracle as o
def relerr(a,b): return np.linalg.norm((a-b).ravel())/np.linalg.norm(b.ravel())
lat=[8,8,8,8]
lo=o.Layout(lat); rf=o.RngField(lo,o.RNG_MILC6,987654321)
g=o.gauge_random(lo,rf); o.rephase(lo,g)
g3=o.gauge_random(lo,rf); o.rephase(lo,g3); g3*=0.3
x=o.vector_gaussian(lo,rf)
h=lo.vol//2
for naik in (0,1):
    ctx=q.Context(lat); ctx.force_halo(True)
    s=q.newStag3(ctx,g,g3) if naik else q.newStag(ctx,g)
    gg3=g3 if naik else None
    r=np.zeros_like(x); s.stagD2ee(r,x,0.01); ref=o.stagD2xx(lo,g,gg3,x,0.01,True)
    print("naik",naik,"halo stagD2ee err",relerr(r[:h],ref[:h]),flush=True)
    for it in (1,2,3,5,10):
        sp=q.SolverParams(r2req=1e-12,maxits=it,verbosity=0); xx=np.zeros_like(x)
        s.solveEE(xx,x,0.1,sp,histcap=64)
        xr,its,fin,hist=o.solveXX(lo,g,gg3,x,0.1,1e-12,it,True,histcap=64)
        print("  its",it,sp.iterations,"hist dev",np.abs(sp.r2hist/hist-1).max(),"x err",relerr(xx,xr),flush=True)
print("rccl self test",flush=True)
t=time.time(); uid=q.Context.unique_id(); print("uid",time.time()-t,flush=True)
ctx=q.Context(lat)
t=time.time(); ctx.comm_init(uid,1,0); print("comm_init",time.time()-t,flush=True)
ctx.force_halo(True)
t=time.time(); s=q.newStag(ctx,g); print("set_links",time.time()-t,flush=True)
r=np.zeros_like(x); t=time.time(); s.D(r,x,0.1); print("D",time.time()-t, relerr(r,o.D(lo,g,None,x,0.1)),flush=True)
