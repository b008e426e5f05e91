import sys, time; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from oracle import oracle as o
def relerr(a,b): return np.linalg.norm((a-b).ravel())/np.linalg.norm(b.ravel())
lat=[8,8,8,8]
lo=o.Layout(lat); rf=o.RngField(lo,o.RNG_MILC6,987654321)
g=o.gauge_random(lo,rf); o.rephase(lo,g)
g3=o.gauge_random(lo,rf); o.rephase(lo,g3); g3*=0.3
x=o.vector_gaussian(lo,rf); y=o.vector_gaussian(lo,rf)
h=lo.vol//2
ctx=q.Context(lat); ctx.force_halo(True)
s=q.newStag3(ctx,g,g3)
for (a,b) in ((0,0),(0,0.4),(1.5,-0.7)):
  for sub,par in (("even",0),("odd",1)):
    r=y.copy(); s.stagD2(r,x,sub,a,b); ref=y.copy(); o.stagD2(lo,g,g3,ref,x,par,a,b)
    print("stagD2",a,b,sub,relerr(r,ref),flush=True)
# two-step through host
t1=np.zeros_like(x); s.stagD2(t1,x,"odd",0,0)
r2=np.zeros_like(x); s.stagD2(r2,t1,"even",0,0)
ref1=np.zeros_like(x); o.stagD2(lo,g,g3,ref1,x,1,0,0); ref2=np.zeros_like(x); o.stagD2(lo,g,g3,ref2,ref1,0,0,0)
print("two-step host",relerr(t1,ref1),relerr(r2,ref2),flush=True)
# device two-step
xi=ctx.field_new(x); ti=ctx.field_new(); ri=ctx.field_new()
ctx.dev_dslash(ti,xi,1,0,0); ctx.dev_dslash(ri,ti,0,0,0); ctx.sync()
td=ctx.field_download(ti); rd=ctx.field_download(ri)
print("two-step dev",relerr(td,ref1),relerr(rd,ref2),flush=True)
d=(rd-ref2)[:h]; bad=np.where(np.abs(d).sum(axis=(1,2))>1e-10)[0]
print("bad sites",len(bad),"t coords",sorted(set(lo.coord(int(i))[3] for i in bad)))
r=np.zeros_like(x); s.stagD2ee(r,x,0.01); ref=o.stagD2xx(lo,g,g3,x,0.01,True)
d=(r-ref)[:h]; bad=np.where(np.abs(d).sum(axis=(1,2))>1e-10)[0]
print("ee err",relerr(r[:h],ref[:h]),"bad sites",len(bad),"t coords",sorted(set(lo.coord(int(i))[3] for i in bad)))
