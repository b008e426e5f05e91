import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, qex_amd as q
from oracle import oracle as o
lat=[8,4,6,4]; lo=o.Layout(lat); rf=o.RngField(lo,o.RNG_MILC6,4711)
for kind in ("random","warm"):
    g=o.gauge_random(lo,rf) if kind=="random" else o.gauge_warm(lo,0.5,rf); o.rephase(lo,g)
    ctx=q.Context(lat); s=q.newStag(ctx,g)
    b=o.vector_gaussian(lo,rf)
    for k in (1,2,3,10,50):
        x=np.zeros_like(b); its,fin=s.solveXX_batch([x],[b],[0.1],0.0,k,True)
        sp=q.SolverParams(r2req=0.0,maxits=k,verbosity=0); x1=np.zeros_like(b); s.solveXX(x1,b,0.1,sp,True)
        print(kind,k,its,sp.iterations,fin[0],sp.r2, np.abs(x-x1).max(), np.array_equal(x,x1))
print("determinism")
xa=np.zeros_like(b); sp=q.SolverParams(r2req=0.0,maxits=50,verbosity=0); s.solveXX(xa,b,0.1,sp,True)
xb=np.zeros_like(b); sp=q.SolverParams(r2req=0.0,maxits=50,verbosity=0); s.solveXX(xb,b,0.1,sp,True)
print("single twice", np.array_equal(xa,xb))
xc=np.zeros_like(b); s.solveXX_batch([xc],[b],[0.1],0.0,50,True)
xd=np.zeros_like(b); s.solveXX_batch([xd],[b],[0.1],0.0,50,True)
print("batch twice", np.array_equal(xc,xd))
b2=o.vector_gaussian(lo,rf)
xe=np.zeros_like(b); xf=np.zeros_like(b); s.solveXX_batch([xe,xf],[b,b2],[0.1,0.3],0.0,50,True)
print("batch n=2 vs n=1", np.array_equal(xe,xc))
