import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q, ctypes as C
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.synthetic_random_su3(lo); q.rephase(lo,g); b=q.synthetic_gaussian_vector(lo)
ctx=q.Context(lat); s=q.newStag(ctx,g)
masses=[float(np.sqrt(k+2))*0.05 for k in range(10)]   # 10 shifts (stagSolve.nim:598 pattern, scaled)
shifts=[masses[0]]+[4*(m*m-masses[0]**2) for m in masses[1:]]
xs=[np.zeros_like(b) for _ in masses]
sp=q.SolverParams(r2req=1e-10,maxits=5000,verbosity=0)
t=time.time(); s.solveXX_multi(xs,b,shifts,sp,parEven=True,histcap=8); dt=time.time()-t
print("multishift 10 masses: its",sp.iterations,"wall",dt,"s  (incl. 10 downloads) ->", dt/sp.iterations*1e6,"us/iter upper bound")
# check each shifted solution with the single-mass operator on GPU
h=lo.vol//2
for k in (0,5,9):
    r=np.zeros_like(b); s.stagD2ee(r,xs[k],masses[k]**2); res=r[:h]-b[:h]
    print(" mass",masses[k],"rel res", np.sqrt((res*res).sum()/(b[:h]*b[:h]).sum()))
