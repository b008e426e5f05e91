import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[int(v) for v in sys.argv[1:5]] if len(sys.argv)>4 else [32,32,32,32]; lo=q.Layout(lat)
g=q.synthetic_random_su3(lo, spread=0.3); q.rephase(lo,g)
ctx=q.Context(lat); s=q.newStag(ctx,g)
b=q.synthetic_gaussian_vector(lo,seed=3); bid=ctx.field_new(b); xid=ctx.field_new()
K=640
for gr in (1,0,1,0):
    ctx.set_option("graph",gr)
    ctx.dev_solve_xx(xid,bid,0.1,0.0,64,True); ctx.sync()
    t=time.perf_counter(); its,fin,_=ctx.dev_solve_xx(xid,bid,0.1,0.0,K,True); ctx.sync(); dt=time.perf_counter()-t
    print("graph",gr,"its",its,"us/iter %.2f"%(dt/K*1e6),"r2",fin,flush=True)
