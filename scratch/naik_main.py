import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.unit(lo); rng=np.random.default_rng(1); g+=0.1*rng.standard_normal(g.shape); g3=0.3*g
ctx=q.Context(lat)
x=q.synthetic_gaussian_vector(lo)
for naik in (0,1):
    s=q.newStag3(ctx,g,g3) if naik else q.newStag(ctx,g)
    xi=ctx.field_new(x); ri=ctx.field_new()
    for (a,b) in ((0,0),(0,0.4)):
        for par in (0,1):
            for i in range(3): ctx.dev_dslash(ri,xi,par,a,b)
            ctx.sync(); ctx.timers_enable(1); ctx.timers_reset()
            for i in range(30): ctx.dev_dslash(ri,xi,par,a,b)
            ctx.sync(); n,ms=ctx.timer("dslash"); ctx.timers_enable(0)
            print("naik",naik,"a,b",a,b,"par",par,"avg us",1e3*ms/n,flush=True)
    # op_xx (two sweeps, t intermediate)
    for i in range(3): ctx.dev_op_xx(ri,xi,0.01,True)
    ctx.sync(); ctx.timers_enable(1); ctx.timers_reset()
    for i in range(30): ctx.dev_op_xx(ri,xi,0.01,True)
    ctx.sync(); n,ms=ctx.timer("dslash"); ctx.timers_enable(0)
    print("naik",naik,"op_xx sweeps avg us",1e3*ms/n,flush=True)
