import sys, time; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from oracle import oracle as o
lat=[32,32,32,32]
lo=o.Layout(lat)
t=time.time(); g=o.gauge_random(lo, seed=987654321); print("gen",time.time()-t,flush=True)
ctx=q.Context(lat)
t=time.time(); pl=q.plaq(ctx,g); print("gpu plaq incl upload",time.time()-t, pl, flush=True)
t=time.time(); plo=o.plaq(lo,g); print("oracle plaq",time.time()-t, np.abs(pl-plo).max(), flush=True)
ctx.timers_enable(1); ctx.timers_reset()
for i in range(5): q.plaq(ctx)
L=q.lib()
from qex_amd._lib import check
t=time.time(); check(L.qexhip_wflow(ctx._h,2,0.01)); ctx.sync(); print("2 flow steps wall",time.time()-t,flush=True)
for name in ("plaq","staple","expupdate"):
    n,ms=ctx.timer(name); print(name,n,"calls avg us",1e3*ms/max(n,1))
pl2=q.plaq(ctx); print("plaq after flow",pl2)
# oracle check of one flow step at full size would take long; check size-independent property: unitarity preserved
g2=np.zeros_like(g); check(L.qexhip_gauge_get(ctx._h, g2.ctypes.data_as(__import__('ctypes').c_void_p)))
m=(g2[...,0]+1j*g2[...,1]).reshape(-1,3,3)[::1000]
print("unitarity dev", np.abs(np.einsum('nij,nkj->nik',m,m.conj())-np.eye(3)).max(), "det dev", np.abs(np.linalg.det(m)-1).max())
