# two Wilson-flow steps + plaquettes on 32^4: timings of k_force (fused stage) and k_plaq; run under rocprofv3 for profiles/
import sys, time; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
lat=[32,32,32,32]
lo=q.Layout(lat)
t=time.time(); g=q.RngField(lat,q.RngMilc6,987654321).random(); print("gen",time.time()-t,flush=True)   # QEX g.random
ctx=q.Context(lat)
t=time.time(); pl=q.plaq(ctx,g); print("gpu plaq incl upload",time.time()-t, pl, flush=True)
ctx.timers_enable(1); ctx.timers_reset()
for i in range(5): q.plaq(ctx)
L=q.lib()
check(L.qexhip_wflow(ctx._h,1,0.01)); ctx.sync(); ctx.timers_reset()
t=time.time(); check(L.qexhip_wflow(ctx._h,4,0.01)); ctx.sync(); print("4 flow steps wall",time.time()-t,flush=True)
for name in ("plaq","staple","expupdate"):
    n,ms=ctx.timer(name); print(name,n,"calls avg us",1e3*ms/max(n,1))
pl2=q.plaq(ctx); print("plaq after flow",pl2)
# size-independent property at full size (parity at small sizes is the job of tests/): unitarity preserved
g2=np.zeros_like(g); check(L.qexhip_gauge_get(ctx._h, g2.ctypes.data_as(__import__('ctypes').c_void_p)))
m=(g2[...,0]+1j*g2[...,1]).reshape(-1,3,3)[::1000]
print("unitarity dev", np.abs(np.einsum('nij,nkj->nik',m,m.conj())-np.eye(3)).max(), "det dev", np.abs(np.linalg.det(m)-1).max())
