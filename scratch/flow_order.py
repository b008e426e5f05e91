# A/B of the workgroup visiting order of k_force: QEXHIP_FORCE_MODE=1|2|3 (+ QEXHIP_ORD_Y/Z/T for mode 3)
import sys, os, time; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
lat=[int(v) for v in (sys.argv[1].split('x') if len(sys.argv)>1 else [32,32,32,32])]
lo=q.Layout(lat)
g=q.synthetic_random_su3(lo)
ctx=q.Context(lat)
q.plaq(ctx,g)
L=q.lib()
check(L.qexhip_wflow(ctx._h,1,0.01)); ctx.sync()
ctx.timers_enable(1); ctx.timers_reset()
for i in range(3): q.plaq(ctx)
check(L.qexhip_wflow(ctx._h,4,0.01)); ctx.sync()
out=[]
for name in ("plaq","staple"):
    n,ms=ctx.timer(name); out.append("%s %d x %.1f us"%(name,n,1e3*ms/max(n,1)))
pl=q.plaq(ctx)
print("mode",os.environ.get("QEXHIP_FORCE_MODE","1"),"ord",os.environ.get("QEXHIP_ORD_Y","-"),os.environ.get("QEXHIP_ORD_Z","-"),os.environ.get("QEXHIP_ORD_T","-"),"|"," ; ".join(out),"| plaq sum %.17g"%pl.sum(),flush=True)
