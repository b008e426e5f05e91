# time per site of the fused RK3 stage against the lattice size (what level of the memory hierarchy pays for it?)
import sys, time, os; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L=q.lib()
for lat in ([16,16,16,16],[32,16,16,16],[32,32,16,16],[32,32,32,16],[32,32,32,32],[32,32,32,64]):
    lo=q.Layout(lat)
    g=q.RngField(lat,q.RngMilc6,987654321).random()
    ctx=q.Context(lat)
    q.gaugeSet(ctx,g)
    check(L.qexhip_wflow(ctx._h,1,0.01)); ctx.sync()
    ctx.timers_enable(1); ctx.timers_reset()
    check(L.qexhip_wflow(ctx._h,3,0.01)); ctx.sync()
    n,ms=ctx.timer("staple"); n2,ms2=ctx.timer("expupdate")
    us=1e3*ms/max(n,1); us2=1e3*ms2/max(n2,1)
    print("mode %s fused %s lat %s: stage %.1f us (+exp %.1f) = %.3f ns/site, links %.0f MB" % (os.environ.get("QEXHIP_FORCE_MODE","3"), os.environ.get("QEXHIP_FLOW_FUSED","1"), lat, us, us2, 1e3*(us+us2)/lo.vol, lo.vol*576/1e6), flush=True)
    ctx.close()
