# visiting-order sweep of the gather kernels at 32^4: one process per (RZ, Y, Z, T) (the table is built once per context)
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L = q.lib()
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
q.gaugeSet(ctx, g)
check(L.qexhip_wflow(ctx._h, 1, 0.01))
res = []
for rnd in range(3):
    ctx.timers_enable(1); ctx.timers_reset()
    check(L.qexhip_wflow(ctx._h, 4, 0.01)); ctx.sync()
    n, ms = ctx.timer("staple")
    res.append(1e3 * ms / n)
cfg = " ".join("%s=%s" % (k[11:], os.environ.get(k, "-")) for k in ("QEXHIP_ORD_RZ", "QEXHIP_ORD_Y", "QEXHIP_ORD_Z", "QEXHIP_ORD_T"))
print("order", cfg, "stage us:", " ".join("%.1f" % r for r in res), "plaq %.12f" % q.plaq(ctx).sum(), flush=True)
ctx.close()
