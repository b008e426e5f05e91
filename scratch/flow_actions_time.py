# time per RK3 stage of the three flow actions of src/flow/flow.nim (Wilson / rect (Symanzik) / adjoint) at 32^4, warm
import sys, time
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L = q.lib()
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
q.gaugeSet(ctx, g)
for name, cp, c2, kind in (("Wilson", 1.0, 0.0, 0), ("rect (Symanzik)", 5.0 / 3.0, -1.0 / 12.0, 0), ("adjoint", 1.0, -0.25, 1)):
    check(L.qexhip_wflow_general(ctx._h, 6, 0.005, cp, c2, kind)); ctx.sync()
    for rnd in range(2):
        ctx.timers_enable(1); ctx.timers_reset()
        t0 = time.perf_counter()
        check(L.qexhip_wflow_general(ctx._h, 6, 0.005, cp, c2, kind)); ctx.sync()
        dt = time.perf_counter() - t0
        n, ms = ctx.timer("staple")
        print("%-16s stage %.1f us (%d launches), wall per RK3 step %.3f ms" % (name, 1e3 * ms / max(n, 1), n, 1e3 * dt / 6), flush=True)
ctx.close()
