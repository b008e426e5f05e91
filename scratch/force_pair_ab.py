# A/B of option force_pair (k_force_lds2: both parities of a tile position per workgroup) against k_force_lds:
# bit-identity of the flowed links on three lattices, then interleaved timing of the fused stage at 32^4.
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L = q.lib()
def flow(ctx, g, nsteps, pair, fe=1):
    ctx.set_option("force_pair", pair); ctx.set_option("flow_exp", fe)
    q.gaugeSet(ctx, g)
    check(L.qexhip_wflow(ctx._h, nsteps, 0.01))
    out = np.zeros_like(g)
    check(L.qexhip_gauge_get(ctx._h, out.ctypes.data_as(C.c_void_p)))
    return out
for lat in ([8, 8, 8, 8], [4, 6, 10, 6], [12, 12, 12, 12], [32, 32, 32, 32]):
    g = q.RngField(lat, q.RngMilc6, 987654321).random()
    ctx = q.Context(lat)
    for fe in (1, 0):
        a, b = flow(ctx, g, 2, 0, fe), flow(ctx, g, 2, 1, fe)
        print(lat, "flow_exp", fe, "pair vs lds after 2 RK3 steps: max abs diff %.3e" % np.abs(a - b).max(), "moved %.3e" % np.abs(a - g).max(), flush=True)
    if lat[0] == 32:
        ctx.set_option("flow_exp", 1)
        q.gaugeSet(ctx, g)
        for pair in (0, 1):
            ctx.set_option("force_pair", pair); check(L.qexhip_wflow(ctx._h, 1, 0.01))
        for rnd in range(4):
            for pair in (0, 1):
                ctx.set_option("force_pair", pair)
                ctx.timers_enable(1); ctx.timers_reset()
                check(L.qexhip_wflow(ctx._h, 4, 0.01)); ctx.sync()
                n, ms = ctx.timer("staple")
                print("round %d  force_pair=%d: stage %.1f us (%d launches)" % (rnd, pair, 1e3 * ms / n, n), flush=True)
    ctx.close()
