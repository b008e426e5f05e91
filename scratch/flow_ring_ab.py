# A/B of the Wilson-flow stage: loader/consumer kernel (flow_stage.hip, option flow_ring = 1) against k_force_lds (0).
# Correctness at 8^4 and 32^4 (same start, one RK3 step each way, links compared), then interleaved timing at 32^4.
import ctypes as C, os, sys, time
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L = q.lib()
def flow(ctx, g, nsteps, ring):
    ctx.set_option("flow_ring", ring)
    q.gaugeSet(ctx, g)
    check(L.qexhip_wflow(ctx._h, nsteps, 0.01))
    out = np.zeros_like(g)
    check(L.qexhip_gauge_get(ctx._h, out.ctypes.data_as(C.c_void_p)))
    return out
dbg = int(os.environ.get("QEXHIP_FLOW_STAGE_DBG", "0"))
for lat in ([8, 8, 8, 8], [4, 6, 10, 6], [32, 32, 32, 32]):
    g = q.RngField(lat, q.RngMilc6, 987654321).random()
    ctx = q.Context(lat)
    if not dbg:
        a, b = flow(ctx, g, 2, 0), flow(ctx, g, 2, 1)
        print(lat, "ring vs lds after 2 RK3 steps: max abs diff %.3e" % np.abs(a - b).max(), "plaq", q.plaq(ctx).sum(), flush=True)
        for fe in (0,):
            ctx.set_option("flow_exp", fe)
            a, b = flow(ctx, g, 1, 0), flow(ctx, g, 1, 1)
            print(lat, "reference exp: max abs diff %.3e" % np.abs(a - b).max(), flush=True)
            ctx.set_option("flow_exp", 1)
    if lat[0] == 32:
        q.gaugeSet(ctx, g)
        for ring in (0, 1):
            ctx.set_option("flow_ring", ring); check(L.qexhip_wflow(ctx._h, 1, 0.01))
        for rnd in range(3):
            for ring in (0, 1):
                ctx.set_option("flow_ring", ring)
                ctx.timers_enable(1); ctx.timers_reset()
                check(L.qexhip_wflow(ctx._h, 4, 0.01)); ctx.sync()
                n, ms = ctx.timer("staple")
                print("round %d  flow_ring=%d dbg=%d: stage %.1f us (%d launches)" % (rnd, ring, dbg, 1e3 * ms / n, n), flush=True)
    ctx.close()
