"""Round 6, VERDICT item 7, third rung: the Wilson-flow RK3 stage (k_force_lds2, closed-form exp) with every global matrix gather
reduced to rows 0,1 + rebuilt row 2, on a g.warm(0.5) field (SU(3) to 1e-15), against the product path on the same field."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qex_amd as q
lat = [32, 32, 32, 32]
ctx = q.Context(lat)
rf = q.RngField(lat, q.RngMilc6, 987654321)
g = rf.warm(0.5)
out = {}
for su3 in (0, 1, 0, 1):
    ctx.set_option("gauge_su3", su3)
    q.gaugeSet(ctx, g)
    q.gaugeFlowResident(ctx, 8, 0.01)          # warm-up: clocks, second link buffer
    ctx.timers_enable(1); ctx.timers_reset()
    q.gaugeFlowResident(ctx, 10, 0.01)
    ctx.sync()
    n, ms = ctx.timer("staple")
    ctx.timers_enable(0)
    pl = q.plaq(ctx)
    out.setdefault(su3, []).append(pl)
    print("gauge_su3 = %d: RK3 stage %.1f us (%d launches), plaq sum after 18 steps %.15f" % (su3, 1e3 * ms / n, n, float(np.sum(pl))), flush=True)
print("max |plaq difference| between the two paths after 18 flow steps:", float(np.abs(out[1][0] - out[0][0]).max()))
