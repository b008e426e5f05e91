"""Round 6, VERDICT item 7 (bounded experiment): does gathering rows 0,1 of SU(3) links and rebuilding row 2 in registers help the
gather-bound gauge kernels?  First rung: k_plaq (16 gathered matrices per site, 230 us at 32^4, 1.70x over-fetch) on a g.warm field."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qex_amd as q

lat = [32, 32, 32, 32]
ctx = q.Context(lat)
rf = q.RngField(lat, q.RngMilc6, 987654321)
g = rf.warm(0.5)
q.gaugeSet(ctx, g)
res = {}
for su3 in (0, 1, 0, 1):
    ctx.set_option("gauge_su3", su3)
    pl = q.plaq(ctx)
    ctx.timers_enable(1)
    ctx.timers_reset()
    for _ in range(20):
        q.plaq(ctx)
    n, ms = ctx.timer("plaq")
    ctx.timers_enable(0)
    res.setdefault(su3, []).append((1e3 * ms / n, pl))
    print("gauge_su3 =", su3, ": k_plaq %.1f us" % (1e3 * ms / n), "sum", float(np.sum(pl)), flush=True)
d = np.abs(res[1][0][1] - res[0][0][1]).max()
print("max |plaq(rows 0,1 + rebuilt row 2) - plaq(18 reals)| =", d)
