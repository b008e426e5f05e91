"""k_plaq timing on 32^4 (round 6: first the rows-0,1 gather experiment -- no change, kernel is latency-bound at one wavefront per SIMD --,
then __launch_bounds__(256, 2): two wavefronts per SIMD at the price of 136-192 B/lane of scratch)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qex_amd as q
lat = [32, 32, 32, 32]
ctx = q.Context(lat)
rf = q.RngField(lat, q.RngMilc6, 987654321)
q.gaugeSet(ctx, rf.random())
for rep in range(3):
    pl = q.plaq(ctx)
    ctx.timers_enable(1); ctx.timers_reset()
    for _ in range(20):
        q.plaq(ctx)
    n, ms = ctx.timer("plaq")
    ctx.timers_enable(0)
    print("k_plaq %.1f us, sum %.17g" % (1e3 * ms / n, float(np.sum(pl))), flush=True)
