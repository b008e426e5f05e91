# does the fused flow stage's time depend on the field (rough -> smooth) or on how long the GPU has been busy?
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L = q.lib()
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
def rounds(tag, n):
    for rnd in range(n):
        ctx.timers_enable(1); ctx.timers_reset()
        check(L.qexhip_wflow(ctx._h, 4, 0.01)); ctx.sync()
        k, ms = ctx.timer("staple")
        print("%s round %d: stage %.1f us, plaq %.4f" % (tag, rnd, 1e3 * ms / k, q.plaq(ctx).sum()), flush=True)
q.gaugeSet(ctx, g); rounds("random start", 8)
out = np.zeros_like(g); check(L.qexhip_gauge_get(ctx._h, out.ctypes.data_as(C.c_void_p)))
q.gaugeSet(ctx, g); rounds("random start again", 3)
q.gaugeSet(ctx, out); rounds("flowed field uploaded again", 3)
one = np.zeros_like(g); one[..., 0, 0, 0] = one[..., 1, 1, 0] = one[..., 2, 2, 0] = 1.0
q.gaugeSet(ctx, one); rounds("unit links", 3)
ctx.close()
