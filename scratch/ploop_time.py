# time of the four Polyakov loops a flow-loop measurement takes (src/flow/gauge_flow.nim:137-156) on the resident 32^4 field
import sys, time
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
q.gaugeSet(ctx, g)
q.gaugeFlowResident(ctx, 4, 0.01)
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    pl = [q.wline(ctx, [d + 1] * lat[d]) for d in range(4)]
    ctx.sync(); dt = time.perf_counter() - t0
    print("4 Polyakov loops: %.3f ms" % (1e3 * dt), " ".join("%.15e%+.15ej" % (z.real, z.imag) for z in pl), flush=True)
t0 = time.perf_counter(); q.flowMeasure(ctx); ctx.sync(); print("flowMeasure %.3f ms" % (1e3 * (time.perf_counter() - t0)))
ctx.close()
