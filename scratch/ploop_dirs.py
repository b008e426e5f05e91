import sys, time
sys.path.insert(0, '.')
import qex_amd as q
lat = [32] * 4
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat); q.gaugeSet(ctx, g); q.gaugeFlowResident(ctx, 4, 0.01)
for rep in range(2):
    for d in range(4):
        ctx.sync(); t0 = time.perf_counter(); w = q.wline(ctx, [d + 1] * 32); ctx.sync()
        print("direction %d: %.1f us" % (d, 1e6 * (time.perf_counter() - t0)))
    ctx.sync(); t0 = time.perf_counter(); q.ploops(ctx); ctx.sync(); print("all four: %.1f us" % (1e6 * (time.perf_counter() - t0)))
    ctx.sync(); t0 = time.perf_counter(); q.plaq(ctx); ctx.sync(); print("(plaq call for scale: %.1f us)" % (1e6 * (time.perf_counter() - t0)))
