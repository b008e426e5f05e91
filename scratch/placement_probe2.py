# the 254 vs 265 us modes alternate with context creation: tied to the context (its HIP streams / hardware queue) or to time?
import sys, time
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
b = np.random.default_rng(5).standard_normal((int(np.prod(lat)), 3, 2))
ctxs = []
def make():
    ctx = q.Context(lat); s = q.Staggered(ctx, g)
    bid = ctx.field_new(b); xid = ctx.field_new(None)
    ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 50); ctx.sync()
    return (ctx, s, bid, xid)
def run(tag, t):
    ctx, s, bid, xid = t
    t0 = time.perf_counter(); ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 400); ctx.sync(); dt = time.perf_counter() - t0
    print("%s: %.2f us/iteration" % (tag, 1e6 * dt / 400), flush=True)
A = make(); run("A (first context)", A)
B = make(); run("B (second context, A alive)", B)
run("A again", A); run("B again", B)
C = make(); run("C (third, A and B alive)", C)
run("A again", A); run("B again", B); run("C again", C)
