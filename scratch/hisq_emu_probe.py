import sys, time; sys.path.insert(0, '.')
import numpy as np, qex_amd as q
lat = [48, 48, 48, 12]
lo = q.Layout(lat)
g = q.RngField(lat, q.RngMilc6, 987654321).warm(0.5)
ctx = q.Context(lat)
ctx.comm_init(q.Context.unique_id(), 1, 0); ctx.force_halo(True); ctx.set_option("multi_reduce", 1)
if len(sys.argv) > 1 and sys.argv[1] == "emu":
    ctx.set_option("emu_exchange_us", 3); ctx.set_option("emu_link_gbs", 45)
hc, hq = q.HypCoefs(0.4, 0.5, 0.5), q.HisqCoefs()
fl, f = np.zeros_like(g), np.zeros_like(g)
def T(what, fn):
    t = time.perf_counter(); r = fn(); ctx.sync(); print("%-14s %8.1f ms" % (what, 1e3 * (time.perf_counter() - t)), flush=True); return r
sf = T("smear", lambda: hc.smearGetForce(ctx, g, fl))
T("gforce", lambda: sf.gforce(f, plaq=1.0))
T("gforce", lambda: sf.gforce(f, plaq=1.0))
psis = [q.synthetic_gaussian_vector(lo, seed=5 + k) for k in range(2)]
T("fforce", lambda: sf.fforce(f, psis, [1.0, 0.5]))
T("fforce", lambda: sf.fforce(f, psis, [1.0, 0.5]))
T("release", lambda: sf.release())
for i in range(4):
    T("hisq %d" % i, lambda: q.Staggered(ctx, g, smear=hq))
print(ctx.comm_transport())
