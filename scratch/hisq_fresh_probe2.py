"""Round 6: the sequence of scratch/config4_emulated.py (nHYP smear, gauge force, fermion force, release, HISQ builds) on a sharded
slab in a fresh process, every call timed: where does the 1.1 s HISQ build come from?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, qex_amd as q
lat = [48, 48, 48, 12]
lo = q.Layout(lat)
g = q.RngField(lat, q.RngMilc6, 987654321).warm(0.5)
psis = [q.synthetic_gaussian_vector(lo, seed=5 + k) for k in range(2)]
ctx = q.Context(lat)
ctx.comm_init(q.Context.unique_id(), 1, 0)
ctx.force_halo(True)
ctx.set_option("multi_reduce", 1)
def T(name, fn):
    t = time.perf_counter(); r = fn(); ctx.sync(); print("%-14s %.1f ms" % (name, 1e3 * (time.perf_counter() - t)), flush=True); return r
fl, f = np.zeros_like(g), np.zeros_like(g)
hc, hq = q.HypCoefs(0.4, 0.5, 0.5), q.HisqCoefs()
steps = os.environ.get("STEPS", "smear,gforce,fforce,release").split(",")
sf = None
if "smear" in steps:
    for _ in range(2): sf = T("smear", lambda: hc.smearGetForce(ctx, g, fl))
if "gforce" in steps:
    for _ in range(2): T("gforce", lambda: sf.gforce(f, plaq=1.0))
if "fforce" in steps:
    for _ in range(2): T("fforce", lambda: sf.fforce(f, psis, [1.0, 0.5]))
if "release" in steps and sf is not None:
    T("release", lambda: sf.release())
    if os.environ.get("SLEEP_AFTER_RELEASE"):
        time.sleep(float(os.environ["SLEEP_AFTER_RELEASE"]))
        T("sync after sleep", lambda: None)
for k in range(4):
    s = T("hisq build", lambda: q.Staggered(ctx, g, smear=hq))
    del s
