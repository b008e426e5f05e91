"""Round 6: the HISQ build of a t-sharded slab in a FRESH process: wall time of six consecutive calls (config4_modes.sh saw 1.1 s)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, qex_amd as q
lat = [48, 48, 48, 12]
g = q.RngField(lat, q.RngMilc6, 987654321).warm(0.5)
ctx = q.Context(lat)
if os.environ.get("HALO", "1") == "1":
    ctx.comm_init(q.Context.unique_id(), 1, 0)
    ctx.force_halo(True)
    ctx.set_option("multi_reduce", 1)
    if os.environ.get("QEX_EMU") == "1":
        ctx.set_option("emu_exchange_us", 3); ctx.set_option("emu_link_gbs", 45); ctx.set_option("emu_allreduce_us", 3)
hq = q.HisqCoefs()
for k in range(6):
    t = time.perf_counter()
    s = q.Staggered(ctx, g, smear=hq)
    ctx.sync()
    t1 = time.perf_counter()
    del s
    print("call %d: build %.1f ms, release %.1f ms" % (k, 1e3 * (t1 - t), 1e3 * (time.perf_counter() - t1)), flush=True)
