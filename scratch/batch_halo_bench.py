# one rank's 48^3 x 12 share, lock-step batched CG (the HMC's Hasenbusch solves) in its sharded form on one GPU:
# ghost zones, one-rank RCCL communicator, multi-rank reduction branch; us per iteration of all systems (wall, resident... host fields in/out excluded by differencing)
import sys, time; sys.path.insert(0, '.')
import numpy as np, qex_amd as q
lat = [int(v) for v in (sys.argv[1].split('x') if len(sys.argv) > 1 else [48, 48, 48, 12])]
halo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = q.Layout(lat)
rf = q.RngField(lat, q.RngMilc6, 987654321)
g = rf.warm(0.5); q.rephase(lo, g)
ctx = q.Context(lat)
if halo:
    ctx.comm_init(q.Context.unique_id(), 1, 0); ctx.force_halo(True); ctx.set_option("batch_multi", 1)
    if len(sys.argv) > 3: ctx.set_option("overlap", int(sys.argv[3]))
    import os
    if os.environ.get("QEX_EMU"):          # "exchange_us,link_gbs,allreduce_us": emulated transport (round 6)
        e = [int(v) for v in os.environ["QEX_EMU"].split(",")]
        ctx.set_option("emu_exchange_us", e[0]); ctx.set_option("emu_link_gbs", e[1]); ctx.set_option("emu_allreduce_us", e[2])
    print("transport", ctx.comm_transport()[0], flush=True)
s = q.newStag(ctx, g)
ms = [0.1, 0.2, 0.4, 0.05]
bs = [rf.gaussian_vector() for _ in range(4)]
for b in bs: b[lo.vol // 2:] = 0
for n in (1, 3, 4):
    xs = [np.zeros_like(b) for b in bs[:n]]
    s.solveXX_batch(xs, bs[:n], ms[:n], 0.0, 10, True)
    t = {}
    for K in (100, 400):
        t0 = time.time(); s.solveXX_batch(xs, bs[:n], ms[:n], 0.0, K, True); t[K] = time.time() - t0
    print("lat %s halo %d overlap %s n=%d: %.1f us per lock-step iteration (%.1f per system)" % (lat, halo, sys.argv[3] if len(sys.argv) > 3 else "-", n, 1e6 * (t[400] - t[100]) / 300, 1e6 * (t[400] - t[100]) / 300 / n), flush=True)
