"""BASELINE configs[4]'s solver on one rank's 48^3 x 12 slab of the 8-GPU job: HISQ fat + Naik links, 10-shift multi-shift CG (m_k = sqrt(k+2),
stagSolve.nim:598), one-rank rehearsal under emulated transport (3 us + bytes / 45 GB/s per exchange, +3 us per rank sum) on the transport
QEXHIP_TRANSPORT names; us per iteration by differencing two iteration counts (device-resident fields)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, qex_amd as q
lat = [int(v) for v in (sys.argv[1].split("x") if len(sys.argv) > 1 else [48, 48, 48, 12])]
halo = len(sys.argv) < 3 or sys.argv[2] != "periodic"
rf = q.RngField(lat, q.RngMilc6, 987654321)
g = rf.random(); b = rf.gaussian_vector()
q.rephase(q.Layout(lat), g)
ctx = q.Context(lat)
if halo:
    ctx.comm_init(q.Context.unique_id(), 1, 0)
    ctx.force_halo(True); ctx.set_option("multi_reduce", 1)
    ctx.set_option("emu_exchange_us", 3); ctx.set_option("emu_link_gbs", 45); ctx.set_option("emu_allreduce_us", 3 if ctx.comm_transport()[0] != "rccl" else 15)
    ctx.set_option("overlap", -2)
s = q.Staggered(ctx, g, smear=q.HisqCoefs())
masses = [float(np.sqrt(k + 2.0)) for k in range(10)]
shifts = [masses[0]] + [4.0 * (m * m - masses[0] ** 2) for m in masses[1:]]
bid = ctx.field_new(b); xids = [ctx.field_new() for _ in masses]
ctx.dev_solve_xx_multi(xids, bid, shifts, 0.0, 5, True); ctx.sync()
t = {}
for K in (40, 160):
    t0 = time.perf_counter(); its, _ = ctx.dev_solve_xx_multi(xids, bid, shifts, 0.0, K, True); ctx.sync(); t[K] = time.perf_counter() - t0
print("lattice %s %s transport %s: Naik 10-shift multi-shift CG %.1f us per iteration; sweep %s" % (
    lat, "sharded (emulated transport)" if halo else "periodic", ctx.comm_transport()[0], 1e6 * (t[160] - t[40]) / 120,
    {k: v for k, v in ctx.sweep_info().items() if k in ("overlap", "form", "exchange_us", "boundary_at", "tuned_us_per_sweep")} if halo else "-"), flush=True)
