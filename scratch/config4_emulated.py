"""BASELINE configs[4]'s gauge-sector pieces on one rank's slab of the 8-GPU job (48^3 x 12), wall time per call:
nHYP smear (smearGetForce), the smeared gauge force chain, the fermion force (2 pseudofermions), the HISQ link build.
  periodic          the slab as a periodic lattice: the kernels alone, nothing exchanged
  halo              ghost zones through the one-rank exchange path (launch / event structure of the sharded job, no transport time)
  halo + emulation  every exchange preceded (RCCL arm) / stretched (peer arm) by 3 us + bytes / 45 GB/s
usage: QEXHIP_TRANSPORT=rccl|mbox|peer python scratch/config4_emulated.py [XxYxZxT] [mode]
Round 6: give ONE mode per process (periodic | halo | halo+emu).  With the three contexts in one process the third one paid ~11 ms per
call that had nothing to do with the emulation (gforce alone in its own process: 32.7 ms halo, 36.8 ms halo + emulation; as the third
context of one process 48.5) -- round 5's "+19 ms exposed" was mostly that."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, qex_amd as q
lat = [int(v) for v in (sys.argv[1].split('x') if len(sys.argv) > 1 else [48, 48, 48, 12])]
only = sys.argv[2] if len(sys.argv) > 2 else None
lo = q.Layout(lat)
g = q.RngField(lat, q.RngMilc6, 987654321).warm(0.5)
psis = [q.synthetic_gaussian_vector(lo, seed=5 + k) for k in range(2)]
hc, hq = q.HypCoefs(0.4, 0.5, 0.5), q.HisqCoefs()


def timed(fn, ctx, n=3):
    fn(); ctx.sync()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ctx.sync()
        ts.append(1e3 * (time.perf_counter() - t))
    if max(ts) > 3 * min(ts):
        print("   (uneven calls: %s ms)" % ", ".join("%.1f" % v for v in ts), flush=True)
    return sorted(ts)[len(ts) // 2]


rows = []
for mode in ("periodic", "halo", "halo+emu"):
    if only and mode != only:
        continue
    ctx = q.Context(lat)
    if mode != "periodic":
        ctx.comm_init(q.Context.unique_id(), 1, 0)
        ctx.force_halo(True)
        ctx.set_option("multi_reduce", 1)
    if mode == "halo+emu":
        ctx.set_option("emu_exchange_us", 3); ctx.set_option("emu_link_gbs", 45); ctx.set_option("emu_allreduce_us", 3 if ctx.comm_transport()[0] == "peer" else 15)
    fl, f = np.zeros_like(g), np.zeros_like(g)
    box = {}
    def smear(): box["sf"] = hc.smearGetForce(ctx, g, fl)
    t_smear = timed(smear, ctx)
    sf = box["sf"]
    t_gf = timed(lambda: sf.gforce(f, plaq=1.0), ctx)
    t_ff = timed(lambda: sf.fforce(f, psis, [1.0, 0.5]), ctx)
    sf.release()
    t_hisq = timed(lambda: q.Staggered(ctx, g, smear=hq), ctx, n=2)
    tr = ctx.comm_transport()
    rows.append((mode, t_smear, t_gf, t_ff, t_hisq))
    print("%-10s transport %-4s  nHYP smear %7.2f ms   gauge force chain %7.2f ms   fermion force %7.2f ms   HISQ build %7.2f ms   %s"
          % (mode, tr[0], t_smear, t_gf, t_ff, t_hisq, tr[1] if tr[0] == "peer" else ""), flush=True)
    ctx.close()
p = rows[0]
for r in (rows[1:] if not only else []):
    print("%-10s / periodic:  smear %.2fx  gauge force %.2fx  fermion force %.2fx  HISQ %.2fx" % (r[0], r[1] / p[1], r[2] / p[2], r[3] / p[3], r[4] / p[4]))
