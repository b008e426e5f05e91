# one rank's 48^3 x 12 share: nHYP smearing and the smeared-force chain on RESIDENT fields (no PCIe), wall clock around
# device syncs, sharded form (ghost zones, one-rank RCCL communicator) against the periodic kernels
import sys, time; sys.path.insert(0, '.')
import numpy as np, qex_amd as q
lat = [int(v) for v in (sys.argv[1].split('x') if len(sys.argv) > 1 else [48, 48, 48, 12])]
halo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = q.RngField(lat, q.RngMilc6, 987654321).warm(0.5)
ctx = q.Context(lat)
if halo:
    ctx.comm_init(q.Context.unique_id(), 1, 0); ctx.force_halo(True)
md = q.ResidentMD(ctx); md.begin(g, None)
hc = q.HypCoefs(0.4, 0.5, 0.5)
sf = hc.smearGetForce(ctx, None); sf.gforce(None, plaq=1.0); ctx.sync()
for rnd in range(3):
    t0 = time.time()
    for _ in range(5): sf = hc.smearGetForce(ctx, None)
    ctx.sync(); t1 = time.time()
    for _ in range(5): sf.gforce(None, plaq=1.0)
    ctx.sync(); t2 = time.time()
    print("lat %s halo %d round %d: nHYP smear %.2f ms, gauge force through the chain %.2f ms (wall, resident)" % (lat, halo, rnd, (t1 - t0) * 200, (t2 - t1) * 200), flush=True)
sf.release(); md.end()
