"""A/B of the plain (non-sharded) sweep kernels between two builds of libqexhip.so on the SAME box, alternating processes:
   python scratch/sweep_ab.py LIBPATH [--naik]    -> one line: us per sweep inside a 32^4 CG (kernel-attached events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qex_amd._lib as L
path = os.path.abspath(sys.argv[1])
L.LIB_PATH = path
import ctypes as C
probe = C.CDLL(path, mode=C.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
L.SYMBOLS = [s for s in L.SYMBOLS if hasattr(probe, s[0])]
import qex_amd as q
naik = "--naik" in sys.argv
lat = [32, 32, 32, 32]
ctx = q.Context(lat)
rf = q.RngField(lat, q.RngMilc6, 987654321)
g = rf.random()
b = rf.gaussian_vector()
q.rephase(q.Layout(lat), g)
s = q.Staggered(ctx, g, smear=q.HisqCoefs()) if naik else q.newStag(ctx, g)
bid, xid = ctx.field_new(b), ctx.field_new()
ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 30, True)
ctx.sync()
out = []
for rep in range(3):
    ctx.timers_enable(2); ctx.timers_reset()
    ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 200, True)
    ctx.sync()
    n, ms = ctx.timer("dslash")
    ctx.timers_enable(0)
    out.append(1e3 * ms / n)
print(("naik " if naik else "8-link ") + os.path.relpath(path), "us per sweep:", " ".join("%.2f" % v for v in out), flush=True)
