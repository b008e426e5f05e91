"""Workload for the rocprofv3 passes of profiles/collect.sh (kernel trace and PMC): every kernel the bench line prices,
on the bench's own inputs, plus 1 GiB streaming kernels of known byte count that calibrate FETCH_SIZE / WRITE_SIZE
(MI355X_MICROARCH.md, HBM section: gfx950 reports half of a wide coalesced read).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 profiles/pmc_workload.py [what ...]

what: any of  cg  cgw  naik  flow  nhyp  (default: all)."""
import ctypes as C
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qex_amd as q  # noqa: E402

what = set(sys.argv[1:]) or {"cg", "cgw", "naik", "flow", "nhyp"}
from qex_amd._lib import tune_lib
L = tune_lib()   # libqexhip_tune.so: measurement scaffolding, not the product library
L.qexhip_tune_stream.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
lat = [32, 32, 32, 32]
lo = q.Layout(lat)
rf = q.RngField(lat, q.RngMilc6, 987654321)
g0 = rf.random()
b = rf.gaussian_vector()
g = g0.copy()
q.rephase(lo, g)
ctx = q.Context(lat)
out = C.c_double(0)
for mode in (0, 1):
    L.qexhip_tune_stream(ctx._h, mode, 1024, 2048, 3, C.byref(out))   # k_copy16 / k_read16: 1 GiB, known byte counts
bid, xid = ctx.field_new(b), ctx.field_new()
if "cg" in what:                       # the headline: 18-real links of g.random
    s = q.newStag(ctx, g)
    ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 20, True)
if "cgw" in what:                      # compressed links of g.warm(0.5)
    gw = rf.warm(0.5)
    q.rephase(lo, gw)
    s = q.newStag(ctx, gw)
    ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 20, True)
if "naik" in what:                     # HISQ links, 10-shift multi-shift CG
    s = q.Staggered(ctx, g, smear=q.HisqCoefs())
    masses = [math.sqrt(k + 2.0) for k in range(10)]
    shifts = [masses[0]] + [4.0 * (m * m - masses[0] ** 2) for m in masses[1:]]
    xids = [ctx.field_new() for _ in masses]
    ctx.dev_solve_xx_multi(xids, bid, shifts, 0.0, 20, True)
if "flow" in what:                     # Wilson flow: 2 RK3 steps + plaquette + clover observables
    q.gaugeSet(ctx, g0)
    q.gaugeFlowResident(ctx, 2, 0.01)
    q.plaq(ctx)
    q.flowEQ(ctx, 1)                   # the clover E, Q a flow loop measures after every step
    q.flowMeasure(ctx)                 # round 3: plaquettes + E, Q from ONE pass (the clover kernel, second dispatch)
if "nhyp" in what:                     # nHYP smearing closure + the force chain (twice)
    hc = q.HypCoefs(0.4, 0.5, 0.5)
    sf = hc.smearGetForce(ctx, g0)
    f = np.zeros_like(g0)
    for _ in range(2):
        sf.gforce(f, plaq=1.0)
    sf.release()
ctx.sync()
print("pmc_workload done:", sorted(what))
