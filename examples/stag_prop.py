#!/usr/bin/env python
"""tests/examples/testStagProp.nim through libqexhip: HISQ-smeared staggered propagator from a point source.

    python examples/stag_prop.py [-lat 8 8 8 8] [-mass 0.001] [-seed 987654321]

Everything runs in the library: the configuration comes from its RngMilc6 field (qexhip_rng_*), boundary condition
and staggered phases on the host as in QEX, HISQ fat + long links are built on the GPU and handed to the operator
without leaving it, D and the solve are the HIP kernels."""
import argparse
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qex_amd as q  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-lat", type=int, nargs=4, default=[8, 8, 8, 8])
ap.add_argument("-mass", type=float, default=0.001)
ap.add_argument("-seed", type=int, default=987654321)
ap.add_argument("-warm", type=float, default=0.5)
a = ap.parse_args()

lo = q.Layout(a.lat)
g = q.RngField(a.lat, q.RngMilc6, a.seed).warm(a.warm)          # g.warm 0.5, r
q.rephase(lo, g)                                                  # g.setBC; g.stagPhase
ctx = q.Context(a.lat)
print(ctx.info())
s = q.Staggered(ctx, g, smear=q.HisqCoefs().init())               # hc.smear(g, fl, ll); newStag3(fl, ll)
print("links per site, storage format, max deviation:", s.links_info())
v1 = np.zeros((lo.vol, 3, 2))
v1[0, 0, 0] = 1.0                                                 # point source, colour 0 at the origin
v2 = np.zeros_like(v1)
s.D(v2, v1, a.mass)
print("|D v1|^2 =", (v2 * v2).sum())
sp = q.SolverParams(r2req=1e-16, maxits=100000, verbosity=0)
t = time.time()
s.solve(v2, v1, a.mass, sp)
print("solve: %d iterations, %.3f s, r2/b2 = %.3e" % (sp.iterations, time.time() - t, sp.r2))
r = np.zeros_like(v1)
s.D(r, v2, a.mass)
print("true residual^2 =", ((r - v1) ** 2).sum())
print("|v2|^2 even, odd =", (v2[:lo.vol // 2] ** 2).sum(), (v2[lo.vol // 2:] ** 2).sum())
