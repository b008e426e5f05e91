#!/usr/bin/env python
"""tests/base/twflow_topo.nim through libqexhip: Wilson flow of a warm MRG32k3a configuration with the clover
observables E_s, E_t, Q after every step (src/flow/gauge_flow.nim:139-156).

    python examples/wflow_topo.py [-lat 8 8 8 16] [-steps 20] [-eps 0.005] [-loop 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qex_amd as q  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-lat", type=int, nargs=4, default=[8, 8, 8, 16])
ap.add_argument("-steps", type=int, default=20)
ap.add_argument("-eps", type=float, default=0.005)
ap.add_argument("-loop", type=int, default=5, choices=[1, 3, 4, 5])
ap.add_argument("-seed", type=int, default=17 ** 13)
a = ap.parse_args()

g = q.RngField(a.lat, q.MRG32k3a, a.seed).warm(0.4)               # g.warm(0.4, r) (twflow_topo.nim:19-24)
ctx = q.Context(a.lat)
print(ctx.info())
print("t = 0      plaq %.12f  E_s, E_t, Q = %s" % (q.plaq(ctx, g).sum(), q.flowEQ(ctx, a.loop)))


def measure(t):
    print("t = %-6g plaq %.12f  E_s, E_t, Q = %s" % (t, q.plaq(ctx).sum(), q.flowEQ(ctx, a.loop)))


q.gaugeFlow(ctx, g, a.steps, a.eps, measure=measure)
