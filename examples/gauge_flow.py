#!/usr/bin/env python
"""The flow loop of the fork's src/flow/gauge_flow.nim through libqexhip: reunitarise, measure, then for every (dt, t_max) pair
flow with `gc.gaugeFlow(flow_act, g, dt)` and print the reference's FLOW line after every step (gauge_flow.nim:380-470):

    FLOW t  plaq  E  t^2E  d(t^2E)/dt  check  Q  t^2E_ss  t^2E_st  Re/Im P_t  Re/Im P_s

Per step on the device: one RK3 step (three fused stages), `qexhip_flow_measure` (plaquettes + clover E_s, E_t, Q in one pass:
`EQ` and `meas_plaq`, :139-156,360-379) and `qexhip_polyakov_loops` (`meas_ploop`, :137-156); only the 17 numbers of the
measurement cross PCIe.  The configuration comes from the library's RngMilc6 field (`-warm s`, default the hot start) or from a
SciDAC file (`-load file`), where the reference reads `<fn>_<config>.lat`.

    python examples/gauge_flow.py [-lat 16 16 16 16] [-act Wilson|rect|adj] [-dt 0.02 0.1] [-tmax 1.0 2.0] [-time]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import qex_amd as q  # noqa: E402
from qex_amd._lib import check, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-lat", type=int, nargs=4, default=[8, 8, 8, 8])
ap.add_argument("-act", default="Wilson", choices=["Wilson", "rect", "adj"])
ap.add_argument("-c1", type=float, default=-1.0 / 12.0, help="rect: plaq = 1 - 8 c1, rect = c1 (input_gf.xml)")
ap.add_argument("-adjplaq", type=float, default=-0.25, help="adj: beta_adj / beta_F")
ap.add_argument("-dt", type=float, nargs="+", default=[0.02, 0.1])
ap.add_argument("-tmax", type=float, nargs="+", default=[0.2, 0.6], help="flow each dt up to this t/a^2 (time1, time2 of input_gf.xml)")
ap.add_argument("-warm", type=float, default=None)
ap.add_argument("-load", default=None)
ap.add_argument("-seed", type=int, default=987654321)
ap.add_argument("-time", action="store_true")
a = ap.parse_args()
assert len(a.dt) == len(a.tmax)

if a.load:
    g = q.loadGauge(a.load, a.lat)
else:
    rf = q.RngField(a.lat, q.RngMilc6, a.seed)
    g = rf.warm(a.warm) if a.warm is not None else rf.random()
ctx = q.Context(a.lat)
print(ctx.info())
q.reunit(ctx, g)                                                     # read_gauge_file -> g.reunit (:344-353)
plaq, rect, adj = {"Wilson": (1.0, 0.0, 0.0), "rect": (1.0 - 8.0 * a.c1, a.c1, 0.0), "adj": (1.0, 0.0, a.adjplaq)}[a.act]
kind = 1 if a.act == "adj" else 0
q.gaugeSet(ctx, g)


def EQ():
    """EQ (:360-379): E_s, E_t, spatial / temporal plaquettes, Q, spatial / temporal Polyakov loops of the resident field"""
    pl, (es, et, qq) = q.flowMeasure(ctx)
    loops = q.ploops(ctx)
    return es, et, 2.0 * pl[:3].sum(), 2.0 * pl[3:].sum(), qq, sum(loops[:3]) / 3.0, loops[3]


def print_info(dtau, tau, m, old_t2E):
    """print_info (:385-470)"""
    es, et, ss, st, qq, pls, plt = m
    pl = (3.0 * ss + 3.0 * st) / 2.0
    clov = es + et
    t2E = tau * tau * clov
    vals = [pl, clov, t2E, (t2E - old_t2E) / dtau, 12.0 * tau * tau * (3.0 - pl), qq, tau * tau * es, tau * tau * et,
            3.0 * plt.real, 3.0 * plt.imag, 3.0 * pls.real, 3.0 * pls.imag]
    print("FLOW %.2f " % tau + " ".join("%.13f" % v for v in vals))
    return t2E


t_all = time.perf_counter()
t2E = print_info(a.dt[0], 0.0, EQ(), 0.0)
tau, nsteps = 0.0, 0
for dt, tmax in zip(a.dt, a.tmax):
    while tau < tmax - 1e-12:
        check(lib().qexhip_wflow_general(ctx._h, 1, dt, plaq, adj if kind else rect, kind))
        tau += dt
        nsteps += 1
        t2E = print_info(dt, tau, EQ(), t2E)
ctx.sync()
if a.time:
    dt_all = time.perf_counter() - t_all
    print("%d flow steps with their measurements: %.3f s, %.3f ms per step" % (nsteps, dt_all, 1e3 * dt_all / max(nsteps, 1)))
