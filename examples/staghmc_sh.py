#!/usr/bin/env python
"""src/examples/staghmc_sh.nim (nHYP-smeared staggered HMC with Hasenbusch masses) with every field operation in
libqexhip: smearing + force chain, solves (lock-step batches), fermion and gauge forces, link updates, action,
plaquettes, Polyakov loops, random fields.  The MD schedule is the one mdevolve prints for the reference run.

    python examples/staghmc_sh.py [-run 0|1|2] [-trajs 2] [-halo]

-run selects the parameter set of tests/extra/staghmc_sh/run (test 0, 1 or 2); the log lines have the format of the
reference's (`Begin H:`, `End H:`, `MEASpbp`, `MEASplaq`, `MEASploop`), so the reference's own `diffnum` comparison
against tests/extra/staghmc_sh/ref.N applies.  -halo runs every kernel in its t-sharded form on one GPU.
The driver logic (integrator schedule, trajectory bookkeeping) is tests/hmc_replay.py."""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qex_amd as q  # noqa: E402
import hmc_replay as R  # noqa: E402


class Host:
    """the two host-side helpers the driver asks of its first argument, from the product instead of the oracle"""
    Layout = staticmethod(q.Layout)

    @staticmethod
    def gauge_unit(lo):
        return q.unit(lo)


class GlobalRng:
    """`var R: RngMilc6; R.seed(seed, 987654321)` (staghmc_sh.nim:178-179): the accept/reject stream"""

    def __init__(self, seed, index=987654321):
        m = 0xFFFFFFFF
        s, self.r = seed & m, []
        for _ in range(7):
            s = ((69607 + 8 * index) * s + 12345) & m
            self.r.append((s >> 8) & 0xFFFFFF)
        self.ic = ((69607 + 8 * index) * s + 12345) & m
        self.mult = (100005 + 8 * index) & m

    def uniform(self):
        r = self.r
        t = (((r[5] >> 7) | (r[6] << 17)) ^ ((r[4] >> 1) | (r[5] << 23))) & 0xFFFFFF
        self.r = [t] + r[:6]
        self.ic = (self.ic * self.mult + 12345) & 0xFFFFFFFF
        return float(np.float32(t ^ ((self.ic >> 8) & 0xFFFFFF)) * np.float32(1.0 / 16777216.0))


def fmt(e):
    return "H: %r  Sg: %r  Sf: @[%s]  T: %r" % (float(e["H"]), float(e["Sg"]), ", ".join(repr(float(v)) for v in e["Sf"]), float(e["T"]))


ap = argparse.ArgumentParser()
ap.add_argument("-run", type=int, default=0, choices=[0, 1, 2])
ap.add_argument("-trajs", type=int, default=2)
ap.add_argument("-halo", action="store_true")
ap.add_argument("-lat", type=int, nargs=4, default=None, help="another lattice than the reference run's 8^4 (no golden log then)")
ap.add_argument("-resident", action="store_true", help="MD evolution with links and momenta resident on the device (qexhip_md_*)")
ap.add_argument("-device", action="store_true",
                help="both ends of the trajectory on the device too (implies -resident): momenta, pseudofermions and pbp sources "
                     "drawn into HBM by the RngMilc6 field, energies and measurements from resident fields; ACCEPT branch, no "
                     "reversibility check (tests/hmc_replay.py: DeviceEndsReplay)")
ap.add_argument("-time", action="store_true", help="print wall time and the kernel-time breakdown of every trajectory")
a = ap.parse_args()

if a.lat:
    R.LAT = list(a.lat)
cfg = R.CONFIGS[a.run]
be = R.HipBackend(q, R.LAT, halo=a.halo, resident=a.resident or a.device)
print(be.ctx.info())
if a.device:
    r = R.DeviceEndsReplay(Host, be, cfg, q.RngField(R.LAT, q.RngMilc6, R.SEED))
else:
    r = R.Replay(Host, be, cfg, rng=q.RngField(R.LAT, q.RngMilc6, R.SEED))
G = GlobalRng(R.SEED)
pl = be.plaq(r.g)
print("MEASplaq ss: %r  st: %r  tot: %r" % (float(2 * sum(pl[:3])), float(2 * sum(pl[3:])), float(0.5 * (2 * sum(pl[:3]) + 2 * sum(pl[3:])))))
import time  # noqa: E402

TIMERS = ("expupdate", "dslash", "dslash_bnd", "dslash_batch", "blas", "reduce", "smear", "nhyp_force", "smear_deriv", "staple", "outer", "plaq")
for n in range(1, a.trajs + 1):
    if a.time:
        be.ctx.timers_enable(1)
        be.ctx.timers_reset()
        t_traj = time.time()
    g0 = None if a.device else r.g.copy()
    b = r.refresh()
    print("Begin " + fmt(b))
    r.evolve()
    e = r.finish_energies()
    print("End " + fmt(e))
    if n % 2 == 0 and not a.device:                            # revCheckFreq = 2
        print("Reversed " + fmt(r.reverse_check()))
    dH = e["H"] - b["H"]
    acc, u = math.exp(-dH), G.uniform()
    ok = u <= acc
    print("%s:  dH: %r  exp(-dH): %r  r: %r" % ("ACCEPT" if ok else "REJECT", float(dH), acc, u))
    if a.time:
        parts = {k: be.ctx.timer(k) for k in TIMERS}
        be.ctx.timers_enable(0)
        be.ctx.sync()
        print("TIME trajectory %d: wall %.2f s; kernel ms: %s" % (n, time.time() - t_traj, "  ".join(
            "%s %.1f (%d)" % (k, ms, cnt) for k, (cnt, ms) in parts.items() if cnt)))
    if a.device and not ok:
        print("REJECT with -device: the resident replay covers the ACCEPT branch only; stopping")
        break
    m = r.measure(accepted=ok, g0=g0)
    for v, its in zip(m["pbp"], m["pbp_iters"]):
        print("stagSolve: %d" % its)
        print("MEASpbp mass %r : %r" % (R.PBPMASS, float(v)))
    print("MEASplaq ss: %r  st: %r  tot: %r" % tuple(float(v) for v in m["plaq"]))
    print("MEASploop spatial: %r %r temporal: %r %r" % tuple(float(v) for v in m["ploop"]))
