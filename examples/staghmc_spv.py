#!/usr/bin/env python
"""src/stagg_pv_hmc/staghmc_spv.nim -- the fork's HMC with nHYP-smeared staggered fermions, Pauli-Villars bosons and an
optional gauge action on the smeared links -- with every field operation in libqexhip.

    python examples/staghmc_spv.py [-lat 8 8 8 16] [-trajs 2] [-host-fields] [-halo] [-time]
    python -m torch.distributed.run --nproc-per-node N examples/staghmc_spv.py ...     # t-sharded over N ranks (one GPU each, or sharing one)

The driver below follows the reference's procs one to one (file:line in the docstrings); the parameters are those of
src/stagg_pv_hmc/input_hmc.xml.  By default the MD loop keeps links, momenta and forces on the device (qexhip_md_*);
-host-fields runs the same trajectory through the host-pointer entry points (what a drop-in for QEX's host-resident
fields does call by call).  There is no golden log for this program in the reference tree; tests/test_spv_hmc.py
holds it to the invariants the reference itself checks at run time (reversibility, staghmc_spv.nim:1091-1160) and to
dH ~ dt^2, which fails unless every force is the gradient of the action it is paired with.
The integrators are mdevolve's Omelyan 2MN members on a shared time axis (input_hmc.xml:22-25), as restated for the
golden replay in tests/hmc_replay.py."""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qex_amd as q  # noqa: E402

LAMBDA_2MN = 0.1931833275037836      # mdevolve's Omelyan2MN default

DEFAULTS = dict(
    tau=1.0, g_steps=8, sg_steps=4, f_steps=4, pv_steps=2,                 # input_hmc.xml:18-22
    gauge_act="adjoint", beta=9.0, adj_fac=-0.25, c1=0.0,                    # :46-56
    sg_opt=1, smeared_gauge_act="Wilson", sm_beta=0.5, sm_adj_fac=0.0, sm_c1=0.0,   # :57-68
    Nf=1, mass=0.1, num_pv=2, mass_pv=0.75, bc="aaaa",                       # :44,69-76
    alpha=(0.4, 0.5, 0.5), a_tol=1e-20, f_tol=1e-12, maxits=10000,           # :80-92
    seed=987654321, start="cold",
)


def coefs(act, beta, adj, c1):
    """newGaugeAction (staghmc_spv.nim:158-197) -> (plaq, rect, adjplaq) of GaugeActionCoeffs"""
    if act == "Wilson":
        return dict(plaq=beta, rect=0.0, adjplaq=0.0)
    if act == "rect":                                   # gaugeActRect(beta, c1): plaq = (1 - 8 c1) beta, rect = c1 beta
        return dict(plaq=(1.0 - 8.0 * c1) * beta, rect=c1 * beta, adjplaq=0.0)
    if act == "adjoint":
        return dict(plaq=beta, rect=0.0, adjplaq=beta * adj)
    raise ValueError(act + " is not a valid action")


class Spv:
    def __init__(self, lat, resident=True, halo=False, ranks=None, **kw):
        """ranks = (world, rank, dist): the lattice split along t over `world` processes (QEX: -rankgeom:1,1,1,N); every rank holds
        its slab of every field, the per-site random streams are seeded by GLOBAL site index (so the fields do not depend on
        the partition), the library exchanges faces and rank-sums its own reductions, and the two sums this driver forms on
        the host (kinetic energy, pseudofermion actions) go through `dist` (torch.distributed, gloo)."""
        self.prm = dict(DEFAULTS)
        self.prm.update(kw)
        P = self.prm
        self.glat, self.resident = list(lat), resident
        self.world, self.rank, self.dist = ranks if ranks else (1, 0, None)
        lt = self.glat[3] // self.world
        self.lat = self.glat[:3] + [lt]
        self.lo = q.Layout(self.lat)
        if self.world > 1:
            self.ctx = q.Context(self.lat, device=self.rank % q.device_count(), rank_geom=(1, 1, 1, self.world), rank_coord=(0, 0, 0, self.rank))
            uid = [q.Context.unique_id() if self.rank == 0 else None]
            self.dist.broadcast_object_list(uid, src=0)
            self.ctx.comm_init(uid[0], self.world, self.rank)
        else:
            self.ctx = q.Context(self.lat)
        if halo:
            self.ctx.force_halo(True)
        self.md = q.ResidentMD(self.ctx)
        self.hc = q.HypCoefs(*P["alpha"])
        self.rng = q.RngField(self.lat, q.RngMilc6, P["seed"], glat=self.glat, t_offset=self.rank * lt) if self.world > 1 \
            else q.RngField(self.lat, q.RngMilc6, P["seed"])
        self.gact = coefs(P["gauge_act"], P["beta"], P["adj_fac"], P["c1"])
        self.sgact = coefs(P["smeared_gauge_act"], P["sm_beta"], P["sm_adj_fac"], P["sm_c1"])
        self.g = q.unit(self.lo) if P["start"] == "cold" else self.rng.warm(float(P["start"]))
        self.p = None
        self.phi = []
        self.iters = dict(action=0, force=0)
        self.sf = self.s = None

    # ---- smearing: closure + operator (staghmc_spv.nim:601-604,989-1004) ----
    def smear(self, g_host, sg_out=None):
        """gsmear.hypcoeffs.smearGetForce(g, sg); sg.rephase(); stag = newStag(sg).  g_host None = the resident links"""
        self.sf = self.hc.smearGetForce(self.ctx, g_host, sg_out)
        self.s = q.Staggered(self.ctx, None, smear=self.hc, bc=self.prm["bc"])

    def _sp(self, tol):
        return q.SolverParams(r2req=tol, maxits=self.prm["maxits"], verbosity=0)

    def _gsum(self, v):
        """rank sum of a host-side scalar (QEX: the threadRankSum at the end of norm2, commsUtils.nim:195-204)"""
        if self.world == 1:
            return float(v)
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t)
        return float(t[0])

    # ---- generate_momenta / generate_pseudoferms (:493-557) ----
    def refresh(self):
        P = self.prm
        self.p = self.rng.randomTAH()
        self.smear(self.g)
        self.phi = []
        h = self.lo.vol // 2
        for k in range(P["Nf"] + P["num_pv"]):
            psi = self.rng.gaussian_vector()
            ph = np.zeros_like(psi)
            if k < P["Nf"]:
                self.s.Ddag(ph, psi, P["mass"])                       # s.Ddag(phi, psi, masses[0])
            else:
                sp = self._sp(P["a_tol"])
                self.s.solve(ph, psi, P["mass_pv"], sp)               # s.solve(phi, psi, masses[1], spa)
                self.iters["action"] += sp.iterations
            ph[h:] = 0                                                # phi.odd := 0
            self.phi.append(ph)

    # ---- calc_action (:559-695) ----
    def action(self):
        P = self.prm
        T = 0.5 * self._gsum((self.p * self.p).sum()) - 16.0 * self.lo.vol * self.world
        ga = q.gaugeAction(self.ctx, self.g, **self.gact)
        sg = np.zeros_like(self.g) if P["sg_opt"] else None
        self.smear(self.g, sg)
        f2 = []
        for k, ph in enumerate(self.phi):
            psi = np.zeros_like(ph)
            if k < P["Nf"]:
                sp = self._sp(P["a_tol"])
                self.s.solve(psi, ph, -P["mass"], sp)                 # solve_fermion(psi, phi, -masses[0], spa)
                self.iters["action"] += sp.iterations
            else:
                self.s.D(psi, ph, P["mass_pv"])                       # s.D(psi, phi, masses[1])
            f2.append(0.5 * self._gsum((psi * psi).sum()))
        sga = q.gaugeAction(self.ctx, sg, **self.sgact) if P["sg_opt"] else 0.0    # sg_act.gaction(sgf), unphased links
        return dict(H=ga + sga + sum(f2) + T, ga=ga, sga=sga, fa=sum(f2), f2=f2, T=T)

    # ---- fforce + smeared_one_link_force (:697-865): the fields whose outer products make the force ----
    def _force_fields(self, t_f, t_pv):
        P = self.prm
        h = self.lo.vol // 2
        psis, scales = [], []
        for k, ph in enumerate(self.phi):
            if k < P["Nf"]:
                if t_f == 0.0:
                    continue
                psi = np.zeros_like(ph)
                sp = self._sp(P["f_tol"])
                self.s.solve(psi, ph, P["mass"], sp)                  # solve_fermion(psi, phi, masses[0], spf)
                self.iters["force"] += sp.iterations
                scales.append(-0.5 * t_f / P["mass"])                 # rescale (:697-713)
            else:
                if t_pv == 0.0:
                    continue
                psi = np.zeros_like(ph)
                self.s.stagD2(psi, ph, "odd", 0.0, 0.0)               # apply_massless_Ddag(psi, phi, "force"): 2 D_oe phi
                psi[:h] = ph[:h]                                      #   x.even := b
                scales.append(0.5 * (-0.5 * t_pv))
            psis.append(psi)
        return psis, scales

    # ---- mdvAllfga (:947-1043) and mdt (:873-888) ----
    def mdt(self, t):
        if self.resident:
            self.md.update_links(t)
        else:
            q.gaugeUpdate(self.ctx, self.g, self.p, t)

    def mdv_all(self, ts):
        """ts = [gauge, smeared gauge, fermion, pv] step sizes of this update (0 = member not due)"""
        res = self.resident
        if ts[1] != 0.0 or ts[2] != 0.0 or ts[3] != 0.0:
            self.smear(None if res else self.g)                       # one smearing for all sectors
        if ts[0] != 0.0:                                              # mdvg, unsmeared: g_act.gforce(g, f); mdv(ts[0])
            if res:
                self.md.gauge_force(**self.gact)
                self.md.kick(self.md.GAUGE, -ts[0])
            else:
                self.p -= ts[0] * q.gaugeForce(self.ctx, self.g, cplaq=self.gact["plaq"], rect=self.gact["rect"], adjplaq=self.gact["adjplaq"])
        f = None if res else np.zeros_like(self.g)
        if ts[1] != 0.0:                                              # sg_act.gforce(g, sg, f, smeared_force); mdv(ts[1])
            self.sf.gforce(f, **self.sgact)
            if res:
                self.md.kick(self.md.NHYP, -ts[1])
            else:
                self.p -= ts[1] * f
        if ts[2] != 0.0 or ts[3] != 0.0:                              # mdvf: stag.fforce(f, g, ...); mdv(1.0)
            psis, scales = self._force_fields(ts[2], ts[3])
            self.sf.fforce(f, psis, scales, bc=self.prm["bc"])
            if res:
                self.md.kick(self.md.NHYP, -1.0)
            else:
                self.p -= f

    def schedule(self):
        """ParallelEvolution of the 2MN members on the shared time axis (staghmc_spv.nim:1045-1061)"""
        P = self.prm
        steps = [P["g_steps"], P["sg_steps"] if P["sg_opt"] else 0, P["f_steps"], P["pv_steps"] if P["num_pv"] > 0 else 0]
        ev = []
        for m, n in enumerate(steps):
            if n <= 0:
                continue
            dt = P["tau"] / n
            for s in range(n):
                ev.append(((s + LAMBDA_2MN) * dt, m, 0.5 * dt))
                ev.append(((s + 1.0 - LAMBDA_2MN) * dt, m, 0.5 * dt))
        ev.sort(key=lambda e: (e[0], e[1]))
        out = []
        for t, m, h in ev:
            if out and abs(out[-1][0] - t) < 1e-12:
                out[-1][1][m] = h
            else:
                ts = [0.0] * 4
                ts[m] = h
                out.append((t, ts))
        return out

    def evolve(self):
        if self.resident:
            self.md.begin(self.g, self.p)
        now = 0.0
        for t, ts in self.schedule():
            self.mdt(t - now)
            now = t
            self.mdv_all(ts)
        self.mdt(self.prm["tau"] - now)
        if self.resident:
            self.md.end(self.g, self.p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-lat", type=int, nargs=4, default=[8, 8, 8, 16])
    ap.add_argument("-trajs", type=int, default=2)
    ap.add_argument("-host-fields", action="store_true", help="every MD update through the host-pointer entry points")
    ap.add_argument("-halo", action="store_true", help="every kernel in its t-sharded form on one GPU")
    ap.add_argument("-time", action="store_true")
    ap.add_argument("-start", default="cold", help="cold, or the spread of a warm start (e.g. 0.3)")
    a = ap.parse_args()
    ranks = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        dist.init_process_group("gloo")
        ranks = (dist.get_world_size(), dist.get_rank(), dist)
    hmc = Spv(a.lat, resident=not a.host_fields, halo=a.halo, start=a.start, ranks=ranks)
    if hmc.rank != 0:
        sys.stdout = open(os.devnull, "w")            # one log, rank 0's (QEX: echo prints on rank 0)
    print(hmc.ctx.info(), "transport", hmc.ctx.comm_transport()[0])
    for n in range(1, a.trajs + 1):
        t0 = time.time()
        g0 = hmc.g.copy()
        hmc.refresh()
        b = hmc.action()
        print("Begin H: %r  Sg: %r  Ssg: %r  Sf: %r  T: %r" % (float(b["H"]), float(b["ga"]), float(b["sga"]), [float(v) for v in b["f2"]], float(b["T"])))
        t1 = time.time()
        hmc.evolve()
        t2 = time.time()
        e = hmc.action()
        print("End H: %r  Sg: %r  Ssg: %r  Sf: %r  T: %r" % (float(e["H"]), float(e["ga"]), float(e["sga"]), [float(v) for v in e["f2"]], float(e["T"])))
        dH = e["H"] - b["H"]
        ok = n <= 1 or np.random.default_rng(n).random() <= math.exp(min(0.0, -dH))     # no_metropolis_until = 1
        print("%s:  dH: %r" % ("ACCEPT" if ok else "REJECT", float(dH)))
        if ok:
            q.reunit(hmc.ctx, hmc.g)
        else:
            hmc.g = g0
        pl = q.plaq(hmc.ctx, hmc.g)
        print("MEASplaq ss: %r  st: %r" % (float(2 * sum(pl[:3])), float(2 * sum(pl[3:]))))
        if a.time:
            print("TIME trajectory %d: %.2f s (MD evolution %.2f s); solver iterations %r" % (n, time.time() - t0, t2 - t1, hmc.iters))


if __name__ == "__main__":
    main()
