"""Replay of the reference's HMC regression run tests/extra/staghmc_sh (golden set G7):
src/examples/staghmc_sh.nim driven with the parameters of tests/extra/staghmc_sh/run:21-26, test 0.

This is TEST INFRASTRUCTURE: the MD driver below exists only to carry the operators under test
(nHYP smearing and its force chain, Staggered.D / solve, the adjoint-plaquette gauge force, the
fermion-force outer product) along the trajectory whose energies the reference printed, so that they
are held to the reference's own numbers on NON-TRIVIAL gauge fields.

The integrator comes from the Nim package `mdevolve` (qex.nimble:29, `mdevolve >= 1.0.0`), which is
not vendored in the reference tree.  Its schedule for this run is printed verbatim in the golden log
(ref.0:67-113) and restated here:
  gauge       Omelyan2MN, lambda 0.19, 18 steps:   T(.19) V(.5) T(.62) V(.5) T(.19)                     [x dt]
  3 fermions  Omelyan4MN3F1GP, lambda 8/27, 3 steps: T(1/8) V(l) T(3/8) [V(1-2l) + 5/972 dt^3 VTV] T(3/8) V(l) T(1/8)
  ParIntegrator, shared T: the updates of all members are ordered on the common time axis; members
  whose V updates fall on the same time are handed to ONE call of mdvAllfga(ts, gs).
The force-gradient update exp(t V + g [V,[T,V]]) is evaluated as in the example (staghmc_sh.nim:
505-640, approximateFGcoeff): shift the links by exp(-tg F), tg = 2 g / t, take the force there with
step t, restore the links (Yin & Mawhinney, arXiv:1111.5059).
"""
import numpy as np

LAT = [8, 8, 8, 8]
SEED = 987654321
BETA, ADJFAC = 6.0, -0.25
TAU = 1.0
MASS, HMASSES = 0.1, [0.2, 0.4]
RSQ = 9.999999999999999e-25          # arsq = frsq = hfrsq = pbprsq
GSTEPS, GLAMBDA = 18, 0.19
FSTEPS, FLAMBDA = 3, 0.2962962962962963
ALPHA = (0.4, 0.5, 0.5)

# tests/extra/staghmc_sh/ref.0:117-128
GOLD = {
    "begin": dict(H=18451.47947589929, Sg=0.0, Sf=[6115.074514620805, 6296.481015505035, 6143.045791623304], T=-103.1218458498552),
    "end": dict(H=18452.64279359589, Sg=18431.57360855611, Sf=[6127.428742650334, 6325.453215672831, 5587.471917645606], T=-18019.28469092899),
    "pbp": [0.2117714665683549, 0.211234484887779],
    "plaq": (0.7798927061684001, 0.7803495769561876, 0.7801211415622938),
    "ploop": (0.1593085565961168, 0.004142883358352041, 0.1806483723808761, 0.003657953473352228),
    "pbp_iters": 101,                      # ref.0:122 "stagSolve: 101"
}


def schedule():
    """[(time, [(member, t, g), ...])]: V updates on the common time axis, tau = 1."""
    ev = []
    dt = TAU / GSTEPS
    for s in range(GSTEPS):
        ev.append(((s + GLAMBDA) * dt, 0, 0.5 * dt, 0.0))
        ev.append(((s + 1.0 - GLAMBDA) * dt, 0, 0.5 * dt, 0.0))
    dt = TAU / FSTEPS
    xi = 0.005144032921810704                       # ref.0:80
    for m in (1, 2, 3):
        for s in range(FSTEPS):
            ev.append(((s + 0.125) * dt, m, FLAMBDA * dt, 0.0))
            ev.append(((s + 0.5) * dt, m, 0.4074074074074074 * dt, xi * dt ** 3))
            ev.append(((s + 0.875) * dt, m, FLAMBDA * dt, 0.0))
    ev.sort(key=lambda e: (e[0], e[1]))
    out = []
    for t, m, ts, gs in ev:
        if out and abs(out[-1][0] - t) < 1e-12:     # nonZeroStep 1e-12 (ref.0:67)
            out[-1][1].append((m, ts, gs))
        else:
            out.append((t, [(m, ts, gs)]))
    return out


def fscale(i, t):
    """staghmc_sh.nim:381-385 for one species with two Hasenbusch masses"""
    if i == 0:
        return 0.5 * t * (HMASSES[0] ** 2 - MASS ** 2) / MASS
    if i < len(HMASSES):
        return 0.5 * t * (HMASSES[i] ** 2 - HMASSES[i - 1] ** 2) / HMASSES[i - 1]
    return 0.5 * t / HMASSES[i - 1]


class Replay:
    """`be` supplies the operators under test:
         be.smear_rephase(g, want_force) -> handle     smearGetForce / smear, then setBC + stagPhase
         be.D(handle, x, m), be.solve(handle, b, m) -> (x, iterations)
         be.fermion_force(handle, g, fields, scales) -> f     fforce + smearedOneLinkForce (:387-427)
         be.gauge_force(g) -> gc.forceA(g);  be.gauge_action(g) -> gc.actionA(g)
         be.plaq(g) -> 6 plaquettes;  be.exp_update(g, p, t): g := exp(t p) g;  be.reunit(g);  be.wline(g, path)
       Only the random numbers (momenta, pseudofermion and pbp sources) are the driver's, from the oracle's RngMilc6."""

    def __init__(self, o, be):
        self.o, self.be = o, be
        self.lo = o.Layout(LAT)
        self.rf = o.RngField(self.lo, o.RNG_MILC6, SEED)
        self.g = o.gauge_unit(self.lo)
        self.p = None
        self.phi = None
        self.stats = {"force_iters": [], "action_iters": []}

    # ---- action pieces (staghmc_sh.nim:330-364) ----
    def faction(self, h):
        be, n = self.be, len(self.phi)
        fa = []
        for i in range(n - 1):
            x, its = be.solve(h, be.D(h, self.phi[i], HMASSES[i]), MASS if i == 0 else HMASSES[i - 1])
            self.stats["action_iters"].append(its)
            fa.append((x * x).sum())
        x, its = be.solve(h, self.phi[-1], HMASSES[-1])
        self.stats["action_iters"].append(its)
        fa.append((x * x).sum())
        return fa

    def energies(self, h, g):
        fa = self.faction(h)
        Sg = self.be.gauge_action(g)
        Sf = [0.5 * v for v in fa]
        T = 0.5 * (self.p * self.p).sum() - 16.0 * self.lo.vol
        return dict(H=Sg + sum(Sf) + T, Sg=Sg, Sf=Sf, T=T)

    def refresh(self):
        """staghmc_sh.nim:716-757"""
        o, lo, be = self.o, self.lo, self.be
        self.p = o.gauge_random_tah(lo, self.rf)
        h = be.smear_rephase(self.g, False)
        psi = [o.vector_gaussian(lo, self.rf) for _ in range(len(HMASSES) + 1)]
        n = len(psi)
        self.phi = []
        for i in range(n):
            mi = -MASS if i == 0 else -HMASSES[i - 1]
            ph = be.solve(h, be.D(h, psi[i], mi), -HMASSES[i])[0] if i != n - 1 else be.D(h, psi[i], mi)
            ph = ph.copy()
            ph[lo.vol // 2:] = 0
            self.phi.append(ph)
        return self.energies(h, self.g)

    # ---- MD updates (staghmc_sh.nim:429-640) ----
    def mdt(self, t):
        self.be.exp_update(self.g, self.p, t)

    def fforce(self, h, g, ix, ts):
        """fields = the force solves (:394-404); scales = fscale(k, i, ts[j])"""
        be = self.be
        fields, scales = [], []
        for j in ix:
            x, its = be.solve(h, self.phi[j], MASS if j == 0 else HMASSES[j - 1])
            self.stats["force_iters"].append(its)
            fields.append(x)
            scales.append(fscale(j, ts[j]))
        return be.fermion_force(h, g, fields, scales)

    def mdv_all(self, group):
        """mdvAllfga(ts, gs) (:505-640), first-order force-gradient approximation (useFG2 = 0)"""
        o, lo, be = self.o, self.lo, self.be
        ts, gs = [0.0] * 4, [0.0] * 4
        for m, t, g_ in group:
            ts[m], gs[m] = t, g_
        assert gs[0] == 0.0                         # the gauge member is plain 2MN in this run
        updateF = [k for k in range(3) if gs[k + 1] == 0.0 and ts[k + 1] != 0.0]
        updateFG = [k for k in range(3) if gs[k + 1] != 0.0]
        if ts[0] != 0.0:                            # mdv (:436-444): p -= t forceA(g)
            self.p -= ts[0] * be.gauge_force(self.g)
        if updateF:                                 # mdvf (:446-453): p += f
            h = be.smear_rephase(self.g, True)
            self.p += self.fforce(h, self.g, updateF, ts[1:])
        if updateFG:
            tf = {k: ts[k + 1] for k in updateFG}                          # approximateFGcoeff
            tg = {k: 2.0 * gs[k + 1] / ts[k + 1] for k in updateFG}
            gg = self.g.copy()                                             # fgsave
            h = be.smear_rephase(gg, True)
            f = self.fforce(h, gg, updateFG, [tg.get(k, 0.0) for k in range(3)])
            be.exp_update(self.g, f, 1.0)                                  # fgvf: g := exp(f) g
            h = be.smear_rephase(self.g, True)
            self.p += self.fforce(h, self.g, updateFG, [tf.get(k, 0.0) for k in range(3)])
            self.g = gg                                                    # fgload

    def evolve(self):
        now = 0.0
        for t, group in schedule():
            self.mdt(t - now)
            now = t
            self.mdv_all(group)
        self.mdt(TAU - now)

    def finish_energies(self):
        h = self.be.smear_rephase(self.g, False)
        return self.energies(h, self.g)

    # ---- measurements after ACCEPT (staghmc_sh.nim:774-789) ----
    def measure(self):
        o, lo, be = self.o, self.lo, self.be
        be.reunit(self.g)                                                  # g.reunit
        h = be.smear_rephase(self.g, False)
        pbp, iters = [], []
        for _ in range(2):                                                 # pbpreps = 2
            src = o.vector_u1(lo, self.rf)
            x, its = be.solve(h, src, MASS)
            pbp.append(MASS * (x * x).sum() / lo.vol)
            iters.append(its)
        pl = be.plaq(self.g)
        ps, pt = 2.0 * sum(pl[:3]), 2.0 * sum(pl[3:])
        loops = [be.wline(self.g, [mu + 1] * LAT[mu]) for mu in range(4)]
        pls = sum(loops[:3]) / 3.0
        return dict(pbp=pbp, pbp_iters=iters, plaq=(ps, pt, 0.5 * (ps + pt)),
                    ploop=(pls.real, pls.imag, loops[3].real, loops[3].imag))


class OracleBackend:
    def __init__(self, o, lo):
        self.o, self.lo = o, lo

    def smear_rephase(self, g, want_force):
        o, lo = self.o, self.lo
        sg = o.nhyp_smear(lo, g, *ALPHA)
        o.rephase(lo, sg)
        return dict(sg=sg, g=g.copy())

    def D(self, h, x, m):
        return self.o.D(self.lo, h["sg"], None, x, m)

    def solve(self, h, b, m):
        x, its, _ = self.o.solve(self.lo, h["sg"], None, b, m, RSQ, 1000000)
        return x, its

    def gauge_force(self, g):
        return self.o.gauge_force_general(self.lo, g, BETA, BETA * ADJFAC, 1)

    def gauge_action(self, g):
        return self.o.gauge_action(self.lo, g, BETA, BETA * ADJFAC, 1)

    def fermion_force(self, h, g, fields, scales):
        o, lo = self.o, self.lo
        f = lo.new_gauge()
        for k, (x, s) in enumerate(zip(fields, scales)):
            o.stag_outer(lo, f, x, s, s, k > 0)
        o.rephase(lo, f)                                   # f.setBC; f.stagPhase
        f[lo.vol // 2:] *= -1.0                            # odd sites
        _, f = o.nhyp_force(lo, g, f, *ALPHA)
        o.force_projTAH(lo, f, g, adj=False)
        return f

    def plaq(self, g):
        return self.o.plaq(self.lo, g)

    def exp_update(self, g, p, t):
        self.o.gauge_exp_update(self.lo, g, p, t)

    def reunit(self, g):
        self.o.gauge_projectSU(self.lo, g)

    def wline(self, g, path):
        return self.o.wline(self.lo, g, path)


class HipBackend:
    """Every operator on the hot path runs through libqexhip (C ABI)."""

    def __init__(self, q, lat):
        self.q = q
        self.ctx = q.Context(lat)
        self.hc = q.HypCoefs(*ALPHA)

    def smear_rephase(self, g, want_force):
        q = self.q
        if want_force:       # smearRephase (:303-312): one smearing feeds the closure and the operator
            sf = self.hc.smearGetForce(self.ctx, g)
            s = q.Staggered(self.ctx, None, smear=self.hc, bc="pppa")
        else:                # smearRephaseDiscardForce (:314-324)
            sf, s = None, q.Staggered(self.ctx, g, smear=self.hc, bc="pppa")
        return dict(s=s, sf=sf)

    def D(self, h, x, m):
        r = np.zeros_like(x)
        h["s"].D(r, x, m)
        return r

    def solve(self, h, b, m):
        sp = self.q.SolverParams(r2req=RSQ, maxits=1000000, verbosity=0)
        x = np.zeros_like(b)
        h["s"].solve(x, b, m, sp)
        return x, sp.iterations

    def gauge_force(self, g):
        return self.q.gaugeForce(self.ctx, g, cplaq=BETA, adjplaq=BETA * ADJFAC)

    def gauge_action(self, g):
        return self.q.gaugeAction(self.ctx, g, plaq=BETA, adjplaq=BETA * ADJFAC)

    def exp_update(self, g, p, t):
        self.q.gaugeUpdate(self.ctx, g, p, t)

    def reunit(self, g):
        self.q.reunit(self.ctx, g)

    def wline(self, g, path):
        return self.q.wline(self.ctx, path, g)

    def fermion_force(self, h, g, fields, scales):
        f = np.zeros_like(g)
        h["sf"].fforce(f, fields, scales, bc="pppa")
        return f

    def plaq(self, g):
        return self.q.plaq(self.ctx, g)
