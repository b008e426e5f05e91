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
RSQ = 9.999999999999999e-25          # arsq = frsq = hfrsq = pbprsq
PBPMASS = 0.1
ALPHA = (0.4, 0.5, 0.5)
XI = 0.005144032921810704            # VTV coefficient of 4MN3F1GP at lambda = 8/27 (ref.0:80)
LAM = 0.2962962962962963


class Config:
    """One run of tests/extra/staghmc_sh/run (test 0, 1 or 2)."""

    def __init__(self, name, masses, hmasses, galg, gsteps, fsteps, hfsteps, gold):
        self.name, self.masses, self.hmasses = name, masses, hmasses
        self.galg, self.gsteps, self.fsteps, self.hfsteps = galg, gsteps, fsteps, hfsteps
        self.gold = gold
        # flattened pseudofermion fields: (species k, level i); level 0 = the light mass, i > 0 = Hasenbusch i-1
        self.fields = [(k, i) for k in range(len(masses)) for i in range(len(hmasses[k]) + 1)]
        self.steps = [[fsteps[k]] + list(hfsteps[k]) for k in range(len(masses))]


def _parse_check(n):
    """tests/golden/staghmc_sh/ref.N.check = the lines of the reference's golden log that its own harness compares
    (tests/extra/staghmc_sh/run:43-44) plus the solver statistics; extracted by tests/golden/make_staghmc_fixtures.py"""
    import os
    import re

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "staghmc_sh", "ref.%d.check" % n)
    num = r"[-+0-9.eE]+"
    traj, cur = [], None
    for line in open(path):
        m = re.match(r"(Begin|End|Reversed) H: (%s)  Sg: (%s)  Sf: @\[(.*)\]  T: (%s)" % (num, num, num), line)
        if m:
            e = dict(H=float(m.group(2)), Sg=float(m.group(3)), T=float(m.group(5)),
                     Sf=[float(v) for v in re.findall(num, m.group(4).replace("@", ""))])
            if m.group(1) == "Begin":
                cur = dict(begin=e, pbp=[], pbp_iters=[], force_stats=[], action_stats=[])
                traj.append(cur)
            else:
                cur[{"End": "end", "Reversed": "reversed"}[m.group(1)]] = e
            continue
        if cur is None:
            continue
        if line.startswith(("ACCEPT", "REJECT")):
            cur["accept"] = line.startswith("ACCEPT")
        elif line.startswith("stagSolve:"):
            cur["pbp_iters"].append(int(line.split()[1]))
        elif line.startswith("MEASpbp"):
            cur["pbp"].append(float(line.split(":")[1]))
        elif line.startswith("MEASplaq"):
            cur["plaq"] = tuple(float(v) for v in re.findall(r": +(%s)" % num, line))
        elif line.startswith("MEASploop"):
            cur["ploop"] = tuple(float(v) for v in re.findall(num, line.split("spatial:")[1].replace("temporal:", " ")))
        elif line.startswith("Solver[pbp]:"):
            m = re.match(r"Solver\[pbp\]: (\d+): (\d+):(\d+)", line)
            cur["pbp_stats"] = (int(m.group(1)), int(m.group(2)), int(m.group(3)))
        elif re.match(r"  [AF] m=", line):
            m = re.match(r"  ([AF]) m=\S+ (\d+): (\d+):(\d+)", line)
            cur["force_stats" if m.group(1) == "F" else "action_stats"].append((int(m.group(2)), int(m.group(3)), int(m.group(4))))
    return traj


def _gold(n, second):
    t = _parse_check(n)
    g = dict(begin=t[0]["begin"], end=t[0]["end"], accept=t[0]["accept"], pbp=t[0]["pbp"], plaq=t[0]["plaq"], ploop=t[0]["ploop"],
             pbp_iters=t[0]["pbp_stats"][2], force_stats=t[0]["force_stats"], action_max=[a[2] for a in t[0]["action_stats"]])
    if second:
        g.update(begin2=t[1]["begin"], end2=t[1]["end"], reversed2=t[1]["reversed"], accept2=t[1]["accept"], pbp2=t[1]["pbp"],
                 plaq2=t[1]["plaq"])
    return g


# parameters of tests/extra/staghmc_sh/run:21-36 (tests 0, 1, 2); golden numbers from the committed log extracts
CONFIGS = {
    0: Config("ref.0", [0.1], [[0.2, 0.4]], ("2MN", 0.19), 18, [3], [[3, 3]], _gold(0, True)),
    1: Config("ref.1", [0.1, 0.05], [[0.2, 0.4], [0.2, 0.4]], ("4MN3F1GP", LAM), 8, [4, 1], [[4, 4], [4, 4]], _gold(1, False)),
    2: Config("ref.2", [0.1, 0.05], [[0.2, 0.4], [0.2, 0.4]], ("4MN3F1GP", LAM), 8, [4, 1], [[2, 8], [4, 4]], _gold(2, False)),
}
GOLD = CONFIGS[0].gold                      # kept for the Begin-H-only tests


def _member_events(alg, steps, member):
    """V updates (time, member, t, g) of one Omelyan member over tau = 1 (coefficients: ref.0:67-113)"""
    ev, dt = [], TAU / steps
    if alg[0] == "2MN":
        lam = alg[1]
        for s in range(steps):
            ev.append(((s + lam) * dt, member, 0.5 * dt, 0.0))
            ev.append(((s + 1.0 - lam) * dt, member, 0.5 * dt, 0.0))
    elif alg[0] == "4MN3F1GP":
        lam = alg[1]
        assert abs(lam - LAM) < 1e-15                # XI belongs to this lambda
        for s in range(steps):
            ev.append(((s + 0.125) * dt, member, lam * dt, 0.0))
            ev.append(((s + 0.5) * dt, member, (1.0 - 2.0 * lam) * dt, XI * dt ** 3))
            ev.append(((s + 0.875) * dt, member, lam * dt, 0.0))
    else:
        raise ValueError(alg)
    return ev


def schedule(cfg):
    """[(time, [(member, t, g), ...])]: member 0 = gauge, member 1 + j = pseudofermion field j"""
    ev = _member_events(cfg.galg, cfg.gsteps, 0)
    for j, (k, i) in enumerate(cfg.fields):
        ev += _member_events(("4MN3F1GP", LAM), cfg.steps[k][i], 1 + j)
    ev.sort(key=lambda e: (e[0], e[1]))
    out = []
    for t, m, ts, gs in ev:
        if out and abs(out[-1][0] - t) < 1e-12:     # nonZeroStep 1e-12 (ref.0:67)
            out[-1][1].append((m, ts, gs))
        else:
            out.append((t, [(m, ts, gs)]))
    return out


class Replay:
    """`be` supplies the operators under test:
         be.smear_rephase(g, want_force) -> handle     smearGetForce / smear, then setBC + stagPhase
         be.D(handle, x, m), be.solve(handle, b, m) -> (x, iterations)
         be.solve_many(handle, [b], [m]) -> [(x, iterations)]   independent systems on the same links
         be.fermion_force(handle, g, fields, scales) -> f     fforce + smearedOneLinkForce (:387-427)
         be.gauge_force(g) -> gc.forceA(g);  be.gauge_action(g) -> gc.actionA(g)
         be.plaq(g) -> 6 plaquettes;  be.exp_update(g, p, t): g := exp(t p) g;  be.reunit(g);  be.wline(g, path)
       The random numbers (momenta, pseudofermion and pbp sources) come from `rng` (see __init__)."""

    def __init__(self, o, be, cfg=None, rng=None, ranks=None):
        """rng: object with randomTAH() / gaussian_vector() / u1_vector() drawing from newRNGField(RngMilc6, SEED);
        default = the oracle's; the GPU replay passes the product's own (qex_amd.RngField).
        ranks = (world, rank, dist): the lattice split along t over `world` processes; every field below is this rank's slab
        (the backend's context carries the rank geometry, the rng is seeded by GLOBAL site index), and the three sums the driver
        forms on the host -- kinetic energy, pseudofermion actions, pbp -- are rank-summed through `dist` (gloo)."""
        self.o, self.be = o, be
        self.cfg = cfg or CONFIGS[0]
        self.world, self.rank, self.dist = ranks if ranks else (1, 0, None)
        self.lo = o.Layout(LAT[:3] + [LAT[3] // self.world])
        self.gvol = LAT[0] * LAT[1] * LAT[2] * LAT[3]
        self.rng = rng or OracleRng(o, self.lo)
        self.g = o.gauge_unit(self.lo)
        self.p = None
        self.phi = None                              # phi[j] for the flattened fields
        self.stats = {"force_iters": [[] for _ in self.cfg.fields], "action_iters": [[] for _ in self.cfg.fields]}

    def gsum(self, v):
        """rank sum of a host-side scalar (the threadRankSum behind QEX's norm2, commsUtils.nim:195-204)"""
        if self.world == 1:
            return v
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.dist.all_reduce(t)
        return float(t[0])

    def _m(self, j):
        """mass the field j is solved with: light mass at level 0, else the previous Hasenbusch mass"""
        k, i = self.cfg.fields[j]
        return self.cfg.masses[k] if i == 0 else self.cfg.hmasses[k][i - 1]

    def _last(self, j):
        k, i = self.cfg.fields[j]
        return i == len(self.cfg.hmasses[k])

    def fscale(self, j, t):
        """staghmc_sh.nim:381-385"""
        k, i = self.cfg.fields[j]
        hm, m = self.cfg.hmasses[k], self.cfg.masses[k]
        if len(hm) == 0:
            return 0.5 * t / m
        if i == 0:
            return 0.5 * t * (hm[0] ** 2 - m ** 2) / m
        if i < len(hm):
            return 0.5 * t * (hm[i] ** 2 - hm[i - 1] ** 2) / hm[i - 1]
        return 0.5 * t / hm[i - 1]

    # ---- action pieces (staghmc_sh.nim:330-364) ----
    def faction(self, h):
        be, cfg = self.be, self.cfg
        srcs, ms = [], []
        for j, (k, i) in enumerate(cfg.fields):
            srcs.append(self.phi[j] if self._last(j) else be.D(h, self.phi[j], cfg.hmasses[k][i]))
            ms.append(cfg.hmasses[k][-1] if self._last(j) and cfg.hmasses[k] else self._m(j))
        fa = []
        for j, (x, its) in enumerate(be.solve_many(h, srcs, ms)):     # the chain's solves share the links
            self.stats["action_iters"][j].append(its)
            fa.append(self.gsum((x * x).sum()))
        return fa

    def energies(self, h, g):
        fa = self.faction(h)
        Sg = self.be.gauge_action(g)
        Sf = [0.5 * v for v in fa]
        T = 0.5 * self.gsum((self.p * self.p).sum()) - 16.0 * self.gvol
        return dict(H=Sg + sum(Sf) + T, Sg=Sg, Sf=Sf, T=T)

    def refresh(self):
        """staghmc_sh.nim:716-757"""
        o, lo, be, cfg = self.o, self.lo, self.be, self.cfg
        self.p = self.rng.randomTAH()
        h = be.smear_rephase(self.g, False)
        # psi[k][i].gaussian r, level by level across the species ("conforms to bsm.lua", :735-745)
        psi = {}
        for lvl in range(max(len(hm) for hm in cfg.hmasses) + 1):
            for k in range(len(cfg.masses)):
                if lvl <= len(cfg.hmasses[k]):
                    psi[(k, lvl)] = self.rng.gaussian_vector()
        self.phi = []
        for j, (k, i) in enumerate(cfg.fields):
            mi = -self._m(j)
            if self._last(j):
                ph = be.D(h, psi[(k, i)], mi)
            else:
                ph = be.solve(h, be.D(h, psi[(k, i)], mi), -cfg.hmasses[k][i])[0]
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
        if hasattr(be, "fforce_solve"):               # solves + outer products + chain in one call
            f, its = be.fforce_solve(h, g, [self.phi[j] for j in ix], [self._m(j) for j in ix],
                                     [self.fscale(j, ts[j]) for j in ix])
            for j, n in zip(ix, its):
                self.stats["force_iters"][j].append(n)
            return f
        fields, scales = [], []
        for j, (x, its) in zip(ix, be.solve_many(h, [self.phi[j] for j in ix], [self._m(j) for j in ix])):
            self.stats["force_iters"][j].append(its)
            fields.append(x)
            scales.append(self.fscale(j, ts[j]))
        return be.fermion_force(h, g, fields, scales)

    def mdv_all(self, group):
        """mdvAllfga(ts, gs) (:505-640), first-order force-gradient approximation (useFG2 = 0):
        exp(t V + g [V,[T,V]]) ~ links shifted by exp(-(2g/t) F), force taken there with step t"""
        be = self.be
        nf = len(self.cfg.fields)
        ts, gs = [0.0] * (nf + 1), [0.0] * (nf + 1)
        for m, t, g_ in group:
            ts[m], gs[m] = t, g_
        updateGG = gs[0] != 0.0
        updateG = (not updateGG) and ts[0] != 0.0
        updateF = [k for k in range(nf) if gs[k + 1] == 0.0 and ts[k + 1] != 0.0]
        updateFG = [k for k in range(nf) if gs[k + 1] != 0.0]
        gg, hshared = None, None
        if updateGG or updateFG:
            gg = self.g.copy()                                             # fgsave
            if updateFG:
                hshared = be.smear_rephase(gg, True)                       # sforceShared
        if updateG:                                                        # mdv: p -= t forceA(g)
            self.p -= ts[0] * be.gauge_force(self.g)
        if updateF:                                                        # mdvf: p += f
            h = hshared if updateFG else be.smear_rephase(self.g, True)
            self.p += self.fforce(h, self.g, updateF, ts[1:])
        if updateGG or updateFG:
            if updateGG:                                                   # fgv: g := exp(-tg forceA(gg)) g
                be.exp_update(self.g, be.gauge_force(gg), -2.0 * gs[0] / ts[0])
            if updateFG:                                                   # fgvf: g := exp(f) g, f at gg with steps tg
                tg = [2.0 * gs[k + 1] / ts[k + 1] if k in updateFG else 0.0 for k in range(nf)]
                be.exp_update(self.g, self.fforce(hshared, gg, updateFG, tg), 1.0)
            if updateGG:                                                   # FG mdv at the shifted links
                self.p -= ts[0] * be.gauge_force(self.g)
            if updateFG:                                                   # FG mdvf at the shifted links
                h = be.smear_rephase(self.g, True)
                self.p += self.fforce(h, self.g, updateFG, ts[1:])
            self.g = gg                                                    # fgload

    # ---- the same MD with links and momenta resident on the device (backend.resident, qexhip_md_*) ----
    def _fforce_resident(self, ix, ts):
        its = self.be.md_fforce_solve([self.phi[j] for j in ix], [self._m(j) for j in ix], [self.fscale(j, ts[j]) for j in ix])
        for j, n in zip(ix, its):
            self.stats["force_iters"][j].append(n)

    def _mdv_all_resident(self, group):
        """mdv_all step for step; nothing crosses PCIe but the pseudofermion fields of the solves"""
        be = self.be
        nf = len(self.cfg.fields)
        ts, gs = [0.0] * (nf + 1), [0.0] * (nf + 1)
        for m, t, g_ in group:
            ts[m], gs[m] = t, g_
        updateGG = gs[0] != 0.0
        updateG = (not updateGG) and ts[0] != 0.0
        updateF = [k for k in range(nf) if gs[k + 1] == 0.0 and ts[k + 1] != 0.0]
        updateFG = [k for k in range(nf) if gs[k + 1] != 0.0]
        if updateGG or updateFG:
            be.md_save()                                                   # fgsave
            if updateFG:
                be.md_smear()                                              # sforceShared (closure + operator from gg)
        if updateG:
            be.md_gauge_force()
            be.md_kick(0, -ts[0])
        if updateF:
            if not updateFG:
                be.md_smear()
            self._fforce_resident(updateF, ts[1:])
            be.md_kick(1, 1.0)
        if updateGG or updateFG:
            if updateGG:                                                   # fgv
                be.md_gauge_force()
                be.md_shift(0, -2.0 * gs[0] / ts[0])
            if updateFG:                                                   # fgvf, closure still the one of gg
                tg = [2.0 * gs[k + 1] / ts[k + 1] if k in updateFG else 0.0 for k in range(nf)]
                self._fforce_resident(updateFG, tg)
                be.md_shift(1, 1.0)
            if updateGG:
                be.md_gauge_force()
                be.md_kick(0, -ts[0])
            if updateFG:
                be.md_smear()
                self._fforce_resident(updateFG, ts[1:])
                be.md_kick(1, 1.0)
            be.md_restore()                                                # fgload

    def evolve(self):
        if getattr(self.be, "resident", False):
            self.be.md_begin(self.g, self.p)
            now = 0.0
            for t, group in schedule(self.cfg):
                self.be.md_T(t - now)
                now = t
                self._mdv_all_resident(group)
            self.be.md_T(TAU - now)
            self.be.md_end(self.g, self.p)
            return
        now = 0.0
        for t, group in schedule(self.cfg):
            self.mdt(t - now)
            now = t
            self.mdv_all(group)
        self.mdt(TAU - now)

    def finish_energies(self):
        h = self.be.smear_rephase(self.g, False)
        return self.energies(h, self.g)

    def reverse_check(self):
        """revCheck (staghmc_sh.nim:642-680): flip the momenta, evolve again, report the energies, restore"""
        g1, p1 = self.g.copy(), self.p.copy()
        self.p = -self.p
        self.evolve()
        e = self.finish_energies()
        self.g, self.p = g1, p1
        return e

    # ---- measurements after ACCEPT / REJECT (staghmc_sh.nim:774-789) ----
    def measure(self, accepted=True, g0=None):
        o, lo, be = self.o, self.lo, self.be
        if accepted:
            be.reunit(self.g)                                              # g.reunit
        else:
            self.g = g0.copy()                                             # g := g0; stag0.pbp uses sg0
        h = be.smear_rephase(self.g, False)
        pbp, iters = [], []
        srcs = [self.rng.u1_vector() for _ in range(2)]                    # pbpreps = 2
        for x, its in be.solve_many(h, srcs, [PBPMASS, PBPMASS]):
            pbp.append(PBPMASS * self.gsum((x * x).sum()) / self.gvol)
            iters.append(its)
        pl = be.plaq(self.g)
        ps, pt = 2.0 * sum(pl[:3]), 2.0 * sum(pl[3:])
        loops = [be.wline(self.g, [mu + 1] * LAT[mu]) for mu in range(4)]
        pls = sum(loops[:3]) / 3.0
        return dict(pbp=pbp, pbp_iters=iters, plaq=(ps, pt, 0.5 * (ps + pt)),
                    ploop=(pls.real, pls.imag, loops[3].real, loops[3].imag))


class OracleRng:
    def __init__(self, o, lo):
        self.o, self.lo, self.rf = o, lo, o.RngField(lo, o.RNG_MILC6, SEED)

    def randomTAH(self):
        return self.o.gauge_random_tah(self.lo, self.rf)

    def gaussian_vector(self):
        return self.o.vector_gaussian(self.lo, self.rf)

    def u1_vector(self):
        return self.o.vector_u1(self.lo, self.rf)


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

    def solve_many(self, h, bs, ms):
        return [self.solve(h, b, m) for b, m in zip(bs, ms)]

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

    def __init__(self, q, lat, halo=False, resident=False, ranks=None):
        """halo=True: every kernel runs in its t-sharded form (ghost zones, face exchanges, rank reductions) on one GPU.
        resident=True: the MD evolution keeps links and momenta on the device (qexhip_md_*, Replay.evolve).
        ranks = (world, rank, dist): this process holds one t-slab of `lat` (GLOBAL extents) among `world` ranks"""
        self.q = q
        if ranks:
            world, rank, dist = ranks
            loc = list(lat[:3]) + [lat[3] // world]
            self.ctx = q.Context(loc, device=rank % q.device_count(), rank_geom=(1, 1, 1, world), rank_coord=(0, 0, 0, rank))
            uid = [q.Context.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            self.ctx.comm_init(uid[0], world, rank)
        else:
            self.ctx = q.Context(lat)
        if halo:
            self.ctx.force_halo(True)
        self.hc = q.HypCoefs(*ALPHA)
        self.resident = resident
        self.md = q.ResidentMD(self.ctx)
        self._sf = None

    # ---- resident MD (Replay._mdv_all_resident) ----
    def md_begin(self, g, p):
        self.md.begin(g, p)

    def md_end(self, g, p):
        self.md.end(g, p)

    def md_T(self, t):
        self.md.update_links(t)

    def md_save(self):
        self.md.save_links()

    def md_restore(self):
        self.md.restore_links()

    def md_smear(self):
        """smearRephase on the resident links: closure + operator links, nothing over PCIe"""
        self._sf = self.hc.smearGetForce(self.ctx, None)
        self._s = self.q.Staggered(self.ctx, None, smear=self.hc, bc="pppa")

    def md_gauge_force(self):
        self.md.gauge_force(plaq=BETA, adjplaq=BETA * ADJFAC)

    def md_kick(self, source, t):
        self.md.kick(source, t)

    def md_shift(self, source, t):
        self.md.shift_links(source, t)

    def md_fforce_solve(self, phis, masses, scales):
        return self._sf.fforce_solve(None, phis, masses, scales, RSQ, bc="pppa")

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

    def solve_many(self, h, bs, ms):
        """lock-step batches of up to four systems (qexhip_stag_solve_batch): same result per system"""
        out = []
        for i0 in range(0, len(bs), 4):
            b4, m4 = bs[i0:i0 + 4], ms[i0:i0 + 4]
            xs = [np.zeros_like(b) for b in b4]
            sps = [self.q.SolverParams(r2req=RSQ, maxits=1000000, verbosity=0) for _ in b4]
            h["s"].solve_batch(xs, b4, m4, sps)
            out += [(x, sp.iterations) for x, sp in zip(xs, sps)]
        return out

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

    def fforce_solve(self, h, g, phis, masses, scales):
        f = np.zeros_like(g)
        its = h["sf"].fforce_solve(f, phis, masses, scales, RSQ, bc="pppa")
        return f, its

    def plaq(self, g):
        return self.q.plaq(self.ctx, g)


class DeviceEndsReplay(Replay):
    """The first trajectory of a run with BOTH ENDS on the device as well (round 3): momenta, pseudofermion and pbp sources are
    drawn by the RngMilc6 field INTO HBM (RngField.dev_*), D / solve / norms run on resident fields, the energies come from the
    resident links and momenta, and so do reunit, pbp, plaquettes and Polyakov loops.  Between `refresh` and `measure` nothing
    but scalars and the 36-byte generator states crosses PCIe.  Needs HipBackend(resident=True)."""

    def __init__(self, o, be, cfg, rng):
        super().__init__(o, be, cfg, rng=rng)
        self.ctx = be.ctx
        self.q = be.q
        self.q.gaugeSet(self.ctx, self.g)             # the unit start goes up once
        self.phi_ids = None
        self._tmp = [self.ctx.field_new() for _ in range(2 * len(self.cfg.fields) + 4)]

    def _faction_dev(self):
        be, cfg, ctx = self.be, self.cfg, self.ctx
        nf = len(cfg.fields)
        srcs, ms = [], []
        for j, (k, i) in enumerate(cfg.fields):
            if self._last(j):
                srcs.append(self.phi_ids[j])
            else:
                t = self._tmp[j]
                ctx.dev_D(t, self.phi_ids[j], cfg.hmasses[k][i])
                srcs.append(t)
            ms.append(cfg.hmasses[k][-1] if self._last(j) and cfg.hmasses[k] else self._m(j))
        xs = self._tmp[nf:2 * nf]
        its, _ = ctx.dev_solve_batch(xs, srcs, ms, RSQ)
        fa = []
        for j in range(nf):
            self.stats["action_iters"][j].append(its[j])
            fa.append(ctx.dev_norm2(xs[j]))
        return fa

    def energies_dev(self):
        fa = self._faction_dev()
        Sg = self.q.gaugeAction(self.ctx, None, plaq=BETA, adjplaq=BETA * ADJFAC)
        Sf = [0.5 * v for v in fa]
        T = 0.5 * self.be.md.momentum_norm2() - 16.0 * self.lo.vol
        return dict(H=Sg + sum(Sf) + T, Sg=Sg, Sf=Sf, T=T)

    def refresh(self):
        be, cfg, ctx, rng = self.be, self.cfg, self.ctx, self.rng
        rng.dev_momenta(ctx)                          # p.randomTAH r, born in HBM
        be.md.begin(None, None)                       # links and momenta are the resident ones
        be.md_smear()                                 # closure + operator from the resident links
        psi = {}
        for lvl in range(max(len(hm) for hm in cfg.hmasses) + 1):
            for k in range(len(cfg.masses)):
                if lvl <= len(cfg.hmasses[k]):
                    psi[(k, lvl)] = ctx.field_new()
                    rng.dev_gaussian_vector(ctx, psi[(k, lvl)])
        for old in getattr(self, "phi_ids", None) or []:      # the previous trajectory's pseudofermions (resident fields)
            ctx.field_free(old)
        self.phi_ids = []
        a, b = self._tmp[-1], self._tmp[-2]
        for j, (k, i) in enumerate(cfg.fields):
            mi = -self._m(j)
            ph = ctx.field_new()
            if self._last(j):
                ctx.dev_D(ph, psi[(k, i)], mi)
            else:
                ctx.dev_D(a, psi[(k, i)], mi)
                ctx.dev_solve_batch([ph], [a], [-cfg.hmasses[k][i]], RSQ)
            ctx.dev_zero(ph, "odd")
            self.phi_ids.append(ph)
        for f in psi.values():
            ctx.field_free(f)
        return self.energies_dev()

    def _fforce_resident(self, ix, ts):
        its = self.be._sf.fforce_solve_dev(None, [self.phi_ids[j] for j in ix], [self._m(j) for j in ix],
                                           [self.fscale(j, ts[j]) for j in ix], RSQ, bc="pppa")
        for j, n in zip(ix, its):
            self.stats["force_iters"][j].append(n)

    def evolve(self):
        now = 0.0
        for t, group in schedule(self.cfg):
            self.be.md_T(t - now)
            now = t
            self._mdv_all_resident(group)
        self.be.md_T(TAU - now)

    def finish_energies(self):
        self.be.md_smear()
        return self.energies_dev()

    def measure(self, accepted=True, g0=None):
        assert accepted, "the device-ends replay covers the ACCEPT branch"
        from qex_amd._lib import check, lib
        ctx, q = self.ctx, self.q
        check(lib().qexhip_gauge_reunit(ctx._h))      # g.reunit on the resident links
        self.be.md_smear()
        srcs = [ctx.field_new() for _ in range(2)]
        for s in srcs:
            self.rng.dev_u1_vector(ctx, s)
        xs = self._tmp[:2]
        its, _ = ctx.dev_solve_batch(xs, srcs, [PBPMASS, PBPMASS], RSQ)
        pbp = [PBPMASS * ctx.dev_norm2(x) / self.lo.vol for x in xs]
        for s in srcs:
            ctx.field_free(s)
        pl = q.plaq(ctx)
        ps, pt = 2.0 * sum(pl[:3]), 2.0 * sum(pl[3:])
        loops = q.ploops(ctx)                                  # the four g.wline([mu+1] * L_mu) of `ploop` in one call
        pls = sum(loops[:3]) / 3.0
        return dict(pbp=pbp, pbp_iters=list(its), plaq=(ps, pt, 0.5 * (ps + pt)),
                    ploop=(pls.real, pls.imag, loops[3].real, loops[3].imag))
