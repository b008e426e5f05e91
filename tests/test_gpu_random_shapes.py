"""GPU parity on a sweep of lattice SHAPES (edge cases of the site / tile / neighbour arithmetic): extents 2 (forward and
backward neighbour coincide), rows shorter and longer than a 64-site tile, ragged last tiles, tile positions whose slots do
and do not pair the parities (k_force_lds2 / k_force_lds), shapes that can and cannot be sharded in t.  Every shape: the
one-parity Dslash, D, the first CG residuals, plaquettes, plaquette force and two Wilson-flow steps against the oracle
(src/physics/stagD.nim:349-395, src/solvers/cg.nim:132-214, src/gauge/gaugeUtils.nim:213-282, src/gauge/wflow.nim:21-67);
with forced ghost zones where the shape allows it; with Naik links where every extent is >= 4; nHYP smearing and its force chain
(hypsmear.nim:49-247) on every shape, the HISQ links (hisqLinks.nim:32-43) where the 3-hop terms fit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 987654321


def relerr(a, b):
    return np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)


def shapes():
    fixed = [[2, 2, 2, 2], [2, 2, 2, 4], [12, 2, 2, 4], [2, 4, 6, 8], [6, 6, 6, 8], [4, 4, 4, 4], [16, 4, 2, 6], [10, 6, 4, 4],
             [4, 12, 2, 8], [8, 8, 4, 6]]
    rng = np.random.default_rng(20261004)
    out = list(fixed)
    while len(out) < 16:
        lat = [int(v) for v in rng.choice([2, 4, 6, 8, 10, 12], size=4)]
        if 32 <= np.prod(lat) <= 6000 and lat not in out:
            out.append(lat)
    return out


@pytest.mark.parametrize("lat", shapes(), ids=lambda l: "x".join(map(str, l)))
def test_shape(oracle, lat):
    import qex_amd as q

    o = oracle
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, SEED)
    g0 = o.gauge_random(lo, rf)                 # unphased: gauge sector
    g = g0.copy()
    o.rephase(lo, g)                            # BC + staggered phases: fermion sector
    naik = min(lat) >= 4
    g3 = None
    if naik:
        g3 = o.gauge_random(lo, rf)
        o.rephase(lo, g3)
        g3 *= 0.3
    x = o.vector_gaussian(lo, rf)
    y = o.vector_gaussian(lo, rf)
    can_shard = (lat[0] * lat[1] * lat[2] // 2) % 64 == 0 and lat[3] >= (6 if naik else 4)
    for halo in ([False, True] if can_shard else [False]):
        ctx = q.Context(lat)
        if halo:
            ctx.force_halo(True)
        for use_naik in ([False, True] if naik else [False]):
            s = q.newStag3(ctx, g, g3) if use_naik else q.newStag(ctx, g)
            l3 = g3 if use_naik else None
            for sub, par in (("even", 0), ("odd", 1), ("all", 2)):
                r_gpu, r_ref = y.copy(), y.copy()
                s.stagD2(r_gpu, x, sub, 1.5, -0.7)
                o.stagD2(lo, g, l3, r_ref, x, par, 1.5, -0.7)
                assert relerr(r_gpu, r_ref) < 1e-13, (lat, halo, use_naik, sub)
            r = np.zeros_like(x)
            s.D(r, x, 0.1)
            assert relerr(r, o.D(lo, g, l3, x, 0.1)) < 1e-13
            for par_even in (True, False):
                sp = q.SolverParams(r2req=1e-30, maxits=12, verbosity=0)
                xs = np.zeros_like(x)
                s.solveXX(xs, x, 0.2, sp, parEven=par_even, histcap=64)
                xr, its, fin, hist = o.solveXX(lo, g, l3, x, 0.2, 1e-30, 12, par_even, histcap=64)
                n = min(len(hist), len(sp.r2hist))
                assert n >= 2 and sp.iterations == its
                assert np.abs(sp.r2hist[:n] / hist[:n] - 1).max() < 1e-9, (lat, halo, use_naik, par_even)
                assert relerr(xs, xr) < 1e-9
            # shifted systems (cgm.nim:84-315, stagSolve.nim:296-345)
            masses = [0.2, 0.5, 1.1]
            shifts = [masses[0]] + [4.0 * (m * m - masses[0] ** 2) for m in masses[1:]]
            xm = [np.zeros_like(x) for _ in masses]
            sp = q.SolverParams(r2req=1e-30, maxits=10, verbosity=0)
            s.solveXX_multi(xm, x, shifts, sp, parEven=True, histcap=64)
            xr, its, hist = o.solveXX_multi(lo, g, l3, x, shifts, 1e-30, 10, True, histcap=64)
            n = min(len(hist), len(sp.r2hist))
            assert sp.iterations == its and np.abs(sp.r2hist[:n] / hist[:n] - 1).max() < 1e-9
            h = lo.vol // 2
            for a, b in zip(xm, xr):
                assert relerr(a[:h], b[:h]) < 1e-9, (lat, halo, use_naik)
        # gauge sector
        assert np.max(np.abs(q.plaq(ctx, g0) - o.plaq(lo, g0))) < 1e-14
        assert relerr(q.gaugeForce(ctx, g0), o.gauge_force(lo, g0)) < 1e-13
        for fe in (1, 0):
            ctx.set_option("flow_exp", fe)
            gf, gr = g0.copy(), g0.copy()
            q.gaugeFlow(ctx, gf, 2, 0.02)
            o.wflow(lo, gr, 2, 0.02)
            assert relerr(gf, gr) < 1e-12, (lat, halo, fe)
        pl, eq = q.flowMeasure(ctx)
        assert np.max(np.abs(pl - o.plaq(lo, gr))) < 1e-13
        # Polyakov loops (gauge_flow.nim:137-156): x lines by a shuffle tree over padded power-of-two groups, the others a lane per line
        pls = q.ploops(ctx)
        for d in range(4):
            assert abs(pls[d] - o.wline(lo, gr, [d + 1] * lat[d])) < 1e-14, (lat, halo, d)
            assert abs(q.wline(ctx, [-(d + 1)] * lat[d]) - pls[d].conjugate()) < 1e-15
        # the fork's action-selectable force and flow (flow/flow.nim:22-90): rectangle (2-hop: extents >= 4) and adjoint
        for cp, c2, kind, act in ((5.0 / 3.0, -1.0 / 12.0, 0, "rect"), (0.9, 0.35, 1, "adj")):
            if kind == 0 and not naik:
                continue
            kw = dict(rect=c2 if kind == 0 else 0.0, adjplaq=c2 if kind == 1 else 0.0)
            assert relerr(q.gaugeForce(ctx, g0, cplaq=cp, **kw), o.gauge_force_general(lo, g0, cp, c2, kind)) < 1e-13, (lat, halo, act)
            gf, gr2 = g0.copy(), g0.copy()
            q.gaugeFlow(ctx, gf, 2, 0.01, flow_act=act, plaq=cp, **kw)
            o.wflow_general(lo, gr2, 2, 0.01, cp, c2, kind)
            assert relerr(gf, gr2) < 1e-12, (lat, halo, act)
        q.gaugeSet(ctx, gr)
        # smearing and its chain rule (hypsmear.nim:49-247; hisqLinks.nim:32-43 where the 3-hop terms fit)
        gw = o.gauge_warm(lo, 0.5, o.RngField(lo, o.RNG_MILC6, 21))
        chain = o.gauge_random_tah(lo, rf) + 0.3 * o.gauge_random(lo, rf)
        fl = np.zeros_like(gw)
        sf = q.HypCoefs(0.4, 0.5, 0.5).smearGetForce(ctx, gw, fl)
        rfl, rfo = o.nhyp_force(lo, gw, chain, 0.4, 0.5, 0.5)
        assert relerr(fl, rfl) < 1e-12, (lat, halo)
        f = np.zeros_like(gw)
        sf(f, chain)
        assert relerr(f, rfo) < 1e-11, (lat, halo)
        sf.release()
        if naik:
            hf, hl = np.zeros_like(gw), np.zeros_like(gw)
            q.HisqCoefs().smear(ctx, gw, hf, hl)
            rhf, rhl = o.hisq_smear(lo, gw)
            assert relerr(hf, rhf) < 1e-12 and relerr(hl, rhl) < 1e-12, (lat, halo)
        ctx.close()
