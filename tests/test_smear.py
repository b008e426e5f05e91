"""Link construction upstream of the solver (forward direction):
fat7 / HISQ (src/gauge/fat7l.nim, src/physics/hisqLinks.nim) and nHYP (src/gauge/hypsmear.nim).

The reference holds no known-answer vector for these (its only test, tests/examples/testStagProp.nim,
asserts not-NaN), so the oracle restatement is pinned by properties that fix every coefficient and
index: free-field normalisation (9/8 and 1/24 for HISQ with staggered phases, identity for nHYP),
gauge covariance, unitarity; the HIP path is then held to the oracle.
"""
import numpy as np
import pytest

cx = lambda a: a[..., 0] + 1j * a[..., 1]


def _transform(o, lo, g, seed=0):
    G = np.zeros((lo.vol, 3, 3, 2))
    for i in range(lo.vol):
        G[i] = o.su3_fn("qo_projectSU", np.random.default_rng(seed + i).standard_normal((3, 3, 2)))
    Gc, gc = cx(G), cx(g)
    nb = [np.array([lo.neighbor(i, mu, 1) for i in range(lo.vol)]) for mu in range(4)]
    g2 = np.zeros_like(g)
    for mu in range(4):
        m = np.einsum("nij,njk,nlk->nil", Gc, gc[:, mu], Gc[nb[mu]].conj())
        g2[:, mu, :, :, 0], g2[:, mu, :, :, 1] = m.real, m.imag
    return Gc, g2


def test_oracle_smearing_properties(oracle):
    o = oracle
    lo = o.Layout([4, 4, 4, 6])
    # free field with staggered phases: HISQ gives the Naik-improved normalisation 9/8 and 1/24
    gu = o.gauge_unit(lo)
    o.stagPhase(lo, gu)
    fl, ll = o.hisq_smear(lo, gu)
    assert np.abs(fl[:, :, 0, 0, 0] / gu[:, :, 0, 0, 0] - 1.125).max() < 1e-14
    assert np.abs(np.abs(ll[:, :, 0, 0, 0]) - 1.0 / 24.0).max() < 1e-15
    assert np.abs(fl[:, :, 0, 1]).max() == 0
    # ... and the long link is naik * the product of the three phased links
    for i in range(0, lo.vol, 17):
        for mu in range(4):
            j, k = lo.neighbor(i, mu, 1), lo.neighbor(i, mu, 2)
            assert abs(ll[i, mu, 0, 0, 0] - (-1.0 / 24.0) * gu[i, mu, 0, 0, 0] * gu[j, mu, 0, 0, 0] * gu[k, mu, 0, 0, 0]) < 1e-15
    # nHYP of the unit field is the unit field
    g1 = o.gauge_unit(lo)
    assert np.abs(o.nhyp_smear(lo, g1) - g1).max() < 1e-15
    # gauge covariance + unitarity on a rough field
    g = o.gauge_warm(lo, 0.5, o.RngField(lo, o.RNG_MILC6, 3))
    Gc, g2 = _transform(o, lo, g)
    nb1 = [np.array([lo.neighbor(i, mu, 1) for i in range(lo.vol)]) for mu in range(4)]
    nb3 = [np.array([lo.neighbor(i, mu, 3) for i in range(lo.vol)]) for mu in range(4)]
    f1, f2 = cx(o.nhyp_smear(lo, g)), cx(o.nhyp_smear(lo, g2))
    h1, h2 = [cx(a) for a in o.hisq_smear(lo, g)], [cx(a) for a in o.hisq_smear(lo, g2)]
    for mu in range(4):
        assert np.abs(np.einsum("nij,njk,nlk->nil", Gc, f1[:, mu], Gc[nb1[mu]].conj()) - f2[:, mu]).max() < 1e-11
        assert np.abs(np.einsum("nij,njk,nlk->nil", Gc, h1[0][:, mu], Gc[nb1[mu]].conj()) - h2[0][:, mu]).max() < 1e-11
        assert np.abs(np.einsum("nij,njk,nlk->nil", Gc, h1[1][:, mu], Gc[nb3[mu]].conj()) - h2[1][:, mu]).max() < 1e-11
    m = f1.reshape(-1, 3, 3)
    assert np.abs(np.einsum("nij,nkj->nik", m, m.conj()) - np.eye(3)).max() < 1e-13      # nHYP links are unitary
    # alpha = 0 is the identity map up to projectU of an already unitary link
    assert np.abs(o.nhyp_smear(lo, g, 0.0, 0.0, 0.0) - g).max() < 1e-13
    # fat7 with only the one-link term is a rescaling
    fl, _ = o.fat7(lo, g, (0.7, 0, 0, 0, 0))
    assert np.abs(fl - 0.7 * g).max() < 1e-15


@pytest.mark.gpu
def test_gpu_hisq_and_fat7(oracle):
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    g = o.gauge_random(lo, seed=987654321)
    o.rephase(lo, g)                                  # g.setBC; g.stagPhase (testStagProp.nim:24-26)
    ctx = q.Context(lat)
    fl, ll = np.zeros_like(g), np.zeros_like(g)
    hc = q.HisqCoefs().init()
    hc.smear(ctx, g, fl, ll)
    rfl, rll = o.hisq_smear(lo, g)
    assert np.linalg.norm(fl - rfl) / np.linalg.norm(rfl) < 1e-13
    assert np.linalg.norm(ll - rll) / np.linalg.norm(rll) < 1e-13
    # a fat7 with every term switched on, incl. Lepage and Naik
    coef = (0.9, -0.11, 0.021, -0.0043, -0.07)
    q.makeImpLinks(ctx, fl, g, coef, ll, naik=-0.05)
    rfl, rll = o.fat7(lo, g, coef, naik=-0.05)
    assert np.linalg.norm(fl - rfl) / np.linalg.norm(rfl) < 1e-13
    assert np.linalg.norm(ll - rll) / np.linalg.norm(rll) < 1e-13
    # the smeared links drive the Naik operator exactly like the oracle's (testStagProp.nim:34-50)
    s = q.newStag3(ctx, fl, ll)
    v1 = np.zeros((lo.vol, 3, 2))
    v1[0, 0, 0] = 1.0
    v2 = np.zeros_like(v1)
    s.D(v2, v1, 0.001)
    assert np.isfinite(v2).all()
    assert np.linalg.norm(v2 - o.D(lo, rfl, rll, v1, 0.001)) / np.linalg.norm(v2) < 1e-12


@pytest.mark.gpu
def test_gpu_nhyp(oracle):
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    g = o.gauge_warm(lo, 0.5, o.RngField(lo, o.RNG_MILC6, 7))
    ctx = q.Context(lat)
    fl = np.zeros_like(g)
    q.HypCoefs(0.4, 0.5, 0.5).smear(ctx, g, fl)        # the alphas of tests/extra/staghmc_sh (ref.0:66-70)
    ref = o.nhyp_smear(lo, g, 0.4, 0.5, 0.5)
    assert np.linalg.norm(fl - ref) / np.linalg.norm(ref) < 1e-12
    assert q.plaq(ctx, fl).sum() > q.plaq(ctx, g).sum()   # smearing smooths


@pytest.mark.gpu
def test_gpu_operator_on_device_smeared_links(oracle):
    """Staggered(..., smear=...) smears on the device and feeds the Dslash directly; it must act
    exactly like smear -> (rephase) -> newStag through host memory, and like the oracle."""
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 99)
    g = o.gauge_warm(lo, 0.5, rf)
    x = o.vector_gaussian(lo, rf)
    ctx = q.Context(lat)
    y, y2 = np.zeros_like(x), np.zeros_like(x)
    # HISQ: phases first, then smear (testStagProp.nim:24-40)
    gp = g.copy()
    o.rephase(lo, gp)
    s = q.Staggered(ctx, gp, smear=q.HisqCoefs().init())
    s.D(y, x, 0.05)
    rfl, rll = o.hisq_smear(lo, gp)
    assert np.linalg.norm(y - o.D(lo, rfl, rll, x, 0.05)) / np.linalg.norm(y) < 1e-12
    # nHYP: smear the unphased field, then BC + phases (staghmc_spv.nim:601-604)
    for bc in ("pppa", "aaaa", "pppp"):
        s = q.Staggered(ctx, g, smear=q.HypCoefs(0.4, 0.5, 0.5), bc=bc)
        s.D(y, x, 0.05)
        sm = o.nhyp_smear(lo, g, 0.4, 0.5, 0.5)
        for mu, ch in enumerate(bc):                   # setBC_cust (staghmc_spv.nim:367-390)
            if ch == "a":
                last = np.array([lo.coord(i)[mu] == lat[mu] - 1 for i in range(lo.vol)])
                sm[last, mu] *= -1.0
        o.stagPhase(lo, sm)
        assert np.linalg.norm(y - o.D(lo, sm, None, x, 0.05)) / np.linalg.norm(y) < 1e-12
        q.newStag(ctx, sm).D(y2, x, 0.05)
        assert np.linalg.norm(y - y2) / np.linalg.norm(y) < 1e-13


def test_oracle_nhyp_force_is_the_gradient(oracle):
    """The restated chain rule (projectUderiv, symStapleDeriv, smearedForce) is held to the definition
    the reference states and checks for projectUderiv itself (matrixFunctions.nim:323-327,570-591):
    d Re tr(C^+ V(U)) = Re tr(dU^+ F)."""
    o = oracle
    rng = np.random.default_rng(1)
    for _ in range(3):
        X, Cm, d = rng.standard_normal((3, 3, 2)), rng.standard_normal((3, 3, 2)), 1e-6 * rng.standard_normal((3, 3, 2))
        S = lambda X: (Cm * o.su3_fn("qo_projectU", X)).sum()
        num, ana = (S(X + d) - S(X - d)) / 2, (d * o.projectUderiv(X, Cm)).sum()
        assert abs(num - ana) < 1e-7 * abs(ana)
    lo = o.Layout([4, 4, 4, 6])
    g = o.gauge_warm(lo, 0.5, o.RngField(lo, o.RNG_MILC6, 3))
    Cf = rng.standard_normal(g.shape)
    fl, f = o.nhyp_force(lo, g, Cf, 0.4, 0.5, 0.5)
    assert np.abs(fl - o.nhyp_smear(lo, g, 0.4, 0.5, 0.5)).max() == 0
    S = lambda g: (Cf * o.nhyp_smear(lo, g, 0.4, 0.5, 0.5)).sum()
    for t in range(3):
        d = np.zeros_like(g)
        if t == 0:
            d = 1e-6 * rng.standard_normal(g.shape)
        else:
            d[int(rng.integers(lo.vol)), int(rng.integers(4))] = 1e-5 * rng.standard_normal((3, 3, 2))
        num, ana = (S(g + d) - S(g - d)) / 2, (d * f).sum()
        assert abs(num - ana) < 1e-7 * abs(ana)
    # alpha = 0: V = P(U) and the chain is projectUderiv alone
    _, f0 = o.nhyp_force(lo, g, Cf, 0.0, 0.0, 0.0)
    i = 77
    assert np.abs(f0[i, 2] - o.projectUderiv(g[i, 2], Cf[i, 2])).max() < 1e-13


@pytest.mark.gpu
def test_gpu_nhyp_force_chain(oracle):
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 21)
    g = o.gauge_warm(lo, 0.5, rf)
    chain = o.gauge_random_tah(lo, rf) + 0.3 * o.gauge_random(lo, rf)      # a generic (non-algebra) chain
    ctx = q.Context(lat)
    fl = np.zeros_like(g)
    smearedForce = q.HypCoefs(0.4, 0.5, 0.5).smearGetForce(ctx, g, fl)
    rfl, rf_ = o.nhyp_force(lo, g, chain, 0.4, 0.5, 0.5)
    assert np.linalg.norm(fl - rfl) / np.linalg.norm(rfl) < 1e-12
    f = np.zeros_like(g)
    smearedForce(f, chain)
    assert np.linalg.norm(f - rf_) / np.linalg.norm(rf_) < 1e-11
    f2 = chain.copy()
    smearedForce(f2, f2)                                  # f.smeared_force(f) (staghmc_spv.nim:740)
    assert np.array_equal(f, f2)
    smearedForce.release()
    with pytest.raises(q.QexHipError):
        smearedForce(f, chain)


@pytest.mark.gpu
def test_gpu_nhyp_md_forces(oracle):
    """The fork's two MD forces through the closure: gforce(act, g, sg, f, smear_force)
    (staghmc_spv.nim:217-228) and fforce + smeared_one_link_force (staghmc_spv.nim:716-865)."""
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 33)
    g = o.gauge_warm(lo, 0.5, rf)
    ctx = q.Context(lat)
    sg = np.zeros_like(g)
    sf = q.HypCoefs(0.4, 0.5, 0.5).smearGetForce(ctx, g, sg)
    # gauge sector, three actions
    for cp, c2, kind in [(1.0, 0.0, 0), (5.0 / 3.0, -1.0 / 12.0, 0), (0.9, 0.35, 1)]:
        f = np.zeros_like(g)
        sf.gforce(f, plaq=cp, rect=c2 if kind == 0 else 0.0, adjplaq=c2 if kind == 1 else 0.0)
        chain = o.gauge_deriv_general(lo, sg, cp, c2, kind)
        _, ref = o.nhyp_force(lo, g, chain, 0.4, 0.5, 0.5)
        o.force_projTAH(lo, ref, g, adj=True)
        assert np.linalg.norm(f - ref) / np.linalg.norm(ref) < 1e-11
    # matter sector: two fields, different scales, bc = "aaaa" (input_hmc.xml:44) and the default "pppa"
    psis = [o.vector_gaussian(lo, rf), o.vector_gaussian(lo, rf)]
    scales = [0.37, -1.9]
    odd = np.arange(lo.vol) >= lo.vol // 2
    for bc in ("aaaa", "pppa"):
        f = np.zeros_like(g)
        sf.fforce(f, psis, scales, bc=bc)
        ref = np.zeros_like(g)
        for k, (p, s) in enumerate(zip(psis, scales)):
            o.stag_outer(lo, ref, p, s, s, k > 0)
        for mu, ch in enumerate(bc):                   # f.rephase (staghmc_spv.nim:723)
            if ch == "a":
                last = np.array([lo.coord(i)[mu] == lat[mu] - 1 for i in range(lo.vol)])
                ref[last, mu] *= -1.0
        o.stagPhase(lo, ref)
        ref[odd] *= -1.0                               # odd sites re-signed (:730-732)
        _, ref = o.nhyp_force(lo, g, ref, 0.4, 0.5, 0.5)
        o.force_projTAH(lo, ref, g, adj=False)
        assert np.linalg.norm(f - ref) / np.linalg.norm(ref) < 1e-11
        if bc == "pppa":
            # fforce incl. its solves in one call == solve each field, then fforce (staghmc_sh.nim:387-427)
            s_op = q.Staggered(ctx, None, smear=q.HypCoefs(0.4, 0.5, 0.5), bc=bc)      # links from the closure
            phis = [p.copy() for p in psis]
            for p in phis:
                p[lo.vol // 2:] = 0
            f1 = np.zeros_like(g)
            its = sf.fforce_solve(f1, phis, [0.1, 0.2], scales, 1e-20, bc=bc)
            sols = []
            for p, m in zip(phis, [0.1, 0.2]):
                x = np.zeros_like(p)
                sp = q.SolverParams(r2req=1e-20, maxits=100000, verbosity=0)
                s_op.solve(x, p, m, sp)
                sols.append(x)
                assert sp.iterations == its[len(sols) - 1]
            f2 = np.zeros_like(g)
            sf.fforce(f2, sols, scales, bc=bc)
            assert np.linalg.norm(f1 - f2) / np.linalg.norm(f2) < 1e-13
        # the force is in the algebra
        fc = cx(f)
        assert np.abs(fc + np.conj(np.swapaxes(fc, -1, -2))).max() < 1e-12
        assert np.abs(np.trace(fc, axis1=-2, axis2=-1)).max() < 1e-12


def test_oracle_hisq_force_is_the_gradient(oracle):
    """fat7lDeriv / the HISQ chain (fat7lderiv.nim, hisqsmear.nim:55-90) restated as reverse accumulation over the
    oracle's own staple graph, held to the definition  d[sum Re tr(C^+ links(U))] = sum Re tr(dU^+ F)."""
    o = oracle
    lo = o.Layout([4, 4, 4, 6])
    g = o.gauge_warm(lo, 0.5, o.RngField(lo, o.RNG_MILC6, 3))
    o.rephase(lo, g)
    rng = np.random.default_rng(2)
    Cf, Cl = rng.standard_normal(g.shape), rng.standard_normal(g.shape)
    coef = (0.9, -0.11, 0.021, -0.0043, -0.07)                    # every term on, incl. Lepage
    cases = [
        (lambda u: sum((c * f).sum() for c, f in zip((Cf, Cl), o.fat7(lo, u, coef, naik=-0.05))), o.fat7_deriv(lo, g, Cf, coef, Cl, -0.05)),
        (lambda u: sum((c * f).sum() for c, f in zip((Cf, Cl), o.hisq_smear(lo, u))), o.hisq_force(lo, g, Cf, Cl)),
    ]
    for S, F in cases:
        for t in range(3):
            d = np.zeros_like(g)
            if t == 0:
                d = 1e-6 * rng.standard_normal(g.shape)
            else:
                d[int(rng.integers(lo.vol)), int(rng.integers(4))] = 1e-5 * rng.standard_normal((3, 3, 2))
            num, ana = (S(g + d) - S(g - d)) / 2, (d * F).sum()
            assert abs(num - ana) < 1e-6 * abs(ana)


@pytest.mark.gpu
def test_gpu_hisq_force(oracle):
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 8)
    g = o.gauge_warm(lo, 0.5, rf)
    o.rephase(lo, g)
    dfl, dll = o.gauge_random_tah(lo, rf) + 0.2 * o.gauge_random(lo, rf), o.gauge_random_tah(lo, rf)
    ctx = q.Context(lat)
    coef = (0.9, -0.11, 0.021, -0.0043, -0.07)
    d = q.fat7lDeriv(ctx, g, dfl, coef, dll, naik=-0.05)
    ref = o.fat7_deriv(lo, g, dfl, coef, dll, -0.05)
    assert np.linalg.norm(d - ref) / np.linalg.norm(ref) < 1e-13
    d = q.fat7lDeriv(ctx, g, dfl, (0.7, -0.1, 0, 0, 0))            # 3-staple only, no long links
    ref = o.fat7_deriv(lo, g, dfl, (0.7, -0.1, 0, 0, 0))
    assert np.linalg.norm(d - ref) / np.linalg.norm(ref) < 1e-13
    f = q.HisqCoefs().init().force(ctx, g, dfl, dll)
    ref = o.hisq_force(lo, g, dfl, dll)
    assert np.linalg.norm(f - ref) / np.linalg.norm(ref) < 1e-12
    # the closure form (hisqsmear.nim:55-90): smear once, reverse pass per call, operator from the closure's links
    fl, ll = np.zeros_like(g), np.zeros_like(g)
    sf = q.HisqCoefs().init().smearGetForce(ctx, g, fl, ll)
    rfl, rll = o.hisq_smear(lo, g)
    assert np.linalg.norm(fl - rfl) / np.linalg.norm(rfl) < 1e-13 and np.linalg.norm(ll - rll) / np.linalg.norm(rll) < 1e-13
    for chains in ((dfl, dll), (dll, dfl)):
        f2 = np.zeros_like(g)
        sf(f2, *chains)
        assert np.linalg.norm(f2 - o.hisq_force(lo, g, *chains)) / np.linalg.norm(ref) < 1e-12
    s = q.Staggered(ctx, None, smear=q.HisqCoefs().init())
    xv = o.vector_gaussian(lo, rf)
    y = np.zeros_like(xv)
    s.D(y, xv, 0.02)
    assert np.linalg.norm(y - o.D(lo, rfl, rll, xv, 0.02)) / np.linalg.norm(y) < 1e-12
    # fermionForce of hisqhmc.nim:496-541: 1-hop and 3-hop outer products, odd sign, chain, TAH(ff u^+)
    psis, scales = [o.vector_gaussian(lo, rf), o.vector_gaussian(lo, rf)], [0.8, -0.3]
    ff = np.zeros_like(g)
    sf.fermionForce(ff, psis, scales)
    f1, f3 = np.zeros_like(g), np.zeros_like(g)
    for k, (p, sc) in enumerate(zip(psis, scales)):
        o.stag_outer(lo, f1, p, sc, -sc, k > 0, hop=1)
        o.stag_outer(lo, f3, p, sc, -sc, k > 0, hop=3)
    rff = o.hisq_force(lo, g, f1, f3)
    o.force_projTAH(lo, rff, g, adj=False)
    assert np.linalg.norm(ff - rff) / np.linalg.norm(rff) < 1e-12
    sf.release()
    with pytest.raises(q.QexHipError, match="prepare"):
        sf(f2, dfl, dll)


@pytest.mark.gpu
def test_gpu_smearing_and_forces_on_the_sharded_path(oracle):
    """t-sharded link construction and force chains (ghost slices for every matrix field that is read at shifted
    sites, face exchange after each such field is produced), on one GPU with forced ghost zones: must reproduce the
    periodic kernels -- fat7 / HISQ links, their derivative, nHYP links, the nHYP closure and both MD forces, and
    the operators built from device-smeared links."""
    import qex_amd as q

    o = oracle
    lat = [8, 8, 8, 8]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 61)
    g = o.gauge_warm(lo, 0.5, rf)
    gp = g.copy()
    o.rephase(lo, gp)
    dfl, dll = o.gauge_random_tah(lo, rf) + 0.2 * o.gauge_random(lo, rf), o.gauge_random_tah(lo, rf)
    x = o.vector_gaussian(lo, rf)
    psis = [o.vector_gaussian(lo, rf), o.vector_gaussian(lo, rf)]
    A, B, Cc = q.Context(lat), q.Context(lat), q.Context(lat)
    B.force_halo(True)                 # nHYP levels communication-avoiding (one depth-3 thin-link exchange, levels on shrinking ghost slices)
    Cc.force_halo(True)
    Cc.set_option("smear_ca", 0)       # ... and with the per-field ghost refreshes of rounds 1-4
    Cc.set_option("chain_overlap", 0)  # ... and the force chain's levels in one pass behind their exchange (B: ghost-free slices beside it)
    eq = lambda a, b: np.linalg.norm(a - b) <= 1e-15 * np.linalg.norm(a)
    coef = (0.9, -0.11, 0.021, -0.0043, -0.07)
    out = {}
    for name, ctx in (("A", A), ("B", B), ("C", Cc)):
        r = {}
        fl, ll = np.zeros_like(g), np.zeros_like(g)
        q.makeImpLinks(ctx, fl, gp, coef, ll, naik=-0.05)
        r["fat7"] = (fl.copy(), ll.copy())
        q.HisqCoefs().init().smear(ctx, gp, fl, ll)
        r["hisq"] = (fl.copy(), ll.copy())
        r["fat7d"] = q.fat7lDeriv(ctx, gp, dfl, coef, dll, naik=-0.05)
        r["hisqf"] = q.HisqCoefs().init().force(ctx, gp, dfl, dll)
        sg = np.zeros_like(g)
        hc = q.HypCoefs(0.4, 0.5, 0.5)
        hc.smear(ctx, g, sg)
        r["nhyp"] = sg.copy()
        sf = hc.smearGetForce(ctx, g, sg)
        r["nhyp2"] = sg.copy()
        f = np.zeros_like(g)
        sf(f, dfl)
        r["chain"] = f.copy()
        for k, kw in enumerate([dict(plaq=1.0), dict(plaq=5.0 / 3.0, rect=-1.0 / 12.0), dict(plaq=6.0, adjplaq=-1.5)]):
            sf.gforce(f, **kw)
            r["gforce%d" % k] = f.copy()
        sf.fforce(f, psis, [0.37, -1.9], bc="aaaa")
        r["fforce"] = f.copy()
        s = q.Staggered(ctx, None, smear=hc, bc="pppa")               # operator on the closure's links
        y = np.zeros_like(x)
        s.D(y, x, 0.05)
        r["D_nhyp"] = y.copy()
        phis = [p.copy() for p in psis]
        for p in phis:
            p[lo.vol // 2:] = 0
        its = sf.fforce_solve(f, phis, [0.1, 0.2], [0.37, -1.9], 1e-20, bc="pppa")
        r["fsolve"], r["its"] = f.copy(), its
        s = q.Staggered(ctx, gp, smear=q.HisqCoefs().init())
        s.D(y, x, 0.05)
        r["D_hisq"] = y.copy()
        out[name] = r
    a = out["A"]
    for b in (out["B"], out["C"]):
        assert a["its"] == b["its"]
        for k in a:
            if k == "its":
                continue
            if isinstance(a[k], tuple):
                assert all(eq(u, v) for u, v in zip(a[k], b[k])), k
            else:
                tol = 1e-9 if k == "fsolve" else 1e-15                     # solves: summation order of the slab reductions
                assert np.linalg.norm(a[k] - b[k]) <= tol * np.linalg.norm(a[k]), k


FAT7_SELFTEST = [("oneLink", (1, 0, 0, 0, 0)), ("threeStaple", (0, 1, 0, 0, 0)), ("fiveStaple", (0, 0, 1, 0, 0)),
                 ("sevenStaple", (0, 0, 0, 1, 0)), ("lepage", (0, 0, 0, 0, 1)), ("all", (1, 1, 1, 1, 1))]


def _checkfat1(plaq_of_fat, coef):
    """checkfat1 of the reference's own fat7l self-test (src/gauge/fat7l.nim:184-214): on the unit gauge field every fat link
    is c * 1 with c = oneLink + 6 threeStaple + 24 fiveStaple + 48 sevenStaple + 6 lepage (the number of staples of each
    kind), so each of the six plaquettes is c^4 / 6; the reference prints `relerr: sqrt(sum (p - s)^2) / s`."""
    c = coef[0] + 6 * coef[1] + 24 * coef[2] + 48 * coef[3] + 6 * coef[4]
    s = c ** 4 / 6.0
    return float(np.sqrt(((np.asarray(plaq_of_fat) - s) ** 2).sum()) / s)


def test_oracle_replays_the_fat7l_selftest(oracle):
    o = oracle
    lo = o.Layout([8, 8, 8, 8])                       # defaultLat of the self-test, unit gauge (defaultSetup)
    g = o.gauge_unit(lo)
    for name, coef in FAT7_SELFTEST:
        fl, _ = o.fat7(lo, g, coef)
        assert _checkfat1(o.plaq(lo, fl), coef) < 1e-14, name
    # the closing lines of the self-test: makeImpLinks with the long links, naik = 1: ll = U U U = 1 -> plaquettes 1/6
    fl, ll = o.fat7(lo, g, (1, 1, 1, 1, 1), naik=1.0)
    assert np.abs(o.plaq(lo, ll) - 1.0 / 6.0).max() < 1e-15 and _checkfat1(o.plaq(lo, fl), (1, 1, 1, 1, 1)) < 1e-14


@pytest.mark.gpu
def test_gpu_replays_the_fat7l_selftest(oracle):
    """the same procedure through qexhip_fat7 and the HIP plaquette kernel (ctx.plaq of the fat links)"""
    import qex_amd as q

    lat = [8, 8, 8, 8]
    g = q.unit(q.Layout(lat))
    ctx = q.Context(lat)
    for name, coef in FAT7_SELFTEST:
        fl = np.zeros_like(g)
        q.makeImpLinks(ctx, fl, g, coef)
        assert _checkfat1(q.plaq(ctx, fl), coef) < 1e-14, name
    fl, ll = np.zeros_like(g), np.zeros_like(g)
    q.makeImpLinks(ctx, fl, g, (1, 1, 1, 1, 1), ll=ll, naik=1.0)
    assert np.abs(q.plaq(ctx, ll) - 1.0 / 6.0).max() < 1e-15
