"""General gauge actions of the flow path (plaq + rect, plaq + adjplaq).

CPU: the oracle's forces are the derivatives of the oracle's actions (restated from
gaugeAction1 / actionA, src/gauge/gaugeAction.nim:61-142,614-681) -- a normalisation check that
does not depend on the code under test.  GPU: the HIP force / flow against the oracle.
"""
import numpy as np
import pytest


def _tah(rng):
    a = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
    t = 0.5 * (a - a.conj().T)
    return t - np.trace(t) / 3 * np.eye(3)


def _expm(t):
    w, v = np.linalg.eig(t)
    return v @ np.diag(np.exp(w)) @ np.linalg.inv(v)


@pytest.mark.parametrize("cp,c2,kind", [(1.3, 0.0, 0), (1.7, -0.12, 0), (0.9, 0.35, 1)])
def test_force_is_minus_gradient_of_action(oracle, cp, c2, kind):
    """d/de S(exp(eT) U_mu(x)) = -Re tr(T F_mu(x)) with F = gaugeForce / forceA."""
    o = oracle
    lo = o.Layout([4, 4, 4, 6])
    g = o.gauge_warm(lo, 0.5, o.RngField(lo, o.RNG_MILC6, 5))
    F = o.gauge_force_general(lo, g, cp, c2, kind)
    Fc = F[..., 0] + 1j * F[..., 1]
    rng = np.random.default_rng(3)
    for _ in range(3):
        x, mu, T, eps = int(rng.integers(lo.vol)), int(rng.integers(4)), _tah(rng), 1e-5
        U = g[x, mu, :, :, 0] + 1j * g[x, mu, :, :, 1]
        S = []
        for e in (eps, -eps):
            g2 = g.copy()
            M = _expm(e * T) @ U
            g2[x, mu, :, :, 0], g2[x, mu, :, :, 1] = M.real, M.imag
            S.append(o.gauge_action(lo, g2, cp, c2, kind))
        num = (S[0] - S[1]) / (2 * eps)
        ana = -np.trace(T @ Fc[x, mu]).real
        assert abs(num - ana) < 1e-6 * max(1.0, abs(ana))
    # with c2 = 0 both code paths reduce to the Wilson force that golden set G2 pins
    assert np.abs(o.gauge_force_general(lo, g, cp, 0.0, kind) - cp * o.gauge_force(lo, g)).max() < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("cp,c2,kind", [(1.0, 0.0, 0), (1.7, -0.12, 0), (5.0 / 3.0, -1.0 / 12.0, 0), (0.9, 0.35, 1)])
def test_gpu_force_general(oracle, cp, c2, kind):
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    g = o.gauge_warm(lo, 0.4, o.RngField(lo, o.RNG_MILC6, 11))
    ctx = q.Context(lat)
    f = q.gaugeForce(ctx, g, cplaq=cp, rect=c2 if kind == 0 else 0.0, adjplaq=c2 if kind == 1 else 0.0)
    ref = o.gauge_force_general(lo, g, cp, c2, kind)
    assert np.linalg.norm(f - ref) / np.linalg.norm(ref) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("act,cp,c2", [("rect", 5.0 / 3.0, -1.0 / 12.0), ("adj", 1.0, 0.25)])
def test_gpu_flow_general(oracle, act, cp, c2):
    """gc.gaugeFlow(flow_act, g, steps, eps) of src/flow/flow.nim:22-90 (Symanzik and adjoint flows)."""
    import qex_amd as q

    o = oracle
    lat = [4, 4, 8, 4]
    lo = o.Layout(lat)
    g = o.gauge_warm(lo, 0.4, o.RngField(lo, o.RNG_MILC6, 12))
    gref = g.copy()
    ctx = q.Context(lat)
    kind = 1 if act == "adj" else 0
    q.gaugeFlow(ctx, g, 3, 0.01, flow_act=act, plaq=cp, rect=c2 if kind == 0 else 0.0, adjplaq=c2 if kind else 0.0)
    o.wflow_general(lo, gref, 3, 0.01, cp, c2, kind)
    assert np.linalg.norm(g - gref) / np.linalg.norm(gref) < 1e-12
    p0, p1 = o.plaq(lo, gref).sum(), q.plaq(ctx, g).sum()
    assert abs(p0 - p1) < 1e-13


@pytest.mark.gpu
def test_gpu_md_building_blocks(oracle):
    """gauge action (three kinds), mdt link update, reunit, Wilson / Polyakov lines on the device."""
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 19)
    g = o.gauge_warm(lo, 0.4, rf)
    p = o.gauge_random_tah(lo, rf)
    ctx = q.Context(lat)
    for cp, c2, kind in [(1.0, 0.0, 0), (5.0 / 3.0, -1.0 / 12.0, 0), (6.0, -1.5, 1)]:
        a = q.gaugeAction(ctx, g, plaq=cp, rect=c2 if kind == 0 else 0.0, adjplaq=c2 if kind == 1 else 0.0)
        ref = o.gauge_action(lo, g, cp, c2, kind)
        assert abs(a - ref) < 1e-12 * max(1.0, abs(ref))
    g1, g2 = g.copy(), g.copy()
    q.gaugeUpdate(ctx, g1, p, 0.37)
    o.gauge_exp_update(lo, g2, p, 0.37)
    assert np.linalg.norm(g1 - g2) / np.linalg.norm(g2) < 1e-14
    g1 *= 1.0 + 1e-9                                    # drift off the group, then reunit
    g2 = g1.copy()
    q.reunit(ctx, g1)
    o.gauge_projectSU(lo, g2)
    assert np.linalg.norm(g1 - g2) / np.linalg.norm(g2) < 1e-14
    for path in ([4] * lat[3], [1] * lat[0], [1, 2, -1, -2], [-3, 4, 4, 3, -4, -4], [2]):
        w = q.wline(ctx, path, g1)
        assert abs(w - o.wline(lo, g1, path)) < 1e-14
    # s4_gauge (staghmc_spv_meas.nim:25-65): even/odd plaquette sums per direction; every plaquette is counted for its two
    # directions, so the eight numbers add up to 8 x the sum of the six plaquettes of g.plaq
    s4 = q.s4_gauge(ctx, g1)
    assert np.abs(s4 - o.s4_gauge(lo, g1)).max() < 1e-14
    assert abs(s4.sum() - 8.0 * q.plaq(ctx).sum()) < 1e-13
    # the four Polyakov loops in one call (meas_ploop, gauge_flow.nim:137-156): one lane per LINE instead of per site
    pl = q.ploops(ctx, g1)
    for d in range(4):
        assert abs(pl[d] - o.wline(lo, g1, [d + 1] * lat[d])) < 1e-14
        assert abs(pl[d] - q.wline(ctx, [d + 1] * lat[d])) < 1e-15 and abs(pl[d].conjugate() - q.wline(ctx, [-(d + 1)] * lat[d])) < 1e-15


@pytest.mark.gpu
def test_gpu_gauge_sector_on_the_sharded_path(oracle):
    """t-sharded gauge-sector kernels (ghost slices of the links, virtual-slice indexing, face exchange to depth
    1 / 2 / 3, rank reductions), exercised on one GPU with forced ghost zones: plaquettes, the three actions and
    forces, Wilson / Symanzik / adjoint flow, clover observables, MD update, reunit == the periodic kernels."""
    import qex_amd as q

    o = oracle
    lat = [8, 8, 8, 8]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 23)
    g = o.gauge_warm(lo, 0.4, rf)
    p = o.gauge_random_tah(lo, rf)
    A, B = q.Context(lat), q.Context(lat)
    B.force_halo(True)
    assert "halo=1" in B.info()
    assert np.array_equal(q.plaq(A, g), q.plaq(B, g))
    for cp, c2, kind in [(1.0, 0.0, 0), (5.0 / 3.0, -1.0 / 12.0, 0), (6.0, -1.5, 1)]:
        kw = dict(plaq=cp, rect=c2 if kind == 0 else 0.0, adjplaq=c2 if kind == 1 else 0.0)
        assert q.gaugeAction(A, g, **kw) == q.gaugeAction(B, g, **kw)
        fa = q.gaugeForce(A, g, cplaq=cp, rect=kw["rect"], adjplaq=kw["adjplaq"])
        fb = q.gaugeForce(B, g, cplaq=cp, rect=kw["rect"], adjplaq=kw["adjplaq"])
        assert np.array_equal(fa, fb)
        ga, gb = g.copy(), g.copy()
        act = {0: "rect" if c2 else "Wilson", 1: "adj"}[kind]
        q.gaugeFlow(A, ga, 2, 0.01, flow_act=act, **kw)
        q.gaugeFlow(B, gb, 2, 0.01, flow_act=act, **kw)
        assert np.array_equal(ga, gb)
        for loop in (1, 3, 4, 5):
            ea, eb = q.flowEQ(A, loop), q.flowEQ(B, loop)          # of the resident (flowed) fields
            assert np.allclose(ea, eb, rtol=0, atol=0)
    # both parities of a tile position per workgroup (option force_pair, k_force_lds2: neighbours inside the tile position
    # come from LDS) against one tile per workgroup (k_force_lds): the same products in the same order, so bit for bit
    ga, gb, gc = g.copy(), g.copy(), g.copy()
    fa = q.gaugeForce(A, g, cplaq=1.0)
    ma, mb = q.flowMeasure(A, g), q.flowMeasure(B, g)           # clover E, Q + plaquettes, paired kernel (k_flow_obs_clover2)
    q.gaugeFlow(A, ga, 2, 0.01)
    for X in (A, B):
        X.set_option("force_pair", 0)
    assert np.array_equal(fa, q.gaugeForce(A, g, cplaq=1.0)) and np.array_equal(fa, q.gaugeForce(B, g, cplaq=1.0))
    # ... and the one-tile clover kernel: same leaves in the same order, only the workgroup partials are grouped differently
    for m2 in (q.flowMeasure(A, g), q.flowMeasure(B, g)):
        for m1 in (ma, mb):
            assert np.abs(np.asarray(m1[0]) - np.asarray(m2[0])).max() < 1e-15 and np.allclose(m1[1], m2[1], rtol=1e-13, atol=1e-13)
    assert np.abs(np.asarray(ma[0]) - o.plaq(lo, g)).max() < 1e-14 and np.allclose(ma[1], o.flow_EQ(lo, g, 1), rtol=1e-11, atol=1e-12)
    q.gaugeFlow(A, gb, 2, 0.01)
    q.gaugeFlow(B, gc, 2, 0.01)
    for X in (A, B):
        X.set_option("force_pair", 1)
    assert np.array_equal(ga, gb) and np.array_equal(ga, gc)
    ga, gb = g.copy(), g.copy()
    q.gaugeUpdate(A, ga, p, 0.3)
    q.gaugeUpdate(B, gb, p, 0.3)
    assert np.array_equal(ga, gb)
    ga *= 1 + 1e-9
    gb = ga.copy()
    q.reunit(A, ga)
    q.reunit(B, gb)
    assert np.array_equal(ga, gb) and np.array_equal(q.plaq(A, ga), q.plaq(B, gb))
    # Wilson lines: loops that stay within the three ghost slices walk the sharded field directly, the straight
    # Polyakov line in t is assembled from per-rank segments (all-gather); anything else is refused
    for path in ([4] * 8, [-4] * 8, [1] * 8, [1, 4, -1, -4], [4, 4, 4, 1, -4, -4, -4, -1], [-4, -4, 2, 4, 4, -2]):
        assert abs(q.wline(A, path, ga) - q.wline(B, path, gb)) < 1e-15
    assert np.abs(q.s4_gauge(A) - q.s4_gauge(B)).max() < 1e-15 and np.abs(q.s4_gauge(A) - o.s4_gauge(lo, ga)).max() < 1e-14
    pa, pb = q.ploops(A), q.ploops(B)
    for d in range(4):
        assert abs(pa[d] - pb[d]) < 1e-15 and abs(pa[d] - o.wline(lo, ga, [d + 1] * 8)) < 1e-14
    with pytest.raises(q.QexHipError, match="wline"):
        q.wline(B, [4] * 5 + [1] + [-4] * 5 + [-1], gb)


@pytest.mark.gpu
@pytest.mark.parametrize("halo", [False, True])
def test_gpu_resident_md_equals_the_host_field_path(halo):
    """qexhip_md_*: a sequence of MD updates with links, momenta and forces resident on the device -- thin-link gauge force
    (forceA), nHYP-smeared gauge force through the closure (gforce, staghmc_spv.nim:217-228), link update, a force-gradient
    style shift bracketed by save / restore -- against the same sequence through the host-pointer entry points."""
    import qex_amd as q

    lat = [8, 4, 4, 8] if halo else [4, 6, 4, 8]          # sharding needs X*Y*Z/2 to be a multiple of 64
    rf = q.RngField(lat, q.RngMilc6, 31)
    g, p = rf.warm(0.4), rf.randomTAH()
    hc = q.HypCoefs(0.4, 0.5, 0.5)
    # --- host-field path
    ch = q.Context(lat)
    if halo:
        ch.force_halo(True)
    gh, ph = g.copy(), p.copy()
    ph -= 0.1 * q.gaugeForce(ch, gh, cplaq=6.0, adjplaq=-1.5)
    sf = hc.smearGetForce(ch, gh)
    f = np.zeros_like(gh)
    sf.gforce(f, plaq=1.3)
    ph -= 0.2 * f
    q.gaugeUpdate(ch, gh, ph, 0.05)
    gshift = gh.copy()
    q.gaugeUpdate(ch, gshift, q.gaugeForce(ch, gh, cplaq=6.0, adjplaq=-1.5), -0.03)
    fshift = q.gaugeForce(ch, gshift, cplaq=6.0, adjplaq=-1.5)
    ph -= 0.07 * fshift
    # --- resident path
    cd = q.Context(lat)
    if halo:
        cd.force_halo(True)
    md = q.ResidentMD(cd)
    md.begin(g, p)
    assert abs(md.momentum_norm2() - (p * p).sum()) <= 1e-13 * (p * p).sum()
    md.gauge_force(plaq=6.0, adjplaq=-1.5)
    md.kick(md.GAUGE, -0.1)
    sfd = hc.smearGetForce(cd, None)
    sfd.gforce(None, plaq=1.3)
    md.kick(md.NHYP, -0.2)
    md.update_links(0.05)
    md.save_links()
    md.gauge_force(plaq=6.0, adjplaq=-1.5)
    md.shift_links(md.GAUGE, -0.03)
    md.gauge_force(plaq=6.0, adjplaq=-1.5)
    md.kick(md.GAUGE, -0.07)
    md.restore_links()
    gd, pd = np.zeros_like(g), np.zeros_like(p)
    md.end(gd, pd)
    # p + t f is one fma on the device, two roundings in numpy: agreement to rounding, not to the bit
    assert np.abs(pd - ph).max() <= 1e-14 * np.abs(ph).max()
    assert np.abs(gd - gh).max() <= 1e-14
    assert abs(md.momentum_norm2() - (pd * pd).sum()) <= 1e-13 * (pd * pd).sum()
    cx = q.Context(lat)
    with pytest.raises(q.QexHipError):                                    # nothing resident yet
        q.ResidentMD(cx).update_links(0.1)
    with pytest.raises(q.QexHipError):
        hc.smearGetForce(cx, None)
