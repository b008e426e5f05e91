"""Pins the CPU oracle against the reference's own known-answer vectors.
Runs without a GPU.  Every expected number below is copied from the cited reference test.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_G1_random_gauge_plaquettes(oracle):
    """tests/reprod/trandgauge.nim:4-27: 8^4, g.random (RngMilc6, seed 17^7), sum diff^2 <= 1e-30.
    Pins RngMilc6 + gaussian + projectSU + site order + plaq."""
    o = oracle
    lo = o.Layout([8, 8, 8, 8])
    g = o.gauge_random(lo)  # default seed 17**7, gaugeUtils.nim:1443-1446 / distributionUtils.nim:307
    P = np.array([0.0006005738094166639, 0.0007744149733359666, 0.000491692592364555,
                  -0.0002244585371871249, -0.000700363878755635, -4.121898341926528e-05])
    pl = o.plaq(lo, g)
    assert ((pl - P) ** 2).sum() <= 1e-30


def test_G2_wilson_flow_plaquettes(oracle):
    """src/gauge/wflow.nim:92,99,124-149: same config after gaugeFlow(6, 0.01), rel diff <= 2e-14.
    Pins staples, force sign/normalisation, TAH, exp, RK3."""
    o = oracle
    lo = o.Layout([8, 8, 8, 8])
    g = o.gauge_random(lo)
    o.wflow(lo, g, 6, 0.01)
    p0 = np.array([0.01960725848281519, 0.01982378149813489, 0.01938877647467847,
                   0.0185899778070918, 0.0180821938831715, 0.01876842496122964])
    pl = o.plaq(lo, g)
    assert np.abs(pl - p0).sum() / p0.sum() <= 2e-14


G3_TABLES = {
    # tests/base/twflow_topo.nim:29-62: [E_s, E_t, Q] for loop = 1,3,4,5
    "t0": {1: [0.9278998428166274, 0.9259837153220379, -0.1798124862963527],
           3: [2.099099182199596, 2.096760447628166, -0.5037001984505194],
           4: [3.627286981133991, 3.619461449116142, -0.6989980450588869],
           5: [2.769773748283728, 2.765062802182978, -0.6040403379495964]},
    "fine": {1: [0.6597045206103821, 0.6563289799384344, 0.03799796090979355],
             3: [1.458814621776772, 1.453027521663594, 0.01066005399628158],
             4: [2.109560159995052, 2.104682631200975, 0.08902327151785866],
             5: [1.749675334680512, 1.744341716556223, 0.05315490339585562]},
    "coarse": {1: [0.3330415059918272, 0.3290982076857988, 0.00251688258620615],
               3: [0.7073852695739449, 0.6993729505801789, -0.008560219145048894],
               4: [0.9144248903211708, 0.9058947912540163, 0.02577408143748513],
               5: [0.8013051494825723, 0.7930601360032297, 0.01073616156391243]},
}


def test_G3_wilson_flow_and_topological_charge(oracle):
    """tests/base/twflow_topo.nim:19-62 (CT = 1e-11): MRG32k3a seed 17^13, warm(0.4); E_s, E_t, Q
    from fmunu(loop) for loop 1,3,4,5 at t=0, after gaugeFlow(20, 0.005), and after a further
    gaugeFlow(1, 0.1).  Pins MRG32k3a gaussians, randTah3, warm, exp, the flow and fmunu/E/Q."""
    o = oracle
    lo = o.Layout([8, 8, 8, 8])
    rf = o.RngField(lo, o.RNG_MRG32K3A, 17 ** 13)
    g = o.gauge_warm(lo, 0.4, rf)

    def check(tab):
        for loop, want in G3_TABLES[tab].items():
            got = o.flow_EQ(lo, g, loop)
            assert np.max(np.abs(got / np.array(want) - 1)) < 1e-11, (tab, loop, got)

    check("t0")
    o.wflow(lo, g, 20, 0.005)
    check("fine")
    o.wflow(lo, g, 1, 0.1)
    check("coarse")


def test_wilson_line_plaquette(oracle):
    """tests/base/tgaugeprod.nim:13-19: plaq vs the ordered path product [mu,nu,-mu,-nu]."""
    o = oracle
    lo = o.Layout([8, 8, 8, 8])
    g = o.gauge_random(lo)
    pl = o.plaq(lo, g)
    lines = [[1, 2, -1, -2], [1, 3, -1, -3], [2, 3, -2, -3], [1, 4, -1, -4], [2, 4, -2, -4], [3, 4, -3, -4]]
    wl = np.array([o.wline(lo, g, p).real / 6.0 for p in lines])
    assert np.max(np.abs(pl - wl)) < 1e-15


def test_G4_mrg32k3a(oracle):
    """tests/base/tmrg32k3a.nim:9-27 (CT = 1e-13 relative)."""
    o = oracle
    seed = 17 ** 13
    res = [[0.3268000301845387, 0.1909631348029552, 0.3976696014207036],
           [0.2491408676889959, 0.8109031896264907, 0.4171423534316965]]
    for k in range(2):
        u = o.mrg32k3a_uniforms(seed, k, 3)
        assert np.max(np.abs(u / np.array(res[k]) - 1)) < 1e-13
    import ctypes as C

    lo = o.Layout([8, 8, 8, 16])
    rf = o.RngField(lo, o.RNG_MRG32K3A, seed)
    v = np.zeros((lo.vol, 24))
    L = o.lib()
    L.qo_field_uniform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.qo_field_uniform(lo._h, rf._h, 24, o._p(v), 0)
    assert abs((v * v).sum() / 65517.83893610391 - 1) < 1e-13  # DiracFermion uniform norm2


def test_G5_rngmilc6_seeding(oracle):
    """tests/base/trngseed.nim:16,53-63: randomTAH norm2 and a global uniform (CT 1e-13)."""
    o = oracle
    seed = 7_005_003_002_001_000_000  # narrowed to uint32 by seedIndep (milcrng.nim:111-112)
    lo = o.Layout([8, 8, 8, 8])
    rf = o.RngField(lo, o.RNG_MILC6, seed)
    p = o.gauge_random_tah(lo, rf)
    assert abs((p * p).sum() / 131563.7475902051 - 1) < 1e-13
    u, _ = o.milc6_stream(seed, 987654321, 1)
    assert float(u[0]) == 0.7708062529563904


def test_G6_unit_gauge_and_single_link(oracle):
    """tests/base/tstressplaq.nim:29-66: unit gauge -> 1/6 each; one perturbed link changes
    exactly the plaquettes that contain it."""
    o = oracle
    lo = o.Layout([4, 4, 4, 4])
    g = o.gauge_unit(lo)
    assert np.max(np.abs(o.plaq(lo, g) - 1.0 / 6.0)) < 1e-15
    # multiply U_0(origin) by a phase e^{i a}: the 6 plaquettes through that link get cos(a)
    a = 0.3
    i0 = lo.index([0, 0, 0, 0])
    m = g[i0, 0, :, :, 0] + 1j * g[i0, 0, :, :, 1]
    m = m * np.exp(1j * a)
    g[i0, 0, :, :, 0], g[i0, 0, :, :, 1] = m.real, m.imag
    pl = o.plaq(lo, g)
    V = lo.vol
    # planes containing direction 0: (1,0)->ip 0, (2,0)->ip 1, (3,0)->ip 3 ; two plaquettes each
    expect = np.full(6, 1.0 / 6.0)
    for ip in (0, 1, 3):
        expect[ip] += 2 * (np.cos(a) - 1.0) * 3 / (V * 18.0)
    assert np.max(np.abs(pl - expect)) < 1e-15


def test_su3_helpers(oracle):
    o = oracle
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 3, 2))
    u = o.su3_fn("qo_projectSU", x)
    m = u[..., 0] + 1j * u[..., 1]
    assert np.abs(m @ m.conj().T - np.eye(3)).max() < 1e-13
    assert abs(np.linalg.det(m) - 1) < 1e-13
    t = o.su3_fn("qo_projectTAH", x)
    mt = t[..., 0] + 1j * t[..., 1]
    assert np.abs(mt + mt.conj().T).max() < 1e-15 and abs(np.trace(mt)) < 1e-15
    e = o.su3_fn("qo_exp", 0.37 * t)
    me = e[..., 0] + 1j * e[..., 1]
    w, v = np.linalg.eig(0.37 * mt)
    ref = v @ np.diag(np.exp(w)) @ np.linalg.inv(v)
    assert np.abs(me - ref).max() < 1e-12


def test_layout_matches_host_layout(oracle):
    """The oracle's site order, the product's host-side Layout and the closed form agree."""
    import qex_amd as q

    lat = [4, 6, 2, 8]
    lo_o, lo_q = oracle.Layout(lat), q.Layout(lat)
    for i in range(0, lo_q.vol, 7):
        x = lo_o.coord(i)
        assert x == lo_q.coord(i)
        lex = x[0] + lat[0] * (x[1] + lat[1] * (x[2] + lat[2] * x[3]))
        assert i == lex // 2 + (sum(x) & 1) * lo_q.vol // 2
        assert lo_q.index(x) == i and lo_o.index(x) == i
        for mu in range(4):
            y = list(x)
            y[mu] = (y[mu] + 1) % lat[mu]
            assert lo_o.neighbor(i, mu, 1) == lo_q.index(y)


def test_operator_identities(oracle):
    """Build-owned invariants closing the gap "no asserted KAT for a Dslash output" (SURVEY 8c):
    anti-Hermiticity of D, A_ee = 4 D^+D on the even subset, gauge covariance, plane waves."""
    o = oracle
    lat = [4, 4, 6, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 987654321)
    g = o.gauge_random(lo, rf)
    o.rephase(lo, g)
    x, y = o.vector_gaussian(lo, rf), o.vector_gaussian(lo, rf)
    cx = lambda a: a[..., 0] + 1j * a[..., 1]
    dot = lambda a, b: np.vdot(cx(a), cx(b))
    # <y, D x> = -<D y, x> at m = 0
    assert abs(dot(y, o.D(lo, g, None, x, 0.0)) + dot(o.D(lo, g, None, y, 0.0), x)) < 1e-10
    # stagD2ee = 4 (m^2 - D_eo D_oe) = 4 D^+ D restricted to even
    h = lo.vol // 2
    xe = x.copy()
    xe[h:] = 0
    m = 0.13
    A = o.stagD2xx(lo, g, None, xe, m * m, True)
    DdD = o.Ddag(lo, g, None, o.D(lo, g, None, xe, m), m)
    assert np.abs(A[:h] - 4 * DdD[:h]).max() < 1e-12
    # eoReduce (stagD.nim:575-581) = the even half of Ddag, odd half of r untouched; eoReconstruct undoes the
    # even-odd elimination: from r.even = (D^+ D)^-1_ee-solution the odd half follows from the odd row of D r = b
    r = y.copy()
    o.eoReduce(lo, g, None, r, x, m)
    assert np.array_equal(r[h:], y[h:]) and np.abs(r[:h] - o.Ddag(lo, g, None, x, m)[:h]).max() < 1e-14
    full = o.D(lo, g, None, x, m)                   # b = D x  =>  eoReconstruct(x.even, b) returns x.odd
    r = x.copy()
    r[h:] = 0
    o.eoReconstruct(lo, g, None, r, full, m)
    assert np.abs(r - x).max() < 1e-12
    # gauge covariance: U'_mu(s) = G(s) U_mu(s) G(s+mu)^+, x' = G x  =>  D'x' = G (D x)
    G = np.zeros((lo.vol, 3, 3, 2))
    for i in range(lo.vol):
        G[i] = o.su3_fn("qo_projectSU", np.random.default_rng(i).standard_normal((3, 3, 2)))
    Gc, gc = cx(G), cx(g)
    g2 = np.zeros_like(g)
    for i in range(lo.vol):
        for mu in range(4):
            j = lo.neighbor(i, mu, 1)
            mm = Gc[i] @ gc[i, mu] @ Gc[j].conj().T
            g2[i, mu, :, :, 0], g2[i, mu, :, :, 1] = mm.real, mm.imag
    xg = np.einsum("nij,nj->ni", Gc, cx(x))
    x2 = np.stack([xg.real, xg.imag], axis=-1)
    lhs = cx(o.D(lo, g2, None, np.ascontiguousarray(x2), 0.2))
    rhs = np.einsum("nij,nj->ni", Gc, cx(o.D(lo, g, None, x, 0.2)))
    assert np.abs(lhs - rhs).max() < 1e-12
    # free field (unit links + staggered phases, periodic): plane wave is an eigenvector of D^2
    gu = o.gauge_unit(lo)
    o.stagPhase(lo, gu)
    k = [1, 0, 2, 1]
    pw = np.zeros((lo.vol, 3, 2))
    for i in range(lo.vol):
        c = lo.coord(i)
        ph = 2 * np.pi * sum(k[d] * c[d] / lat[d] for d in range(4))
        pw[i, 0] = [np.cos(ph), np.sin(ph)]
    DDpw = o.D(lo, gu, None, o.D(lo, gu, None, pw, 0.0), 0.0)
    lam = -sum(np.sin(2 * np.pi * k[d] / lat[d]) ** 2 for d in range(4))
    assert np.abs(DDpw - lam * pw).max() < 1e-12


def test_solver_fixture(oracle):
    """Committed fixture (tests/golden/cg_8x8x8x8.json, written by tests/golden/make_fixtures.py
    from the oracle): first residuals of the 8^4 m=0.1 CG for a gaussian and a point source
    (src/physics/stagSolve.nim:542,576-583).  Guards the oracle against silent edits."""
    fx = json.load(open(os.path.join(HERE, "golden", "cg_8x8x8x8.json")))
    o = oracle
    lo = o.Layout(fx["lat"])
    rf = o.RngField(lo, o.RNG_MILC6, fx["seed"])
    g = o.gauge_random(lo, rf)
    assert np.max(np.abs(o.plaq(lo, g) - np.array(fx["plaq"]))) < 1e-15
    o.rephase(lo, g)
    b = o.vector_gaussian(lo, rf)
    _, its, _, hist = o.solveXX(lo, g, None, b, fx["mass"], fx["r2req"], 1000, True, histcap=64)
    n = len(fx["hist_gaussian"])
    assert np.max(np.abs(hist[:n] / np.array(fx["hist_gaussian"]) - 1)) < 1e-11
    assert abs(its - fx["its_gaussian"]) <= 1
    p = np.zeros_like(b)
    p[0, 0, 0] = 1.0
    x, its, fin = o.solve(lo, g, None, p, fx["mass"], fx["r2req"], 10000)
    assert abs(its - fx["its_point"]) <= 2
    assert abs((x * x).sum() / fx["x2_point"] - 1) < 1e-9
    g3 = o.gauge_random(lo, rf)
    o.rephase(lo, g3)
    g3 *= 0.3
    r = np.zeros_like(b)
    o.stagD2(lo, g, g3, r, b, 2, 0.0, 0.0)
    assert abs((r * r).sum() / fx["naik_D2_norm2"] - 1) < 1e-13
    xs, its, fin = o.solve_multi(lo, g, None, b, fx["masses"], fx["r2req"], 10000)
    for k, v in enumerate(fx["multi_x2"]):
        assert abs((xs[k] * xs[k]).sum() / v - 1) < 1e-8


def _tmatfun_bound(u):
    """tests/base/tmatfun.nim:11-31 (chkzero / chkeq) for t = r y r against 1, y = x^+ x, r = rsqrtPH(y): with
    W = x r (projectU), t = W^+ W.  The reference accepts max|t - 1| / max|1| / (rows * cols) < 384 * rows * eps."""
    m = u[..., 0] + 1j * u[..., 1]
    t = np.einsum("nki,nkj->nij", m.conj(), m)
    md = np.abs(np.stack([(t - np.eye(3)).real, (t - np.eye(3)).imag])).max(axis=(0, 2, 3))
    return md / 9.0, 384 * 3 * np.finfo(np.float64).eps


def test_rsqrtPH_property_of_the_reference(oracle):
    """`suite "Test matrix rsqrtPH"` (tests/base/tmatfun.nim:33-77) on complex 3x3 gaussian matrices: r y r = 1 to the
    reference's own bound, through the restated projectU (= x (x^+ x)^(-1/2), matrixFunctions.nim:301-313)."""
    o = oracle
    x = np.random.default_rng(13).standard_normal((2000, 3, 3, 2))
    u = np.stack([o.su3_fn("qo_projectU", xi) for xi in x])
    s, bound = _tmatfun_bound(u)
    assert s.max() < bound


def test_s4_gauge_sums_to_the_plaquettes(oracle):
    """s4_gauge (stagg_pv_hmc/staghmc_spv_meas.nim:25-65) adds every site plaquette to the even/odd bin of its two
    directions and normalises by physVol * 0.5 * (nd - 1) * nc; g.plaq (pinned by G1) normalises by physVol * 6 * nc:
    the eight numbers add up to 8 x the sum of the six plaquettes, and on the unit gauge every bin is 1."""
    o = oracle
    lo = o.Layout([8, 8, 8, 8])
    g = o.gauge_random(lo)
    s4 = o.s4_gauge(lo, g)
    assert abs(s4.sum() - 8.0 * o.plaq(lo, g).sum()) < 1e-13
    assert np.abs(o.s4_gauge(lo, o.gauge_unit(lo)) - 1.0).max() < 1e-15
    # a field that depends on the parity of x_0 only through U_1 shows up in direction 0's even / odd split
    lo2 = o.Layout([4, 6, 4, 2])
    g2 = o.gauge_warm(lo2, 0.3, o.RngField(lo2, o.RNG_MILC6, 5))
    s2 = o.s4_gauge(lo2, g2)
    assert s2.shape == (4, 2) and abs(s2.sum() - 8.0 * o.plaq(lo2, g2).sum()) < 1e-13
