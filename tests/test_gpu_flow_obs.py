"""GPU parity of the flow observables: fmunu(loop) -> E_s, E_t, Q through the
HIP kernels, against the reference's golden set G3 (tests/base/twflow_topo.nim:19-62) and the oracle.
"""
import os

import numpy as np
import pytest

from test_oracle_golden import G3_TABLES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flow_exp", [1, 0])
def test_G3_on_gpu(oracle, flow_exp):
    """The whole twflow_topo.nim sequence with the flow AND the observables on the GPU (flow_exp: the flow's closed-form
    exp, the default, and the reference's Taylor + squarings algorithm)."""
    import qex_amd as q

    o = oracle
    lo = o.Layout([8, 8, 8, 8])
    rf = o.RngField(lo, o.RNG_MRG32K3A, 17 ** 13)
    g = o.gauge_warm(lo, 0.4, rf)
    ctx = q.Context([8, 8, 8, 8])
    ctx.set_option("flow_exp", flow_exp)

    def check(tab):
        for loop, want in G3_TABLES[tab].items():
            got = q.flowEQ(ctx, loop)
            assert np.max(np.abs(got / np.array(want) - 1)) < 1e-11, (tab, loop, got)   # CT of the reference test

    q.plaq(ctx, g)                       # uploads g
    check("t0")
    q.gaugeFlow(ctx, g, 20, 0.005)
    check("fine")
    q.gaugeFlow(ctx, g, 1, 0.1)
    check("coarse")


def test_flow_obs_vs_oracle_other_lattice(oracle):
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 987654321)
    g = o.gauge_warm(lo, 0.3, rf)
    ctx = q.Context(lat)
    for loop in (1, 3, 4, 5):
        got = q.flowEQ(ctx, loop, g)
        want = o.flow_EQ(lo, g, loop)
        assert np.max(np.abs(got - want)) < 1e-12 * max(1.0, np.abs(want).max()), (loop, got, want)
    with pytest.raises(q.QexHipError):
        q.flowEQ(ctx, 2)                 # fmunu uses loop in [1,3,4,5]


@pytest.mark.parametrize("lat,halo", [([8, 8, 8, 8], False), ([4, 6, 8, 4], False), ([12, 6, 6, 10], False),
                                     ([8, 8, 8, 8], True), ([8, 8, 4, 6], True)])
def test_clover_kernel_against_the_path_walker_and_the_oracle(oracle, lat, halo):
    """fmunu(loop = 1) has its own kernel (workgroup = tile x six planes, shared links through LDS): it must give what the
    generic path walker (option obs_clover = 0) gives -- same leaves, same order -- on lattices with a ragged last tile and
    through the ghost zones of a t-sharded rank, and both must match the oracle."""
    import qex_amd as q

    o = oracle
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 987654321)
    g = o.gauge_warm(lo, 0.3, rf)
    want = o.flow_EQ(lo, g, 1)
    ctx = q.Context(lat)
    if halo:
        ctx.force_halo(True)
    got = {}
    for v in (1, 0):
        ctx.set_option("obs_clover", v)
        got[v] = q.flowEQ(ctx, 1, g)
        assert np.max(np.abs(got[v] - want)) < 1e-12 * max(1.0, np.abs(want).max()), (v, got[v], want)
    assert np.max(np.abs(got[1] - got[0])) < 1e-14 * max(1.0, np.abs(want).max()), got


@pytest.mark.gpu
@pytest.mark.parametrize("halo", [False, True])
def test_flow_measure_is_plaq_and_EQ_in_one_pass(oracle, halo):
    """qexhip_flow_measure: the plaquettes (gaugeUtils.nim:213-282) and the clover E_s, E_t, Q (gauge_flow.nim:360-379) a flow
    loop prints after every step, from ONE kernel -- the plaquette of a plane is the trace of one clover leaf.  Against the
    separate entry points, the oracle, and the reference's own plaquette vector G1 (tests/reprod/trandgauge.nim:17, its bound
    1e-30 on the summed squared differences)."""
    import qex_amd as q

    lat = [8, 8, 8, 8]
    lo = oracle.Layout(lat)
    g = oracle.gauge_random(lo)                    # G1: RngMilc6, seed 17^7
    ctx = q.Context(lat)
    if halo:
        ctx.force_halo(True)
    pl, eq = q.flowMeasure(ctx, g)
    g1 = np.array([0.0006005738094166639, 0.0007744149733359666, 0.000491692592364555,
                   -0.0002244585371871249, -0.000700363878755635, -4.121898341926528e-05])
    assert ((pl - g1) ** 2).sum() <= 1e-30
    assert np.abs(pl - q.plaq(ctx)).max() < 1e-17 and np.abs(pl - oracle.plaq(lo, g)).max() < 1e-17
    assert np.array_equal(eq, q.flowEQ(ctx, 1))    # the same kernel, the same sums
    assert np.allclose(eq, oracle.flow_EQ(lo, g, 1), rtol=1e-12, atol=1e-12)
    # after a few flow steps (smooth field: plaquettes of order 0.02) and on a lattice with a ragged last tile
    q.gaugeFlowResident(ctx, 3, 0.01)
    pl, eq = q.flowMeasure(ctx)
    assert np.abs(pl - q.plaq(ctx)).max() < 1e-16 and np.array_equal(eq, q.flowEQ(ctx, 1))
    if not halo:
        lat2 = [4, 6, 10, 6]
        lo2 = oracle.Layout(lat2)
        g2 = oracle.gauge_random(lo2, seed=5)
        c2 = q.Context(lat2)
        pl, eq = q.flowMeasure(c2, g2)
        assert np.abs(pl - oracle.plaq(lo2, g2)).max() < 1e-16
        assert np.allclose(eq, oracle.flow_EQ(lo2, g2, 1), rtol=1e-12, atol=1e-12)
        c2.set_option("obs_clover", 0)             # without the dedicated kernel: two passes, same numbers
        pl2, eq2 = q.flowMeasure(c2)
        assert np.abs(pl2 - pl).max() < 1e-16 and np.allclose(eq2, eq, rtol=1e-12, atol=1e-13)


@pytest.mark.gpu
def test_flow_loop_example_prints_the_reference_line(oracle):
    """examples/gauge_flow.py = the loop of src/flow/gauge_flow.nim: its FLOW lines (tau, plaquette normalised to 3, clover E,
    t^2 E, d t^2E / dt, the `check` column 12 t^2 (3 - plaq), Q, t^2 E_ss, t^2 E_st, Polyakov loops; gauge_flow.nim:385-470)
    recomputed from the oracle's flow, plaquettes, clover observables and Wilson lines."""
    import re
    import subprocess
    import sys

    o = oracle
    lat = [4, 6, 4, 8]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "examples", "gauge_flow.py"), "-lat"] + [str(v) for v in lat] +
                       ["-warm", "0.3", "-dt", "0.02", "0.05", "-tmax", "0.06", "0.16", "-seed", "77"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = [[float(v) for v in ln.split()[1:]] for ln in p.stdout.splitlines() if ln.startswith("FLOW ")]
    assert len(rows) == 1 + 3 + 2
    lo = o.Layout(lat)
    g = o.gauge_warm(lo, 0.3, o.RngField(lo, o.RNG_MILC6, 77))
    o.gauge_projectSU(lo, g)
    tau, old = 0.0, 0.0
    steps = [None, 0.02, 0.02, 0.02, 0.05, 0.05]
    for row, dt in zip(rows, steps):
        if dt is not None:
            o.wflow(lo, g, 1, dt)
            tau += dt
        pl = o.plaq(lo, g)
        es, et, qq = o.flow_EQ(lo, g, 1)
        ss, st = 2.0 * pl[:3].sum(), 2.0 * pl[3:].sum()
        p3 = (3.0 * ss + 3.0 * st) / 2.0
        t2E = tau * tau * (es + et)
        loops = [o.wline(lo, g, [d + 1] * lat[d]) for d in range(4)]
        pls, plt = sum(loops[:3]) / 3.0, loops[3]
        ref = [tau, p3, es + et, t2E, (t2E - old) / (dt or 0.02), 12.0 * tau * tau * (3.0 - p3), qq, tau * tau * es, tau * tau * et,
               3.0 * plt.real, 3.0 * plt.imag, 3.0 * pls.real, 3.0 * pls.imag]
        old = t2E
        assert abs(row[0] - ref[0]) < 0.006                              # tau is printed with two decimals
        for a, b in zip(row[1:], ref[1:]):
            assert abs(a - b) < 2e-12 * max(1.0, abs(b)) + 6e-14, (tau, row, ref)   # 13 printed decimals
