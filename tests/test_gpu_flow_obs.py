"""GPU parity of the flow observables: fmunu(loop) -> E_s, E_t, Q through the
HIP kernels, against the reference's golden set G3 (tests/base/twflow_topo.nim:19-62) and the oracle.
"""
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
