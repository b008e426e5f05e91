"""The product's own random number fields and configuration generators (csrc/rng.hip):
held to the same golden sets as the oracle (G1, G4, G5) and, deviate for deviate, to the oracle itself.
Host code only: runs without a GPU."""
import numpy as np


def test_streams_equal_the_oracles_bit_for_bit(oracle):
    import qex_amd as q

    o = oracle
    lat = [4, 6, 2, 8]
    lo = o.Layout(lat)
    for kind, okind in ((q.RngMilc6, o.RNG_MILC6), (q.MRG32k3a, o.RNG_MRG32K3A)):
        r, ro = q.RngField(lat, kind, 987654321), o.RngField(lo, okind, 987654321)
        assert np.array_equal(r.gaussian_vector(), o.vector_gaussian(lo, ro))
        assert np.array_equal(r.randomTAH(), o.gauge_random_tah(lo, ro))
        assert np.array_equal(r.u1_vector(), o.vector_u1(lo, ro))
        assert np.array_equal(r.gaussian_vector(), o.vector_gaussian(lo, ro))        # streams stay in step
        g, go = r.random(), o.gauge_random(lo, ro)
        assert np.abs(g - go).max() < 1e-9                   # projectSU of an ill-conditioned gaussian matrix amplifies rounding
        assert np.median(np.abs(g - go)) < 1e-15
        w, wo = r.warm(0.4), o.gauge_warm(lo, 0.4, ro)
        assert np.abs(w - wo).max() < 1e-14
        assert np.array_equal(r.gaussian_vector(), o.vector_gaussian(lo, ro))


def test_golden_sets_G1_G4_G5_through_the_product_generators(oracle):
    import qex_amd as q

    o = oracle
    # G1 (tests/reprod/trandgauge.nim:4-27): g.random with RngMilc6 seed 17^7, six plaquettes, sum diff^2 <= 1e-30
    lo = o.Layout([8, 8, 8, 8])
    g = q.RngField([8, 8, 8, 8], q.RngMilc6, 17 ** 7).random()
    gold = np.array([0.0006005738094166639, 0.0007744149733359666, 0.000491692592364555,
                     -0.0002244585371871249, -0.000700363878755635, -4.121898341926528e-05])
    assert ((o.plaq(lo, g) - gold) ** 2).sum() <= 1e-30
    # G4 (tests/base/tmrg32k3a.nim:22-27): DiracFermion-shaped uniform field, norm2
    v = q.RngField([8, 8, 8, 16], q.MRG32k3a, 17 ** 13).uniform(24)
    assert abs((v * v).sum() / 65517.83893610391 - 1) < 1e-13
    # G5 (tests/base/trngseed.nim:53-56): randomTAH norm2, seed narrowed to uint32
    p = q.RngField([8, 8, 8, 8], q.RngMilc6, 7_005_003_002_001_000_000).randomTAH()
    assert abs((p * p).sum() / 131563.7475902051 - 1) < 1e-13


def test_sharded_fields_are_slices_of_the_global_one():
    """newRNGField seeds by GLOBAL lexicographic index: a rank's slab must draw what the same sites draw in a
    one-rank run (reproducibility across rank counts, src/rng/distributionUtils.nim:306-331)."""
    import qex_amd as q

    glat = [4, 4, 2, 8]
    full = q.RngField(glat, q.RngMilc6, 4242).gaussian_vector()
    lo = q.Layout(glat)
    for rank in range(2):
        lat = glat[:3] + [4]
        part = q.RngField(lat, q.RngMilc6, 4242, glat=glat, t_offset=4 * rank).gaussian_vector()
        ll = q.Layout(lat)
        for i in range(0, ll.vol, 7):
            x = ll.coord(i)
            xg = x[:3] + [x[3] + 4 * rank]
            assert np.array_equal(part[i], full[lo.index(xg)])
