"""The product's own random number fields and configuration generators (csrc/rng.hip):
held to the same golden sets as the oracle (G1, G4, G5) and, deviate for deviate, to the oracle itself.
Host code only: runs without a GPU."""
import numpy as np
import pytest


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


@pytest.mark.gpu
@pytest.mark.parametrize("lat", [[8, 8, 8, 8], [16, 16, 16, 16], [4, 6, 10, 6]])
def test_device_side_generation_against_the_host_generators(lat):
    """qexhip_rng_dev_* / qexhip_md_refresh_momenta: one lane per site advances ITS RngMilc6 stream on the GPU.  The generator
    states afterwards are bit for bit the host's (integer arithmetic); the deviates go through the device's log / cos / sqrt, so
    after RngMilc6.gaussian's float32 rounding they equal the host's except where a double lands on a float32 rounding
    boundary (expected ~1e-8 of the deviates, each off by one float32 ulp: counted and bounded here); u1 phases agree to 1e-15."""
    import qex_amd as q

    ctx = q.Context(lat)
    dev, host = q.RngField(lat, q.RngMilc6, 1234567), q.RngField(lat, q.RngMilc6, 1234567)
    vol = int(np.prod(lat))
    q.gaugeSet(ctx, q.unit(q.Layout(lat)))
    md = q.ResidentMD(ctx)
    mism, total = 0, 0
    for rep in range(2):
        dev.dev_momenta(ctx)
        md.begin(None, None)
        p = np.zeros((vol, 4, 3, 3, 2))
        md.end(None, p)
        ph = host.randomTAH()
        bad = p != ph
        mism += int(bad.sum()); total += p.size
        assert np.abs(p - ph).max() <= 1.3e-7 * max(1.0, np.abs(ph).max())        # one float32 ulp at most
        assert abs(md.momentum_norm2() - (ph * ph).sum()) < 1e-9 * (ph * ph).sum()
        fid = ctx.field_new()
        dev.dev_gaussian_vector(ctx, fid)
        v, vh = ctx.field_download(fid), host.gaussian_vector()
        mism += int((v != vh).sum()); total += v.size
        assert np.abs(v - vh).max() <= 1.3e-7 * max(1.0, np.abs(vh).max())
        dev.dev_u1_vector(ctx, fid)
        u, uh = ctx.field_download(fid), host.u1_vector()
        assert np.abs(u - uh).max() < 1e-15
        ctx.field_free(fid)
        assert np.array_equal(dev.state(), host.state())
    # host and device draws can be mixed: the stream just continues
    assert np.array_equal(dev.gaussian_vector(), host.gaussian_vector())
    print("lattice %s: %d of %d float32-rounded deviates differ from the host's" % (lat, mism, total))
    assert mism <= max(2, int(2e-6 * total))
    mr = q.RngField(lat, q.MRG32k3a, 7)
    with pytest.raises(q.QexHipError, match="RngMilc6"):
        mr.dev_gaussian_vector(ctx, ctx.field_new())
