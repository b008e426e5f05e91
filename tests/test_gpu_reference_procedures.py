"""Two of the reference's own procedure tests, replayed through the HIP kernels with exact (integer-valued) data:

tests/base/tshift.nim   a field whose value at a site encodes the site's coordinates is shifted by d in direction mu and must
                        equal the field encoded with offset d ("f^n = fn", "b f = 1", "f^L = 1").  The product has no free
                        shift operator; its shifts live inside the Dslash.  With unit links (no phases) stagD2 IS a sum of
                        shifts: r(s) = sum_mu [x(s + h mu) - x(s - h mu)], h = 1 (and h = 3 with long links) -- every hop of
                        every direction, its sense and its wrap, checked EXACTLY (small integers in doubles), on the
                        periodic kernels, with ghost zones, and with the exchange overlapped.
tests/base/treduce.nim  v := i; sqrt(norm2 v) == i * sqrt(3 V) to 1e-10, for i = 1..n: the reductions behind the CG scalars
                        (fieldET.nim:605-625 norm2P).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def coded_field(lo, lat, offset=(0, 0, 0, 0)):
    """tshift.nim `set`: t = 1 + 10 t + ((offset_i + lat_i + x_i) mod lat_i), i = 0..3, in colour 0 (others: fixed small integers)"""
    x = lo.coords
    t = np.zeros(lo.vol)
    for i in range(4):
        t = 1 + 10 * t + ((offset[i] + lat[i] + x[:, i]) % lat[i])
    v = np.zeros((lo.vol, 3, 2))
    v[:, 0, 0] = t
    v[:, 1, 1] = 1.0 + (x[:, 0] % 3)
    v[:, 2, 0] = -2.0
    return v, t


@pytest.mark.parametrize("lat", [[8, 8, 8, 8], [4, 6, 10, 6], [16, 4, 2, 6], [12, 4, 8, 4]])
@pytest.mark.parametrize("mode", ["periodic", "halo", "halo+overlap"])
def test_tshift_through_the_dslash(lat, mode):
    import qex_amd as q

    lo = q.Layout(lat)
    if mode != "periodic" and (lat[0] // 2 * lat[1] * lat[2]) % 64:
        pytest.skip("t-sharding needs whole tiles per slice")
    ctx = q.Context(lat)
    if mode != "periodic":
        ctx.force_halo(True)
        ctx.set_option("overlap", 1 if mode == "halo+overlap" else 0)
    unit = np.zeros((lo.vol, 4, 3, 3, 2))
    for a in range(3):
        unit[:, :, a, a, 0] = 1.0
    x, t = coded_field(lo, lat)
    naik_ok = min(lat) >= 4
    for naik in ([False, True] if naik_ok else [False]):
        s = q.newStag3(ctx, unit, 2.0 * unit) if naik else q.newStag(ctx, unit)    # long links = 2: the 3-hop terms stay distinguishable
        r = np.zeros_like(x)
        s.stagD2(r, x, "all", 0.0, 0.0)
        want = np.zeros(lo.vol)
        for mu in range(4):
            for h, w in ((1, 1.0),) + (((3, 2.0),) if naik else ()):
                off = [0, 0, 0, 0]
                off[mu] = h
                _, fwd = coded_field(lo, lat, off)            # "f ^* x": y(s) = x(s + h mu) = the field encoded with offset +h
                off[mu] = -h
                _, bwd = coded_field(lo, lat, off)
                want += w * (fwd - bwd)
        assert np.array_equal(r[:, 0, 0], want), (lat, mode, naik, np.abs(r[:, 0, 0] - want).max())
        assert not r[:, 0, 1].any() and not r[:, 2, 0].any()     # colour 2 is constant: forward and backward hops cancel exactly
        # "b f = 1" / "f^L = 1": the antisymmetric combination annihilates a field that is constant along every direction
        c = np.zeros_like(x)
        c[:, 1, 0] = 7.0
        s.stagD2(r, c, "all", 0.0, 0.0)
        assert not r.any()
    ctx.close()


def test_treduce_norm2_of_constant_fields():
    import qex_amd as q

    lat = [16, 16, 16, 16]
    lo = q.Layout(lat)
    ctx = q.Context(lat)
    v1x = np.sqrt(3.0 * lo.vol)
    v = np.zeros((lo.vol, 3, 2))
    fid = ctx.field_new(v)
    for i in list(range(1, 40)) + [100, 999, 1000]:
        v[:, :, 0] = float(i)
        ctx.field_upload(fid, v)
        got = np.sqrt(ctx.dev_norm2(fid))
        assert abs(got - i * v1x) < 1e-10, (i, got, i * v1x)            # treduce.nim's bound
        assert abs(np.sqrt(ctx.dev_norm2(fid, "even")) - i * v1x / np.sqrt(2.0)) < 1e-10
    ctx.close()
