"""bench.py's CPU baseline (oracle/cpu_simd: QEX's AoSoA V=8 even-odd layout, -O3 -march=native -fopenmp) against the
pinned oracle on the same links and source: same stagD2 to rounding, same CG residual history.  CPU only."""
import numpy as np
import pytest


@pytest.mark.parametrize("lat", [[8, 8, 8, 8], [4, 8, 12, 8]])
def test_cpu_simd_matches_the_oracle(oracle, lat):
    from oracle import cpu_simd as cs

    o = oracle
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 987654321)
    g = o.gauge_random(lo, rf)
    o.rephase(lo, g)
    x, y = o.vector_gaussian(lo, rf), o.vector_gaussian(lo, rf)
    L = cs.Lattice(lat, g)
    for par in (0, 1, 2):
        r1, r2 = y.copy(), y.copy()
        L.stagD2(r1, x, par, 0.4)
        o.stagD2(lo, g, None, r2, x, par, 0.0, 0.4)
        assert np.linalg.norm(r1 - r2) / np.linalg.norm(r2) < 1e-14
    for par_even in (True, False):
        xs, its, hist, _ = L.solveXX(x, 0.1, 1e-12, 2000, par_even, histcap=4096)
        xr, itr, _, histr = o.solveXX(lo, g, None, x, 0.1, 1e-12, 2000, par_even, histcap=4096)
        assert abs(its - itr) <= 1
        n = min(len(hist), len(histr))
        assert np.abs(hist[:100] / histr[:100] - 1).max() < 1e-10 and n > 100
        h = lo.vol // 2
        sl = slice(0, h) if par_even else slice(h, None)
        assert np.linalg.norm(xs[sl] - xr[sl]) / np.linalg.norm(xr[sl]) < 1e-6


def test_cpu_simd_rejects_lattices_its_inner_geometry_cannot_split():
    from oracle import cpu_simd as cs

    with pytest.raises(ValueError):
        cs.Lattice([4, 6, 8, 8], np.zeros((4 * 6 * 8 * 8, 4, 3, 3, 2)))
