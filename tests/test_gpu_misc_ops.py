"""Remaining pieces of SURVEY.md 8 rows a3 / a9: Staggered.peqDdag and sp.usePrevSoln."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())


@pytest.fixture(scope="module")
def S(oracle):
    import qex_amd as q

    class X:
        pass

    lat = [8, 4, 6, 4]
    X.o, X.q, X.lo = oracle, q, oracle.Layout(lat)
    rf = oracle.RngField(X.lo, oracle.RNG_MILC6, 987654321)
    X.g = oracle.gauge_random(X.lo, rf)
    oracle.rephase(X.lo, X.g)
    X.x, X.y = oracle.vector_gaussian(X.lo, rf), oracle.vector_gaussian(X.lo, rf)
    X.ctx = q.Context(lat)
    X.s = q.newStag(X.ctx, X.g)
    return X


def test_peqDdag(S):
    """r += Ddag x  (stagD.nim:572-574: stagD(..., m, -1, 1))"""
    r = S.y.copy()
    S.s.peqDdag(r, S.x, 0.3)
    ref = S.y + S.o.Ddag(S.lo, S.g, None, S.x, 0.3)
    assert relerr(r, ref) < 1e-13
    ro = S.y.copy()
    S.o.stagD(S.lo, S.g, None, ro, S.x, 0, 0.3, -1.0, 1.0)
    S.o.stagD(S.lo, S.g, None, ro, S.x, 1, 0.3, -1.0, 1.0)
    assert relerr(r, ro) < 1e-13


def test_usePrevSoln(S):
    """A second solve started from a slightly perturbed solution needs far fewer iterations and
    reproduces the oracle's usePrevSoln path (stagSolve.nim:234-243)."""
    q = S.q
    sp = q.SolverParams(r2req=1e-12, maxits=10000, verbosity=0)
    x = np.zeros_like(S.x)
    S.s.solve(x, S.x, 0.1, sp)
    cold = sp.iterations
    x0 = x + 1e-4 * S.y
    sp2 = q.SolverParams(r2req=1e-12, maxits=10000, verbosity=0, usePrevSoln=True)
    x1 = x0.copy()
    S.s.solve(x1, S.x, 0.1, sp2)
    xr, its, fin = S.o.solve_prev(S.lo, S.g, None, x0, S.x, 0.1, 1e-12, 10000)
    assert sp2.iterations < cold and abs(sp2.iterations - its) <= 2
    assert relerr(x1, xr) < 1e-7
    r = S.o.D(S.lo, S.g, None, x1, 0.1) - S.x
    assert (r * r).sum() / (S.x * S.x).sum() <= 1e-12
