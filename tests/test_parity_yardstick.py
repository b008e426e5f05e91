"""The yardstick of the history-parity tests (tests/parity_log.py, oracle/qex_oracle_ext.inc), on the CPU.

CG's residual history is compared with the oracle's to 1e-10 over the first 100 iterations and to 1e-6 over the whole history
on the BASELINE configs.  On harder systems two fp64 runs of the same algorithm differ by per cent along the tail, and the bound
there is set by the SAME algorithm run in IEEE binary128: how far the fp64 reference algorithm strays from that trajectory, at
several thread counts, is how far an fp64 implementation may.  This file pins what that construction rests on."""
import numpy as np
import pytest


def setup(o, lat, seed=987654321):
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, seed)
    g = o.gauge_random(lo, rf)
    o.rephase(lo, g)
    return lo, g, o.vector_gaussian(lo, rf)


@pytest.mark.parametrize("lat", [[8, 8, 8, 8], [4, 6, 10, 6]])
def test_binary128_twin_is_the_same_algorithm_and_reproducible(oracle, lat):
    o = oracle
    lo, g, x = setup(o, lat)
    nt = o.num_threads()
    _, its, _, h = o.solveXX(lo, g, None, x, 0.1, 1e-12, 2000, True, histcap=4096)
    it_t, ht = o.solveXX_ext(lo, g, None, x, 0.1, 1e-12, 2000, True, histcap=4096)
    assert abs(its - it_t) <= 2
    n = min(len(h), len(ht))
    assert np.abs(h[:60] / ht[:60] - 1).max() < 1e-12          # the same iteration, before rounding is amplified
    try:
        o.lib().qo_set_num_threads(2 if nt != 2 else 3)
        _, ht2 = o.solveXX_ext(lo, g, None, x, 0.1, 1e-12, 2000, True, histcap=4096)
        _, _, _, h2 = o.solveXX(lo, g, None, x, 0.1, 1e-12, 2000, True, histcap=4096)
    finally:
        o.lib().qo_set_num_threads(nt)
    assert np.array_equal(ht2, ht)                             # binary128: another reduction order, the same rounded history
    if lat == [4, 6, 10, 6]:
        # fp64: two thread counts differ from each other by per cent, and BOTH sit further from the truth than from each other --
        # the deviation belongs to the precision, which is why the truth and not a second fp64 run is the yardstick
        m = min(n, len(h2))
        self_dev = np.abs(h[:m] / h2[:m] - 1).max()
        assert 1e-3 < self_dev < 0.3
        assert np.abs(h[:n] / ht[:n] - 1).max() > self_dev


def test_judge_accepts_the_reference_algorithm_and_rejects_a_wrong_history(oracle):
    import parity_log

    o = oracle
    lo, g, x = setup(o, [4, 6, 10, 6])
    nt = o.num_threads()
    run_o = lambda: o.solveXX(lo, g, None, x, 0.1, 1e-12, 2000, True, histcap=4096)[3]      # noqa: E731
    run_t = lambda: o.solveXX_ext(lo, g, None, x, 0.1, 1e-12, 2000, True, histcap=4096)[1]   # noqa: E731
    try:
        o.lib().qo_set_num_threads(3 if nt != 3 else 2)
        stand_in = run_o()                                      # the fp64 algorithm in another summation order: what the HIP path is
    finally:
        o.lib().qo_set_num_threads(nt)
    n0 = len(parity_log.RECORDS)
    r = parity_log.judge("yardstick self-test", stand_in, o, run_o, run_t, cache_key="yardstick")
    assert r["truth_spread"] == 0.0 and r["dev_from_truth_hip"] <= r["tolerance"] and "rule" in r
    wrong = stand_in.copy()
    wrong[150:] *= 3.0                                          # a tail that is off by a factor: far outside 3 x yard
    with pytest.raises(AssertionError):
        parity_log.judge("yardstick self-test (wrong)", wrong, o, run_o, run_t, cache_key="yardstick")
    early = stand_in.copy()
    early[50] *= 1 + 1e-8                                       # the first 100 iterations are held to 1e-10 whatever the tail does
    with pytest.raises(AssertionError):
        parity_log.judge("yardstick self-test (early)", early, o, run_o, run_t, cache_key="yardstick")
    del parity_log.RECORDS[n0:]                                 # CPU runs write no parity record file
    with pytest.raises(AssertionError):
        parity_log.tolerance(0.02)                              # the old escape above the cap is gone
