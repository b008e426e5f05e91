"""Record of what the history-parity tests actually measured (round-2 verdict, "parity transparency").

Every test that compares a residual history of the HIP path with the oracle's calls `record(...)`; at the end of the
session the records are written to gpurun_out/r05_parity_devs.json (gpurun merges that directory back; the file is then
committed as profiles/r05_parity_devs.json).  Nothing is written when no record was made (the CPU suite).

The first 100 iterations are always held to 1e-10 against the oracle, the iteration count to +-1, and the north star's 1e-6
over the WHOLE history on the BASELINE configs.  Elsewhere CG amplifies rounding along its tail until two fp64 runs of the SAME
algorithm differ by per cent, and the bound on the tail needs a yardstick that is not one oracle run's luck (round-4 verdict):

`judge(...)`  measures everything against the TRUTH -- the same algorithm in IEEE binary128 throughout
    (oracle/qex_oracle_ext.inc; reproducible to the last bit of the rounded history across thread counts wherever it is used,
    which `truth_spread` records: 80-bit long double was tried first and is not).  The yardstick is how far the fp64 REFERENCE
    ALGORITHM strays from that trajectory at several thread counts (`yard` = max over thread counts and iterations); the HIP
    path must stay within max(1e-6, 3 x yard) of the truth.  Measured: on 4x6x10x6 every fp64 run -- any thread count, and
    the HIP path -- sits 1e-1 from the truth and within a few 1e-2 of each other: the loss of orthogonality is a property of the
    precision, not of the implementation.  Where even binary128 is not reproducible (truth_spread > 1e-9: no trajectory exists
    to compare with) the tails are compared through their convergence ENVELOPES instead: the iteration at which the running
    minimum of r2/b2 first passes each level 10^(-j/4) may differ from the oracle's by at most max(2, 2 x the oracle's own
    shift between thread counts).
`tolerance(spread)`  is the older rule, kept for the one place where a binary128 run is not affordable (32^4, 25 M numbers per
    vector): 10 x the fp64 oracle's self-spread over thread counts, floor 1e-6, cap 1e-2 -- and NO escape above the cap any more:
    a spread above half the cap is an error that says to use judge().
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = []
FACTOR, CAP, FLOOR = 10.0, 1e-2, 1e-6
TRUTH_FACTOR, TRUTH_REPRO = 3.0, 1e-9


def spread_over_threads(o, solve_hist, hist_ref, counts=None):
    """max_k max_i |h_k[i]/hist_ref[i] - 1| with h_k the oracle's history at thread count k (hist_ref: all threads)"""
    nt = o.num_threads()
    counts = counts or sorted({1, 2, 3, max(1, nt // 2)} - {nt})
    worst, per = 0.0, {}
    try:
        for k in counts:
            o.lib().qo_set_num_threads(int(k))
            h = np.asarray(solve_hist())
            n = min(len(h), len(hist_ref))
            s = float(np.max(np.abs(h[:n] / np.asarray(hist_ref)[:n] - 1)))
            per[int(k)] = s
            worst = max(worst, s)
    finally:
        o.lib().qo_set_num_threads(nt)
    return worst, per


def tolerance(spread):
    """FACTOR x the fp64 oracle's own spread over thread counts, at least FLOOR, at most CAP.  Where the oracle deviates from
    ITSELF by more than CAP / 2 this rule has nothing to say: use judge() (round 4 had a 4 x spread escape here)."""
    assert 2.0 * spread <= CAP, "fp64 self-spread %.2e: this history needs the binary128 yardstick (parity_log.judge)" % spread
    return min(CAP, max(FLOOR, FACTOR * spread))


def _dev(a, b):
    n = min(len(a), len(b))
    return np.abs(np.asarray(a)[:n] / np.asarray(b)[:n] - 1)


def envelope_crossings(h):
    """first iteration at which the running minimum of the history is below 10^(-j/4), j = 1..48 (len(h) where never)"""
    e = np.minimum.accumulate(np.asarray(h))
    lv = 10.0 ** (-np.arange(1, 49) / 4.0)
    return np.array([int(np.argmax(e <= L)) if (e <= L).any() else len(e) for L in lv])


_CACHE = {}


def judge(name, hip_hist, o, run_oracle, run_truth, its=None, baseline=False, counts=None, cache_key=None, **extra):
    """Hold one residual history of the HIP path to the oracle and to the binary128 truth (module docstring).
    run_oracle() / run_truth(): histories of the fp64 oracle / its binary128 twin at the CURRENT thread count.
    cache_key: tests that judge several HIP runs of one system share the CPU runs."""
    nt = o.num_threads()
    counts = counts or sorted({1, 2, 3, max(1, nt // 2)} - {nt})
    if cache_key is not None and cache_key in _CACHE:
        truth, truth_spread, orc = _CACHE[cache_key]
    else:
        truth = np.asarray(run_truth())
        orc = {nt: np.asarray(run_oracle())}
        try:
            o.lib().qo_set_num_threads(2 if nt != 2 else 3)
            truth_spread = float(_dev(run_truth(), truth).max())
            for k in counts:
                o.lib().qo_set_num_threads(int(k))
                orc[int(k)] = np.asarray(run_oracle())
        finally:
            o.lib().qo_set_num_threads(nt)
        if cache_key is not None:
            _CACHE[cache_key] = (truth, truth_spread, orc)
    h_orc = orc[nt]
    dev_orc = _dev(hip_hist, h_orc)
    assert len(dev_orc) > 10
    yard_by = {k: float(_dev(h, truth).max()) for k, h in orc.items()}
    yard = max(yard_by.values())
    dev_truth = _dev(hip_hist, truth)
    cross = envelope_crossings(h_orc)
    shift_hip = int(np.abs(envelope_crossings(hip_hist) - cross).max())
    shift_yard = max(int(np.abs(envelope_crossings(h) - cross).max()) for h in orc.values())
    r = record(name, dev_orc, its=its, truth_spread=truth_spread, dev_from_truth_hip=float(dev_truth.max()),
               dev_from_truth_oracle_by_threads={str(k): v for k, v in yard_by.items()}, envelope_shift_hip=shift_hip,
               envelope_shift_oracle=shift_yard, **extra)
    assert dev_orc[:100].max() < 1e-10, dev_orc[:100].max()                  # before amplification sets in
    if baseline:
        assert dev_orc.max() < 1e-6, dev_orc.max()                          # north star, BASELINE configs
    if truth_spread <= TRUTH_REPRO:
        r["tolerance"] = tol = max(FLOOR, TRUTH_FACTOR * yard)
        r["rule"] = "|hip / truth - 1| <= max(1e-6, 3 x max over thread counts of |oracle / truth - 1|)"
        assert dev_truth.max() <= tol, (name, float(dev_truth.max()), yard_by)
    else:
        r["truth_unreproducible"] = True
        r["tolerance_iterations"] = tol = max(2, 2 * shift_yard)
        r["rule"] = "binary128 not reproducible: convergence envelopes within max(2, 2 x the oracle's own shift) iterations"
        assert shift_hip <= tol, (name, shift_hip, shift_yard)
    return r


def record(name, dev, spread=None, spread_by_threads=None, its=None, tol=None, **extra):
    dev = np.asarray(dev)
    r = {"test": name, "n": int(len(dev)), "dev_first100_max": float(dev[:100].max()) if len(dev) else 0.0,
         "dev_max": float(dev.max()) if len(dev) else 0.0, "dev_argmax": int(dev.argmax()) if len(dev) else 0}
    if spread is not None:
        r["cpu_self_spread"] = float(spread)
    if spread_by_threads:
        r["cpu_self_spread_by_threads"] = {str(k): float(v) for k, v in spread_by_threads.items()}
    if its is not None:
        r["iterations_hip_oracle"] = [int(its[0]), int(its[1])]
    if tol is not None:
        r["tolerance"] = float(tol)
    r.update(extra)
    RECORDS.append(r)
    return r


def flush():
    if not RECORDS:
        return None
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "r05_parity_devs.json")
    with open(path, "w") as f:
        json.dump({"rule": "first 100 iterations < 1e-10 and iteration count +-1 against the fp64 oracle; whole history: 1e-6 on the BASELINE "
                           "configs; elsewhere |hip/truth - 1| <= max(1e-6, 3 x max over thread counts |oracle/truth - 1|) with truth = "
                           "the same CG in binary128 (records with `rule`), or, where a binary128 run is not affordable (32^4), "
                           "min(%g, max(%g, %g x fp64 self-spread over thread counts))" % (CAP, FLOOR, FACTOR), "records": RECORDS}, f, indent=1)
        f.write("\n")
    return path
