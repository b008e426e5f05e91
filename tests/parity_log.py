"""Record of what the history-parity tests actually measured (round-2 verdict, "parity transparency").

Every test that compares a residual history of the HIP path with the oracle's calls `record(...)`; at the end of the
session the records are written to gpurun_out/r04_parity_devs.json (gpurun merges that directory back; the file is then
committed as profiles/r04_parity_devs.json).  Nothing is written when no record was made (the CPU suite).

`tolerance(...)` is the bound those tests enforce on the WHOLE history: CG amplifies the rounding differences between two
equivalent summation orders, so the yardstick is the CPU path against itself when only its reduction order changes
(OpenMP thread count; the reference has the same run-to-run spread).  The drift is chaotic -- one pair
of thread counts is one sample of it -- so the spread is the maximum over several thread counts, the factor over it is 10
and the bound is capped at 1 % (round 2: one pair, factor 1000, cap 10 %).  The first 100 iterations are held to 1e-10
and the north star's 1e-6 applies wherever the CPU path itself stays inside 1e-7.
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = []
FACTOR, CAP, FLOOR, OVER = 10.0, 1e-2, 1e-6, 4.0


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
    """FACTOR x the CPU path's own spread, at least FLOOR, at most CAP -- except where the CPU path deviates from ITSELF by
    more than CAP / 2.  Measured on the GPU box (profiles/r04_parity_devs.json): only the 4 x 6 x 10 x 6 fixtures `sodd` / `soddw`
    (CPU self-spread 1.5e-2 ... 6e-2 over 1, 2, 3, 8 threads; the HIP path 0.3 ... 2.8 times that) and the light-mass Naik
    multi-shift ladder on the same lattice (400 iterations, 3e-2).  Both quantities are single samples of a chaotic drift, so a
    bound below ~3 x the oracle's own reproducibility would test the oracle's luck, not the HIP path: there the bound is OVER = 4
    times that spread and the record carries `cpu_spread_exceeds_cap`.  Everywhere else the cap holds: 8^4 random 4e-7, 8^4
    Naik 2.5e-4, exactly unitary starts 1e-9, and the 32^4 headline 1e-14 over all 192 iterations."""
    if 2.0 * spread > CAP:
        return OVER * spread
    return min(CAP, max(FLOOR, FACTOR * spread))


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
        if spread is not None and 2.0 * spread > CAP:
            r["cpu_spread_exceeds_cap"] = True
    r.update(extra)
    RECORDS.append(r)
    return r


def flush():
    if not RECORDS:
        return None
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "r04_parity_devs.json")
    with open(path, "w") as f:
        json.dump({"rule": "whole history < min(%g, max(%g, %g x CPU self-spread over thread counts)), or %g x that spread where the "
                           "CPU path deviates from itself by more than %g (flag cpu_spread_exceeds_cap); first 100 iterations < 1e-10"
                           % (CAP, FLOOR, FACTOR, OVER, CAP / 2), "records": RECORDS}, f, indent=1)
        f.write("\n")
    return path
