"""Writes tests/golden/cg_8x8x8x8.json from the CPU oracle.

The reference holds no asserted known-answer for a Dslash output or a CG residual history
(SURVEY.md 8c "Gap"), and it cannot be run here (Nim toolchain absent), so these vectors come
from the oracle AFTER it has been pinned to the reference's golden sets G1/G2/G4/G5/G6
(tests/test_oracle_golden.py).  They freeze the oracle's Dslash/CG behaviour: BASELINE.json
configs[0] (8^4 staggered CG, mass 0.1), gaussian source (src/physics/stagSolve.nim:542) and
point source (:576-583), RngMilc6 seed 987654321 (src/bench/benchStagProp.nim:22).

Run:  python tests/golden/make_fixtures.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as o  # noqa: E402


def main():
    lat, seed, mass, r2req = [8, 8, 8, 8], 987654321, 0.1, 1e-12
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, seed)
    g = o.gauge_random(lo, rf)
    plaq = o.plaq(lo, g).tolist()
    o.rephase(lo, g)
    b = o.vector_gaussian(lo, rf)
    _, its, fin, hist = o.solveXX(lo, g, None, b, mass, r2req, 1000, True, histcap=64)
    p = np.zeros_like(b)
    p[0, 0, 0] = 1.0
    x, itp, finp = o.solve(lo, g, None, p, mass, r2req, 10000)
    g3 = o.gauge_random(lo, rf)
    o.rephase(lo, g3)
    g3 *= 0.3
    r = np.zeros_like(b)
    o.stagD2(lo, g, g3, r, b, 2, 0.0, 0.0)
    masses = [0.1, 0.2, 0.4]
    xs, itm, finm = o.solve_multi(lo, g, None, b, masses, r2req, 10000)
    fx = {
        "lat": lat, "seed": seed, "mass": mass, "r2req": r2req, "plaq": plaq,
        "hist_gaussian": hist[:40].tolist(), "its_gaussian": int(its),
        "its_point": int(itp), "x2_point": float((x * x).sum()),
        "naik_D2_norm2": float((r * r).sum()),
        "masses": masses, "multi_x2": [float((v * v).sum()) for v in xs], "its_multi": int(itm),
    }
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cg_8x8x8x8.json")
    json.dump(fx, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
