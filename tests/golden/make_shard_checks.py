#!/usr/bin/env python
"""Generate tests/golden/shard_checks.json: the partition-independent quantities `bench.py --gpus N` recomputes on
its sharded lattice before the timed region (qex_amd/selfcheck.py).

    python tests/golden/make_shard_checks.py                # on ONE GPU: 8^4, 32^4, 48^3x96 from the product (N = 1, periodic)
    python tests/golden/make_shard_checks.py --oracle-only   # no GPU: the oracle's numbers only (8^4, 32^4), printed, not written

The committed numbers are the PRODUCT's on one GPU (that is what a sharded run must reproduce: same kernels, another
partition), each lattice cross-checked against the oracle (oracle/qex_oracle.c, the pinned CPU restatement of the
reference) where the oracle finishes in minutes: everything at 8^4 and 32^4; at 48^3x96 the non-Naik quantities.  The
agreement found is stored next to the values (`vs_oracle`).  The reference holds no such numbers (QEX prints its
benchmark residuals, it asserts none: src/bench/benchStagProp.nim:59-72).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

LATS = [[8, 8, 8, 8], [32, 32, 32, 32], [48, 48, 48, 96]]
MASS = 0.1


def oracle_values(lat, mass=MASS, naik=True):
    """the same quantities from the CPU oracle (test infrastructure), same seeds, same order of draws"""
    from oracle import oracle as o
    from qex_amd import selfcheck as sc

    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, sc.SEED)
    g0 = o.gauge_random(lo, rf)
    b = o.vector_gaussian(lo, rf)
    g = g0.copy()
    o.rephase(lo, g)
    out = {"plaq": [float(v) for v in o.plaq(lo, g0)], "b2": float(o.norm2(lo, b))}
    out["Db2"] = float(o.norm2(lo, o.D(lo, g, None, b, mass)))
    x, its, fin, hist = o.solveXX(lo, g, None, b, mass, 0.0, sc.NHIST, True, histcap=sc.NHIST + 1)
    out["cg_hist"] = [float(v) for v in hist]
    out["cg_x2"] = float(o.norm2(lo, x, 0))
    if naik:
        fl, ll = o.hisq_smear(lo, g)
        m = sc.NAIK_MASSES
        shifts = [m[0]] + [4.0 * (mk * mk - m[0] ** 2) for mk in m[1:]]
        xs, nits, nh = o.solveXX_multi(lo, fl, ll, b, shifts, sc.NAIK_R2REQ, 500, True, histcap=501)
        out["naik_its"] = int(nits)
        out["naik_hist_last"] = float(nh[-1])
        out["naik_x2"] = [float(o.norm2(lo, v, 0)) for v in xs]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-only", action="store_true")
    ap.add_argument("--no-oracle-48x96", action="store_true", help="skip the oracle cross-check of the largest lattice")
    ap.add_argument("--crosscheck", default=None, metavar="LAT",
                    help="no GPU: recompute EVERY quantity of the committed entry LAT (e.g. 48x48x48x96, Naik multi-shift included: "
                         "~40 GB, several minutes) with the oracle, compare, and record the agreement in the entry's vs_oracle")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "shard_checks.json"))
    args = ap.parse_args()
    from qex_amd import selfcheck as sc

    if args.crosscheck:
        fx = json.load(open(args.out))
        e = fx["lattices"][args.crosscheck]
        lat = [int(v) for v in args.crosscheck.split("x")]
        t0 = time.time()
        ov = oracle_values(lat, mass=e["mass"], naik=True)
        c = sc.compare(ov, e)                           # the oracle's numbers held to the committed (product) ones
        e["vs_oracle"] = {"max_rel": c["max_rel"], "ok": c["ok"], "failed": c["failed"], "covers": sorted(ov.keys()),
                          "how": "tests/golden/make_shard_checks.py --crosscheck %s (CPU oracle, %d threads, %.0f s)" % (
                              args.crosscheck, __import__("oracle.oracle", fromlist=["x"]).num_threads(), time.time() - t0)}
        assert c["ok"], c
        with open(args.out, "w") as f:
            json.dump(fx, f, indent=1)
            f.write("\n")
        print(args.crosscheck, json.dumps(e["vs_oracle"]))
        return
    if args.oracle_only:
        for lat in LATS[:2]:
            print(sc.lat_key(lat), json.dumps(oracle_values(lat)))
        return
    import qex_amd as q

    fx = {"seed": sc.SEED, "nhist": sc.NHIST, "naik_masses": sc.NAIK_MASSES, "naik_r2req": sc.NAIK_R2REQ,
          "generator": "tests/golden/make_shard_checks.py on one MI355X (product, N = 1, periodic kernels)", "lattices": {}}
    for lat in LATS:
        t0 = time.time()
        g0, g, b = sc.bench_inputs(lat)
        ctx = q.Context(lat)
        vals = sc.compute(ctx, g0, g, b, MASS)
        info = ctx.info()
        ctx.close()
        del g0, g, b
        entry = {"mass": MASS, "values": vals, "source": "libqexhip.so on one GPU (%s)" % info.split(";")[0]}
        big = lat == LATS[2]
        if not (big and args.no_oracle_48x96):
            ov = oracle_values(lat, naik=not big)
            cmp_ = sc.compare({k: v for k, v in vals.items() if k in ov}, {"values": ov})
            entry["vs_oracle"] = {"max_rel": cmp_["max_rel"], "ok": cmp_["ok"], "failed": cmp_["failed"],
                                  "covers": sorted(ov.keys())}
            assert cmp_["ok"], (lat, cmp_)
        fx["lattices"][sc.lat_key(lat)] = entry
        print("%s done in %.1f s: %s" % (sc.lat_key(lat), time.time() - t0, json.dumps(entry.get("vs_oracle"))), flush=True)
    with open(args.out, "w") as f:
        json.dump(fx, f, indent=1)
        f.write("\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
