"""Self-verification of a (possibly t-sharded) run against single-GPU numbers committed in
tests/golden/shard_checks.json.

The reference has no such facility: QEX's multi-rank runs are checked by eye against single-rank logs
(the regression harness tests/extra/staghmc_sh/run:43-44 diffs printed energies).  Here every rank of
`bench.py --gpus N` recomputes, on its slab of the benchmark inputs and through the same entry points the
timed region uses, a handful of GLOBAL quantities whose value cannot depend on the partition:

    plaq        the six plaquettes of the unphased start         src/gauge/gaugeUtils.nim:213-282
    b2, Db2     |b|^2 and |D(m) b|^2 on both parities             src/physics/stagD.nim:566-568
    cg_hist     r2/b2 of the first `nhist` CG iterations          src/solvers/cg.nim:172,215-217
    cg_x2       |x|^2 (even sites) after those iterations
    naik_its    iterations of the HISQ Naik 10-shift solve        src/physics/stagSolve.nim:296-345,598
    naik_x2     |xs[k]|^2 (even sites) of every shift

All sums end in the library's all-reduce (src/comms/commsUtils.nim:195-204 in the reference), so every
rank holds the same numbers; rank 0 compares them with the fixture.  Tolerances: 1e-10 relative on the
operator-level quantities (a different partition only changes the summation order), 1e-6 on the
iteration history (the north star's bound), 1e-8 on the converged multi-shift norms.

Nothing here computes on field data: every number comes out of libqexhip.so.
"""
import json
import os

import numpy as np

from . import staggered as st
from .gauge import rephase
from .layout import Layout
from .rng import RngField, RngMilc6

SEED = 987654321                 # src/bench/benchStagProp.nim:22
NHIST = 20
NAIK_MASSES = [float(np.sqrt(k + 2.0)) for k in range(10)]      # src/physics/stagSolve.nim:598
NAIK_R2REQ = 1e-20               # the action-solver setting, src/stagg_pv_hmc/input_hmc.xml:86
TOL = {"operator": 1e-10, "history": 1e-6, "solution": 1e-8}
FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "shard_checks.json")


def lat_key(lat):
    return "x".join(str(int(v)) for v in lat)


def bench_inputs(lat, nranks=1, rank=0):
    """this rank's slab of QEX's benchmark start (benchStagProp.nim:22-33): g.random, its rephased copy, gaussian source"""
    lt = lat[3] // nranks
    lat_loc = list(lat[:3]) + [lt]
    rf = RngField(lat_loc, RngMilc6, SEED, glat=lat, t_offset=rank * lt)
    g0 = rf.random()
    b = rf.gaussian_vector()
    g = g0.copy()
    rephase(Layout(lat_loc), g, t_offset=rank * lt, t_global=lat[3])
    return g0, g, b


def compute(ctx, g0, g, b, mass=0.1, naik=True, kick=None):
    """The checked quantities on the context's (local) lattice.  g0: unphased links, g: with BC + phases, b: source.
    Leaves the operator's links as newStag(g) (the bench's main leg) when it returns."""
    kick = kick or (lambda what: None)
    out = {}
    kick("shard_check: plaquette")
    st.gaugeSet(ctx, g0)
    out["plaq"] = [float(v) for v in st.plaq(ctx)]
    kick("shard_check: D b")
    st.newStag(ctx, g)
    bid, rid, xid = ctx.field_new(b), ctx.field_new(), ctx.field_new()
    out["b2"] = ctx.dev_norm2(bid)
    ctx.dev_D(rid, bid, mass, 1.0)
    out["Db2"] = ctx.dev_norm2(rid)
    kick("shard_check: CG history")
    its, _, hist = ctx.dev_solve_xx(xid, bid, mass, 0.0, NHIST, True, histcap=NHIST + 1)
    out["cg_hist"] = [float(v) for v in hist]
    out["cg_x2"] = ctx.dev_norm2(xid, "even")
    if naik:
        kick("shard_check: HISQ links + Naik multi-shift")
        st.Staggered(ctx, g, smear=st.HisqCoefs())
        m = NAIK_MASSES
        shifts = [m[0]] + [4.0 * (mk * mk - m[0] ** 2) for mk in m[1:]]      # stagSolve.nim:391-394
        xids = [ctx.field_new() for _ in m]
        nits, nh = ctx.dev_solve_xx_multi(xids, bid, shifts, NAIK_R2REQ, 500, True, histcap=501)
        out["naik_its"] = int(nits)
        out["naik_hist_last"] = float(nh[-1])
        out["naik_x2"] = [ctx.dev_norm2(i, "even") for i in xids]
        for i in xids:
            ctx.field_free(i)
        ctx.release_workspace()
        st.newStag(ctx, g)
    for i in (bid, rid, xid):
        ctx.field_free(i)
    ctx.sync()
    return out


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return float("inf")
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def compare(got, want):
    """-> {"ok", "max_rel": {class: value}, "failed": [...]}; `want` is one lattice entry of the fixture"""
    w = want["values"]
    res = {"operator": 0.0, "history": 0.0, "solution": 0.0}
    failed = []

    def chk(cls, name, a, b):
        r = _rel(a, b)
        res[cls] = max(res[cls], r)
        if not (r <= TOL[cls]):
            failed.append("%s: rel %.3e > %.0e" % (name, r, TOL[cls]))

    # the plaquettes of a random start are ~1e-3 of the sum's terms: compare on the scale of the largest (1/6 per plane for U = 1)
    pa, pb = np.asarray(got["plaq"]), np.asarray(w["plaq"])
    r = float(np.max(np.abs(pa - pb)) / (1.0 / 6.0))
    res["operator"] = max(res["operator"], r)
    if not (r <= TOL["operator"]):
        failed.append("plaq: %.3e of 1/6" % r)
    chk("operator", "b2", got["b2"], w["b2"])
    chk("operator", "Db2", got["Db2"], w["Db2"])
    chk("history", "cg_hist", got["cg_hist"], w["cg_hist"])
    chk("history", "cg_x2", got["cg_x2"], w["cg_x2"])
    if "naik_x2" in got and "naik_x2" in w:
        # another t-partition sums the all-reduced partials in another order: a residual within rounding of the stop
        # threshold may end the loop one iteration earlier or later; the difference is reported, only |d| > 1 fails
        res["naik_its_diff"] = int(got["naik_its"]) - int(w["naik_its"])
        if abs(res["naik_its_diff"]) > 1:
            failed.append("naik_its: %d vs %d" % (got["naik_its"], w["naik_its"]))
        chk("solution", "naik_x2", got["naik_x2"], w["naik_x2"])
    its_diff = res.pop("naik_its_diff", None)
    out = {"ok": not failed, "max_rel": {k: float("%.3e" % v) for k, v in res.items()}, "tolerance": dict(TOL), "failed": failed}
    if its_diff is not None:
        out["naik_its_minus_fixture"] = its_diff
    return out


def load_fixture(lat, mass=0.1, path=None):
    path = path or os.environ.get("QEX_SHARD_FIXTURE", FIXTURE)       # the override exists for the test of the failure path
    try:
        fx = json.load(open(path))
    except OSError:
        return None
    e = fx.get("lattices", {}).get(lat_key(lat))
    if e is None or abs(e.get("mass", -1) - mass) > 0 or fx.get("seed") != SEED:
        return None
    return e


def run(ctx, lat, nranks, rank, g0, g, b, mass=0.1, naik=True, kick=None):
    """bench.py's hook: compute on this rank's slab, compare with the committed single-GPU numbers.  None when the
    fixture holds nothing for this lattice / mass."""
    want = load_fixture(lat, mass)
    if want is None:
        return {"ok": None, "skipped": "no fixture for %s, mass %g in tests/golden/shard_checks.json" % (lat_key(lat), mass)}
    got = compute(ctx, g0, g, b, mass, naik=naik and "naik_x2" in want["values"], kick=kick)
    res = compare(got, want)
    res["fixture"] = "tests/golden/shard_checks.json[%s]: %s" % (lat_key(lat), want.get("source", "?"))
    # the committed numbers are the product's on one GPU, and every one of them was recomputed by the CPU oracle (the restatement
    # pinned to the reference's golden vectors); the agreement found then is part of the fixture
    vo = want.get("vs_oracle") or {}
    res["fixture_pinned_to_oracle"] = {"ok": bool(vo.get("ok")), "max_rel": vo.get("max_rel"), "covers": vo.get("covers"), "how": vo.get("how")}
    res["ranks"] = nranks
    return res
