"""Golden set G7: the reference's HMC regression log tests/extra/staghmc_sh/ref.0,
which its own harness (tests/extra/staghmc_sh/run) compares at 2e-11 relative.

`Begin H` of trajectory 1 is a closed-form consequence of the path this repository implements
(src/examples/staghmc_sh.nim:716-765): momentum refresh p.randomTAH(r) -> T; nHYP smear + setBC +
stagPhase; psi_i gaussian; phi_i = D(-h_i)^-1 D(-m_i) psi_i (even sites); the action solves
psi_i = D(m_i)^-1 D(h_i) phi_i and Sf_i = |psi_i|^2 / 2.  It pins RngMilc6 gaussians, randomTAH,
Staggered.D (phases, boundary, normalisation, mass sign) and Staggered.solve (even-odd CG +
reconstruction, r2req = 1e-24) to numbers printed by the reference itself.
"""
import os

import numpy as np
import pytest

LAT = [8, 8, 8, 8]
SEED = 987654321
MASS, HMASSES = 0.1, [0.2, 0.4]
ARSQ = 9.999999999999999e-25
# tests/extra/staghmc_sh/ref.0:117
BEGIN_H = 18451.47947589929
BEGIN_SF = [6115.074514620805, 6296.481015505035, 6143.045791623304]
BEGIN_T = -103.1218458498552
RTOL = 2e-11          # tests/extra/staghmc_sh/run:45


def begin_H(lo, o, D, solve):
    """staghmc_sh.nim:716-765 with g = unit; D(x, m) and solve(b, m) are the operator under test."""
    rf = o.RngField(lo, o.RNG_MILC6, SEED)
    p = o.gauge_random_tah(lo, rf)                         # p.randomTAH r
    T = 0.5 * (p * p).sum() - 16.0 * lo.vol
    psi = [o.vector_gaussian(lo, rf) for _ in range(len(HMASSES) + 1)]
    n = len(psi)
    phi = []
    for i in range(n):
        mi = -MASS if i == 0 else -HMASSES[i - 1]
        ph = solve(D(psi[i], mi), -HMASSES[i]) if i != n - 1 else D(psi[i], mi)
        ph = ph.copy()
        ph[lo.vol // 2:] = 0                               # phi.odd := 0
        phi.append(ph)
    fa = []
    for i in range(n - 1):                                 # faction (:339-364)
        x = solve(D(phi[i], HMASSES[i]), MASS if i == 0 else HMASSES[i - 1])
        fa.append((x * x).sum())
    x = solve(phi[-1], HMASSES[-1])
    fa.append((x * x).sum())
    return T, [0.5 * v for v in fa]


def check(T, Sf):
    assert abs(T - BEGIN_T) < RTOL * abs(16.0 * 4096)      # T is a difference of two ~6.5e4 numbers
    for a, b in zip(Sf, BEGIN_SF):
        assert abs(a - b) < RTOL * b
    assert abs(sum(Sf) + T - BEGIN_H) < RTOL * BEGIN_H     # Sg = 0 on the unit gauge


def test_oracle_reproduces_reference_begin_H(oracle):
    o = oracle
    lo = o.Layout(LAT)
    sg = o.nhyp_smear(lo, o.gauge_unit(lo), 0.4, 0.5, 0.5)
    o.rephase(lo, sg)
    D = lambda x, m: o.D(lo, sg, None, x, m)
    solve = lambda b, m: o.solve(lo, sg, None, b, m, ARSQ, 1000000)[0]
    check(*begin_H(lo, o, D, solve))


@pytest.mark.gpu
def test_gpu_reproduces_reference_begin_H(oracle):
    import qex_amd as q

    o = oracle
    lo = o.Layout(LAT)
    ctx = q.Context(LAT)
    s = q.Staggered(ctx, o.gauge_unit(lo), smear=q.HypCoefs(0.4, 0.5, 0.5), bc="pppa")   # smear + setBC + stagPhase on device

    def D(x, m):
        r = np.zeros_like(x)
        s.D(r, x, m)
        return r

    def solve(b, m):
        x = np.zeros_like(b)
        s.solve(x, b, m, q.SolverParams(r2req=ARSQ, maxits=1000000, verbosity=0))
        return x

    check(*begin_H(lo, o, D, solve))


# ---- whole trajectories: End H, pbp, plaquette, Polyakov loops, reversibility (ref.0, ref.1, ref.2) ----
def _cmp(e, G):
    for k in ("H", "Sg", "T"):
        assert abs(e[k] - G[k]) < RTOL * max(abs(G[k]), 16.0 * 4096), (k, e[k], G[k])
    assert len(e["Sf"]) == len(G["Sf"])
    for a, g_ in zip(e["Sf"], G["Sf"]):
        assert abs(a - g_) < RTOL * g_


def _check_trajectory(r, second=True):
    G = r.cfg.gold
    g0 = r.g.copy()
    _cmp(r.refresh(), G["begin"])
    r.evolve()
    _cmp(r.finish_energies(), G["end"])
    m = r.measure(accepted=G["accept"], g0=g0)
    for a, g_ in zip(m["pbp"], G["pbp"]):
        assert abs(a - g_) < RTOL * g_
    for a, g_ in zip(m["plaq"], G["plaq"]):
        assert abs(a - g_) < RTOL * g_
    for a, g_ in zip(m["ploop"], G["ploop"]):
        assert abs(a - g_) < RTOL * max(abs(g_), 0.1)
    if "pbp_iters" in G:
        assert m["pbp_iters"] == [G["pbp_iters"]] * 2                 # "stagSolve: 101" twice (ref.0:122,124)
    if "force_stats" in G:
        # Solver[force] / Solver[action] statistics the reference printed: count, floor(avg), max per field
        for v, (cnt, avg, mx) in zip(r.stats["force_iters"], G["force_stats"]):
            assert (len(v), sum(v) // len(v), max(v)) == (cnt, avg, mx)
        assert [max(v) for v in r.stats["action_iters"]] == G["action_max"]
    if second and "begin2" in G:
        # trajectory 2: new momenta and pseudofermions from the continuing RNG streams, End H, the
        # reversibility check (Reversed H comes back to Begin H), accept/reject, measurements
        g0 = r.g.copy()
        _cmp(r.refresh(), G["begin2"])
        r.evolve()
        _cmp(r.finish_energies(), G["end2"])
        _cmp(r.reverse_check(), G["reversed2"])
        m2 = r.measure(accepted=G["accept2"], g0=g0)
        for a, g_ in zip(m2["pbp"], G["pbp2"]):
            assert abs(a - g_) < RTOL * g_
        for a, g_ in zip(m2["plaq"], G["plaq2"]):
            assert abs(a - g_) < RTOL * g_


@pytest.mark.parametrize("run", [0, 1, 2])
def test_oracle_replays_reference_trajectory(oracle, run):
    """G7 in full: the oracle's nHYP smearing + force chain, fermion and adjoint-plaquette gauge
    forces, D and solve carry the reference's HMC runs (one species + 2 Hasenbusch masses, 2MN gauge
    integrator; two species, force-gradient gauge integrator; unequal Hasenbusch step counts) to the
    printed End H, pbp, plaquettes and Polyakov loops (the reference's harness compares these at 2e-11)."""
    import hmc_replay as R

    _check_trajectory(R.Replay(oracle, R.OracleBackend(oracle, oracle.Layout(R.LAT)), R.CONFIGS[run]))


@pytest.mark.gpu
@pytest.mark.parametrize("run,halo,resident", [(0, False, False), (1, False, False), (0, True, False), (0, False, True), (1, False, True),
                                               (2, True, True)])
def test_gpu_replays_reference_trajectory(oracle, run, halo, resident):
    """The same trajectories with every operator -- smearing, solves, forces, link update, action,
    reunitarisation, plaquettes, Polyakov loops -- AND the random numbers (momenta, pseudofermions,
    pbp sources: qex_amd.RngField) coming from libqexhip; the oracle supplies nothing but the unit start."""
    import qex_amd as q
    import hmc_replay as R

    rng = q.RngField(R.LAT, q.RngMilc6, R.SEED)          # the product's own newRNGField
    # halo: the same trajectory with every kernel in its t-sharded form (forced ghost zones on one GPU)
    # resident: the MD evolution through qexhip_md_* (links, momenta and forces never leave the device)
    _check_trajectory(R.Replay(oracle, R.HipBackend(q, R.LAT, halo=halo, resident=resident), R.CONFIGS[run], rng=rng),
                      second=(run == 0 and not halo and not resident))


@pytest.mark.gpu
@pytest.mark.parametrize("run,resident", [(0, False), (1, True)])
def test_gpu_replays_reference_trajectory_over_real_ranks(run, resident):
    """G7 with REAL neighbours: the reference's golden HMC log -- Begin / End H sector by sector, pbp, plaquettes, Polyakov loops,
    the reversibility check and the second trajectory (run 0), the solver iteration statistics -- reproduced at the reference
    harness's own 2e-11 by TWO processes that each hold half of the 8^4 lattice and share the one GPU (peer-memory transport):
    nHYP closure and force chain, fermion forces with their Hasenbusch solves, adjoint-plaquette gauge force, force-gradient
    updates (run 1, MD loop resident on the device), all t-sharded with faces and sums crossing between processes
    (tests/golden_rank_worker.py)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", QEXHIP_PEER_TIMEOUT="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29581 + run), os.path.join(root, "tests", "golden_rank_worker.py"), str(run), str(int(resident))]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=root, env=env)
    ok = [ln for ln in p.stdout.splitlines() if ln.startswith("GOLDEN_RANK_OK")]
    if p.returncode != 0 or len(ok) != 2:
        print(p.stdout[-3000:])
        print(p.stderr[-8000:])
    assert p.returncode == 0 and len(ok) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("run", [0, 1])
def test_gpu_replays_reference_trajectory_with_both_ends_on_the_device(oracle, run):
    """G7 with nothing but scalars (and the 36-byte generator states) crossing PCIe between the unit start and the last
    printed number: momenta, pseudofermions and pbp sources drawn into HBM by the RngMilc6 field's device form, `Begin H` /
    `End H` from resident links, momenta and vectors, reunit / pbp / plaquettes / Polyakov loops on the resident field --
    against the reference's log at its 2e-11, CG iteration statistics included; and the generator states afterwards are bit for
    bit those of the host generators after the same draws."""
    import qex_amd as q
    import hmc_replay as R

    rng = q.RngField(R.LAT, q.RngMilc6, R.SEED)
    be = R.HipBackend(q, R.LAT, resident=True)
    r = R.DeviceEndsReplay(oracle, be, R.CONFIGS[run], rng)
    G = r.cfg.gold
    _cmp(r.refresh(), G["begin"])
    r.evolve()
    _cmp(r.finish_energies(), G["end"])
    assert G["accept"]
    m = r.measure(accepted=True)
    for a, g_ in zip(m["pbp"], G["pbp"]):
        assert abs(a - g_) < RTOL * g_
    for a, g_ in zip(m["plaq"], G["plaq"]):
        assert abs(a - g_) < RTOL * g_
    for a, g_ in zip(m["ploop"], G["ploop"]):
        assert abs(a - g_) < RTOL * max(abs(g_), 0.1)
    if "pbp_iters" in G:
        assert m["pbp_iters"] == [G["pbp_iters"]] * 2
    if "force_stats" in G:
        for v, (cnt, avg, mx) in zip(r.stats["force_iters"], G["force_stats"]):
            assert (len(v), sum(v) // len(v), max(v)) == (cnt, avg, mx)
        assert [max(v) for v in r.stats["action_iters"]] == G["action_max"]
    # the same draws on the host generators: identical states word for word
    ref = q.RngField(R.LAT, q.RngMilc6, R.SEED)
    ref.randomTAH()
    for _ in r.cfg.fields:
        ref.gaussian_vector()
    ref.u1_vector()
    ref.u1_vector()
    assert np.array_equal(ref.state(), rng.state())


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["-resident"]])
def test_example_log_against_reference_log(extra):
    """The reference's own regression procedure (tests/extra/staghmc_sh/run:43-46): run the example, keep the
    MEAS / Begin / End / Reversed lines, compare number by number with the golden log at 2e-11 -- here for
    examples/staghmc_sh.py, which performs every field operation in libqexhip."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "staghmc_sh.py"), "-run", "0", "-trajs", "2"] + extra,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
    keep = re.compile(r"^MEASploop|^MEASplaq|^MEASpbp|(Begin|End|Reversed) H:|^(ACCEPT|REJECT)")
    mine = [ln for ln in out.splitlines() if keep.search(ln)]
    gold = [ln for ln in open(os.path.join(root, "tests", "golden", "staghmc_sh", "ref.0.check")).read().splitlines() if keep.search(ln)]
    assert len(mine) == len(gold), out
    num = re.compile(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?")
    compared = 0
    for a, b in zip(mine, gold):
        assert a.split(":")[0].split()[0] == b.split(":")[0].split()[0], (a, b)     # same kind of line, same verdict
        na, nb = [float(v) for v in num.findall(a)], [float(v) for v in num.findall(b)]
        assert na and len(na) == len(nb), (a, b)
        if " T: " in b:
            # "<Begin|End|Reversed> H: h  Sg: sg  Sf: @[@[..]]  T: t": H, Sg and T carry the 16 V = 65536 offset of the
            # kinetic term (as _cmp above), every Sf_i is held to the harness's 2e-11 of itself
            scales = [max(abs(y), 16.0 * 4096) for y in nb]
            scales[2:-1] = [abs(y) for y in nb[2:-1]]
        elif "dH" in b:
            scales = [1e-6 / RTOL] * len(nb)                # dH is a difference of two H ~ 2e4..4e4: absolute 1e-6
        else:
            scales = [max(abs(y), 0.1) for y in nb]
        for x, y, sc in zip(na, nb, scales):
            assert abs(x - y) <= RTOL * sc, (a, b)
            compared += 1
    assert compared >= 40, compared          # 2 trajectories: 5 energy lines, 4 pbp, 3 plaq, 3 ploop, 2 verdict lines
