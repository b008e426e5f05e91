"""Remaining pieces of SURVEY.md 8 rows a3 / a9: Staggered.peqDdag and sp.usePrevSoln."""
import os

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


@pytest.mark.parametrize("halo", [False, True])
def test_fermion_force_outer_product(oracle, halo):
    """stagDeriv (stagD.nim:634-664) and the fforce loop (staghmc_spv.nim:831-854)."""
    import qex_amd as q

    o = oracle
    lat = [8, 8, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 4242)
    g = o.gauge_random(lo, rf)
    o.rephase(lo, g)
    x, f0 = o.vector_gaussian(lo, rf), o.gauge_random_tah(lo, rf)
    ctx = q.Context(lat)
    if halo:
        ctx.force_halo(True)
    s = q.newStag(ctx, g)
    f = f0.copy()
    s.stagDeriv(f, x)
    ref = f0.copy()
    o.stag_outer(lo, ref, x, 1.0, -1.0, True)
    assert relerr(f, ref) < 1e-14
    f = f0.copy()
    s.outer(f, x, 0.37, accumulate=False)          # first field: f := scale psi psi^+
    s.outer(f, x, -1.9, accumulate=True)           # later fields: f += ...
    ref = f0.copy()
    o.stag_outer(lo, ref, x, 0.37, 0.37, False)
    o.stag_outer(lo, ref, x, -1.9, -1.9, True)
    assert relerr(f, ref) < 1e-14
    # property: tr f[mu](s) = scale * <x(s+mu), x(s)>, so sum_s tr f = scale * <shift x, x>
    fc, xc = ref[..., 0] + 1j * ref[..., 1], x[..., 0] + 1j * x[..., 1]
    for mu in range(4):
        nb = np.array([lo.neighbor(i, mu, 1) for i in range(lo.vol)])
        want = (0.37 - 1.9) * np.vdot(xc[nb], xc)
        got = np.trace(fc[:, mu], axis1=1, axis2=2).sum()
        assert abs(got - want) < 1e-9 * abs(want)


def test_error_behaviour_of_the_abi(oracle):
    """Bad arguments and out-of-order calls come back as error codes with a message (the host mirror raises);
    nothing aborts, nothing falls back."""
    import ctypes as C
    import qex_amd as q

    L = q.lib()
    lat = [4, 4, 4, 4]
    lo = oracle.Layout(lat)
    ctx = q.Context(lat)
    x = np.zeros((lo.vol, 3, 2))
    from qex_amd._lib import check

    with pytest.raises(q.QexHipError, match="links not set"):            # operator before set_links
        rc = L.qexhip_stag_D(ctx._h, x.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_double(0.1), C.c_double(1.0))
        assert rc != 0
        check(rc)
    with pytest.raises(q.QexHipError):                                   # odd extent
        q.Context([4, 4, 3, 4])
    g = oracle.gauge_unit(lo)
    s = q.newStag(ctx, g)
    with pytest.raises(q.QexHipError, match="Naik|extents"):             # 3-hop links need extents >= 4
        q.newStag3(q.Context([2, 4, 4, 4]), oracle.gauge_unit(oracle.Layout([2, 4, 4, 4])), oracle.gauge_unit(oracle.Layout([2, 4, 4, 4])))
    with pytest.raises(q.QexHipError, match="prepare"):                  # closure used before smearGetForce
        f = np.zeros_like(g)
        check(L.qexhip_nhyp_force(ctx._h, f.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)))
    with pytest.raises(ValueError):                                      # batch sizes
        s.solveXX_batch([x] * 5, [x] * 5, [0.1] * 5, 1e-10, 10)
    with pytest.raises(q.QexHipError, match="mass"):
        s.solveXX_batch([x.copy()], [x], [0.0], 1e-10, 10)
    with pytest.raises(ValueError):                                      # wrong dtype / layout
        s.D(x.astype(np.float32), x, 0.1)
    with pytest.raises(q.QexHipError, match="unknown option"):
        ctx.set_option("nonsense", 1)
    # null handle
    assert L.qexhip_sync(None) != 0


@pytest.mark.gpu
def test_gpu_rsqrtPH_property_of_the_reference():
    """tests/base/tmatfun.nim:33-77 on the device: projectSU of gaussian matrices (qexhip_gauge_reunit, the same
    (x^+ x)^(-1/2) kernel code as projectU) is unitary to the reference's own bound 384 * 3 * eps (per element average)."""
    import qex_amd as q
    from test_oracle_golden import _tmatfun_bound

    lat = [4, 4, 4, 8]
    lo = q.Layout(lat)
    g = q.unit(lo)
    g.reshape(-1, 3, 3, 2)[:2000] = np.random.default_rng(13).standard_normal((2000, 3, 3, 2))   # the matrices of the oracle's test
    ctx = q.Context(lat)
    q.reunit(ctx, g)
    s, bound = _tmatfun_bound(g.reshape(-1, 3, 3, 2)[:2000])
    assert s.max() < bound
    m = g.reshape(-1, 3, 3, 2)
    det = np.linalg.det(m[..., 0] + 1j * m[..., 1])
    assert np.abs(det - 1).max() < 1e-11


@pytest.mark.gpu
def test_gpu_action_and_plaquette_repeat_exactly():
    """tests/base/tactionstress.nim:18-45 and tstressplaq.nim: the same field gives the same action / plaquette on every
    one of many calls (the reference allows 1e-8 relative; the device reductions are ordered, so the bits repeat)."""
    import qex_amd as q

    lat = [8, 8, 8, 16]
    lo = q.Layout(lat)
    g = q.synthetic_random_su3(lo, seed=99)
    ctx = q.Context(lat)
    p0 = q.plaq(ctx, g)
    a0 = q.gaugeAction(ctx, None, plaq=6.0)
    for _ in range(64):
        assert np.array_equal(q.plaq(ctx), p0)
        assert q.gaugeAction(ctx, None, plaq=6.0) == a0
    # gaugeAction1 = -(beta / 3) sum_P Re tr U_P (gaugeAction.nim:61-84) and plaq_i = sum Re tr / (18 V)
    assert abs(a0 + 6.0 * 6.0 * lo.vol * p0.sum()) <= 1e-12 * abs(a0)


@pytest.mark.parametrize("nranks", [2, 4])
def test_bench_two_ranks_share_one_gpu(nranks):
    """The driver's N = 2 (and N = 4) launch line on a ONE-GPU box: the ranks under torch.distributed.run share GPU 0, the communicator takes
    the peer-memory transport by itself (RCCL refuses duplicate devices), and the faces / reductions are REAL exchanges with the
    other process: the self-verification against the committed single-GPU numbers must pass (round 4 could only rehearse the
    control flow here, each rank wrapping its own slab).  One parsable line that carries both ranks."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # QEXHIP_OVERLAP=1 QEXHIP_HOP_SPLIT=2 (left to itself set_links finds that overlapping does not pay between two processes on ONE
    # chip and posts the exchange first): the fused self-pushing sweep -- what 8 GPUs run -- on BOTH legs, the 48^3 x 96 one included (its HISQ Naik
    # multi-shift self-check has 1 296 boundary workgroups per sweep: the case that ran round 5's build into the wait bound,
    # profiles/r06_notes.md section 1)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", QEX_BENCH_FORCE_CANARY="1", QEXHIP_HOP_SPLIT="2", QEXHIP_OVERLAP="1", QEXHIP_PEER_TIMEOUT="20")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29531 + nranks), os.path.join(root, "bench.py"), "--gpus", str(nranks), "--steps", "20", "--warmup", "5"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=root, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2000:])
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    ln = lines[0]
    assert "error" not in ln and ln["n_gpus"] == nranks and ln["value"] > 0 and ln["scaling"] == "strong"
    assert sorted(r["rank"] for r in ln["ranks"]) == list(range(nranks))
    assert 0 < ln["roofline"]["frac"] <= 1
    assert ln["transport"] == "peer" and ln["shared_device"] is True and ln["rccl_nranks"] == nranks
    assert ln["shard_check"]["ok"] is True and ln["repeats"]["n"] >= 1, ln["shard_check"]
    for k in ("interior_us", "boundary_us", "exchange_us", "allreduce_us", "comm_count", "overlap", "per_rank"):
        assert k in ln["multi_gpu"], k
    # the peer-transport canary that precedes the RCCL measurement on a multi-GPU node (forced here): verified like every sharded leg
    can = ln["cg_32x4_peer_transport"]
    assert "error" not in can and can["transport"] == "peer" and can["shard_check"]["ok"] is True and can["value"] > 0, can
    assert ln["multi_gpu"]["sweep"]["form"] == "fused", ln["multi_gpu"]["sweep"]
    # BASELINE configs[3] over the same two real ranks, fused sweeps: every quantity of the oracle-pinned fixture, Naik 10-shift included
    big = ln["cg_48x48x48x96"]
    assert "error" not in big and big["transport"] == "peer" and big["value"] > 0, big
    assert big["shard_check"]["ok"] is True and big["shard_check"]["fixture_pinned_to_oracle"]["ok"] is True, big["shard_check"]
    assert big["multi_gpu"]["sweep"]["form"] == "fused", big["multi_gpu"]["sweep"]
    nk = ln["naik_multishift_48x48x48x96"]
    assert "error" not in nk and nk["links_per_site"] == 16 and nk["value"] > 0, nk


@pytest.mark.parametrize("comm2", ["1", "0", "mbox"])
def test_bench_self_verification_on_the_sharded_code_path(comm2):
    """bench.py --halo on one GPU: ghost zones, faces through a one-rank RCCL communicator on the second stream (its own
    communicator, split off the first: QEXHIP_COMM2 = 1, the default; 0 = the single communicator), the multi-rank reduction
    branches with real all-reduces -- and the run holds ITSELF to tests/golden/shard_checks.json before it times anything:
    plaquettes, |D b|^2, the first 20 CG residuals, the HISQ Naik 10-shift norms.  A wrong face, a wrong ghost link or a wrong
    reduction makes the line carry "error" and the exit status non-zero."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", QEXHIP_COMM2=comm2)
    if comm2 == "mbox":          # the transport `auto` picks between distinct devices of one node: RCCL faces, mailbox rank sums
        env.update(QEXHIP_COMM2="1", QEXHIP_TRANSPORT="mbox")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--halo", "--steps", "20", "--warmup", "3", "--repeats", "2", "--no-cpu", "--no-48x96"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    ln = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    sc = ln["shard_check"]
    assert sc["ok"] is True and not sc["failed"], sc
    assert sc["max_rel"]["operator"] <= 1e-10 and sc["max_rel"]["history"] <= 1e-6 and sc["max_rel"]["solution"] <= 1e-8
    assert ln["rccl_nranks"] == 1 and "error" not in ln
    assert ln["transport"] == ("rccl+mbox" if comm2 == "mbox" else "rccl")


def test_bench_self_verification_fails_loudly(tmp_path):
    """The other half: against a fixture whose |D b|^2 is off by 1e-8 the same run must say which quantity failed, carry
    "error" in its line and exit non-zero (4), without giving up the measurement it already holds."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(root, "tests", "golden", "shard_checks.json")))
    fx["lattices"]["32x32x32x32"]["values"]["Db2"] *= 1 + 1e-8
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(fx))
    env = dict(os.environ, QEX_SHARD_FIXTURE=str(bad))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "2", "--repeats", "1", "--no-cpu", "--no-48x96", "--no-extra"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=root, env=env)
    assert p.returncode == 4, (p.returncode, p.stderr[-1000:])
    ln = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert ln["shard_check"]["ok"] is False and any("Db2" in f for f in ln["shard_check"]["failed"])
    assert "self-verification failed" in ln["error"] and ln["value"] > 0


@pytest.mark.parametrize("lt", [4, 8])
def test_self_verification_quantities_on_thin_slabs(lt):
    """What one rank of an 8- (4-) way split of 32^4 holds is a 32^3 x 4 (x 8) slab: ghost depth 3 of the Naik operator and of
    the HISQ link construction against only four local slices.  The checked quantities of qex_amd.selfcheck on such a slab
    with every kernel in its sharded form (ghost zones, one-rank RCCL exchange, multi-rank reduction branches) must equal the
    periodic kernels' on the same (periodic) lattice."""
    import qex_amd as q
    from qex_amd import selfcheck as sc

    lat = [32, 32, 32, lt]
    g0, g, b = sc.bench_inputs(lat)
    A = q.Context(lat)
    ref = sc.compute(A, g0, g, b)
    A.close()
    B = q.Context(lat)
    B.comm_init(q.Context.unique_id(), 1, 0)
    B.force_halo(True)
    B.set_option("multi_reduce", 1)
    got = sc.compute(B, g0, g, b)
    B.close()
    r = sc.compare(got, {"values": ref})
    assert r["ok"] and r["max_rel"]["operator"] < 1e-12 and r["max_rel"]["history"] < 1e-9, r
    assert got["naik_its"] == ref["naik_its"] and len(got["cg_hist"]) == sc.NHIST + 1


def test_resident_field_entry_points_against_the_oracle(oracle):
    """Round-3 entry points on resident fields, each against the oracle: dev_D (Staggered.D / Ddag), dev_norm2 / dev_redot
    (fieldET.nim:605-625,704-724), dev_zero, dev_solve_batch (n x Staggered.solve), the alias checks of the multi-shift solver,
    release_workspace, comm_count."""
    import qex_amd as q

    o = oracle
    lat = [4, 6, 8, 4]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 77)
    g = o.gauge_warm(lo, 0.5, rf)
    o.rephase(lo, g)
    x, y = o.vector_gaussian(lo, rf), o.vector_gaussian(lo, rf)
    ctx = q.Context(lat)
    assert ctx.comm_count() == 0
    q.newStag(ctx, g)
    xi, yi, ri = ctx.field_new(x), ctx.field_new(y), ctx.field_new()
    for sub, par in (("all", 2), ("even", 0), ("odd", 1)):
        assert abs(ctx.dev_norm2(xi, sub) - o.norm2(lo, x, par)) < 1e-12 * o.norm2(lo, x, par)
        assert abs(ctx.dev_redot(xi, yi, sub) - o.redot(lo, x, y, par)) < 1e-12 * o.norm2(lo, x, par)
    for sc, ref in ((1.0, o.D(lo, g, None, x, 0.13)), (-1.0, o.Ddag(lo, g, None, x, 0.13))):
        ctx.dev_D(ri, xi, 0.13, sc)
        r = ctx.field_download(ri)
        assert np.linalg.norm(r - ref) / np.linalg.norm(ref) < 1e-13
    ctx.dev_zero(ri, "odd")
    r = ctx.field_download(ri)
    assert not r[lo.vol // 2:].any() and r[:lo.vol // 2].any()
    # n x Staggered.solve on resident fields (six systems: two lock-step batches), iteration counts as the single solves
    srcs = [x, y, x + y, x - y, 2 * x, y]
    ms = [0.1, 0.2, 0.15, 0.3, 0.1, 0.25]
    bids = [ctx.field_new(b) for b in srcs]
    xids = [ctx.field_new() for _ in srcs]
    its, r2 = ctx.dev_solve_batch(xids, bids, ms, 1e-14)
    for k, (b, m) in enumerate(zip(srcs, ms)):
        xr, oits, fin = o.solve(lo, g, None, b, m, 1e-14, 100000)
        got = ctx.field_download(xids[k])
        assert abs(its[k] - oits) <= 2 and r2[k] <= 1e-14 and np.linalg.norm(got - xr) / np.linalg.norm(xr) < 1e-7, (k, its[k], oits)
    with pytest.raises(q.QexHipError, match="aliases the source"):
        ctx.dev_solve_batch([bids[0]], [bids[0]], [0.1], 1e-10)
    with pytest.raises(q.QexHipError, match="aliases the source"):        # x of system 0 is b of system 1
        ctx.dev_solve_batch([bids[1], xids[1]], [bids[0], bids[1]], [0.1, 0.1], 1e-10)
    with pytest.raises(q.QexHipError, match="aliases the solution"):      # the same solution field twice, across two lock-step batches
        ctx.dev_solve_batch([xids[0]] + xids[1:5] + [xids[0]], bids, ms, 1e-10)
    # multi-shift on resident fields: aliased solution / source fields are refused, the workspace can be handed back
    sh = [0.2, 4 * (0.4 ** 2 - 0.2 ** 2)]
    with pytest.raises(q.QexHipError, match="source field"):
        ctx.dev_solve_xx_multi([xids[0], bids[0]], bids[0], sh, 1e-10, 100)
    with pytest.raises(q.QexHipError, match="x_ids"):
        ctx.dev_solve_xx_multi([xids[0], xids[0]], bids[0], sh, 1e-10, 100)
    n1, _ = ctx.dev_solve_xx_multi(xids[:2], bids[0], sh, 1e-12, 1000)
    a = [ctx.field_download(i) for i in xids[:2]]
    ctx.release_workspace()
    n2, _ = ctx.dev_solve_xx_multi(xids[:2], bids[0], sh, 1e-12, 1000)
    assert n1 == n2 and all(np.array_equal(u, ctx.field_download(i)) for u, i in zip(a, xids[:2]))
    ctx.comm_init(q.Context.unique_id(), 1, 0)
    assert ctx.comm_count() in (1, 2)


def test_mailbox_selftest_failure_falls_back_to_rccl(oracle):
    """`auto` between distinct devices of one node takes RCCL for the faces and the mailboxes for the rank sums only if a self-test of
    the mailbox all-reduce passes on every rank (comm.cpp); if not, every rank drops the control block and RCCL carries the sums too -- a
    working communicator, not an error; an explicit `mbox` wish turns the same failure into QEXHIP_ERR_COMM.  No one-GPU box reaches
    either branch by itself, so the failure is injected (QEXHIP_TEST_FAIL_MBOX); run in child processes because the hook is read from the
    environment.  commsUtils.nim:195-204 (threadRankSum) is what the sums restate on either transport."""
    import subprocess
    import sys

    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import qex_amd as q
ctx = q.Context([8, 8, 8, 8])
ctx.set_option("transport", 3)
try:
    ctx.comm_init(q.Context.unique_id(), 1, 0)
except q.QexHipError as e:
    print("COMM_INIT_ERROR", str(e)[:200]); sys.exit(0)
ctx.force_halo(True); ctx.set_option("multi_reduce", 1); ctx.set_option("overlap", 1)
rng = np.random.default_rng(1)
g = 0.3 * rng.standard_normal((4096, 4, 3, 3, 2)); b = rng.standard_normal((4096, 3, 2)); b[2048:] = 0
s = q.newStag(ctx, g)
sp = q.SolverParams(r2req=1e-10, maxits=500, verbosity=0)
x = np.zeros_like(b); s.solveEE(x, b, 0.5, sp); ctx.sync()
print("TRANSPORT", ctx.comm_transport()[0], "ITS", sp.iterations)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for hook in ("0", "1", "2"):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", QEXHIP_TEST_FAIL_MBOX=hook)
        p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200, env=env)
        assert p.returncode == 0, (hook, p.stdout[-500:], p.stderr[-1500:])
        out[hook] = [ln for ln in p.stdout.splitlines() if ln.startswith(("TRANSPORT", "COMM_INIT_ERROR"))][-1]
    assert out["0"].startswith("TRANSPORT rccl+mbox"), out
    assert out["1"].startswith("COMM_INIT_ERROR") and "self-test" in out["1"], out
    assert out["2"].startswith("TRANSPORT rccl ITS"), out
    assert out["2"].split()[-1] == out["0"].split()[-1], out          # the same solve on either transport


def test_fused_sweep_placement_follows_the_measured_exchange():
    """Where the fused sweep puts its boundary workgroups -- and how long they wait before they park -- goes by the face exchange
    set_links MEASURES (ten exchanges on the compute stream, max over ranks), not by an assumed link rate: under emulated transport
    of 40 and 80 us per exchange the measured figure grows by what was added, the boundary workgroups move back in the dispatch order,
    the short wait grows with it.  (qshifts.nim:51-131 is the exchange being timed; no oracle: device-side random links.)"""
    import qex_amd as q

    lat = [32, 32, 32, 16]
    vol = int(np.prod(lat))
    rng = np.random.default_rng(3)
    g = 0.3 * rng.standard_normal((vol, 4, 3, 3, 2))
    seen = []
    for emu in (40, 80):
        ctx = q.Context(lat)
        ctx.set_option("transport", 2)
        ctx.comm_init(q.Context.unique_id(), 1, 0)
        ctx.force_halo(True)
        ctx.set_option("emu_exchange_us", emu)
        ctx.set_option("overlap", -2)                 # measure on one rank too
        t0 = ctx.sweep_tuning()
        assert t0["tuned_us_per_sweep"] == [0.0, 0.0, 0.0]
        s = q.newStag(ctx, g)
        t = ctx.sweep_tuning()
        assert "(measured)" in ctx.info() and "boundary_at=" in ctx.info(), ctx.info()
        assert all(v > 0 for v in t["tuned_us_per_sweep"]), t          # exchange first | by sites | fused: all three were timed
        seen.append(t)
        del s
        ctx.close()
    a, b = seen
    assert emu_close(b["exchange_us"] - a["exchange_us"], 40.0), (a, b)
    assert 0.65 <= a["boundary_at"] < b["boundary_at"] <= 1.0, (a, b)
    assert b["fused_spin_us"] > a["fused_spin_us"] >= 25.0, (a, b)


def emu_close(d, want):
    return abs(d - want) < 0.5 * want


def test_overlap_decision_is_measured_when_asked(oracle):
    """With a communicator of more than one rank the library times exchange-first against overlapped sweeps at set_links and
    takes the faster (collective, the slowest rank decides).  Option overlap = -2 asks for the same measurement on one rank:
    the decision is reported with the two timings, a second set_links reuses it, and the operator is unchanged by it."""
    import qex_amd as q

    o = oracle
    lat = [16, 16, 16, 16]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 5)
    g = o.gauge_random(lo, rf)
    o.rephase(lo, g)
    x = o.vector_gaussian(lo, rf)
    ref = o.D(lo, g, None, x, 0.1)
    ctx = q.Context(lat)
    ctx.comm_init(q.Context.unique_id(), 1, 0)
    ctx.force_halo(True)
    s = q.newStag(ctx, g)
    assert ctx.sweep_info()["overlap_measured"] is False          # one rank, default option: the static rule
    ctx.set_option("overlap", -2)
    s = q.newStag(ctx, g)
    si = ctx.sweep_info()
    assert si["overlap_measured"] is True and si["halo"] and si["option_overlap"] == -2
    m = si["measured_us_per_sweep"]
    assert m["exchange_first"] > 0 and m["overlapped"] > 0 and si["overlap"] == (m["overlapped"] < m["exchange_first"])
    r = np.zeros_like(x)
    s.D(r, x, 0.1)
    assert np.linalg.norm(r - ref) / np.linalg.norm(ref) < 1e-13
    s = q.newStag(ctx, g)                                          # measured once per operator shape
    assert ctx.sweep_info()["measured_us_per_sweep"] == m
    for forced in (0, 1):
        ctx.set_option("overlap", forced)
        assert ctx.sweep_info()["overlap"] == bool(forced)
        s.D(r, x, 0.1)
        assert np.linalg.norm(r - ref) / np.linalg.norm(ref) < 1e-13
    ctx.close()


@pytest.mark.parametrize("overlap", [0, 1])
def test_sharded_paths_under_emulated_transport_latency(oracle, overlap):
    """Between distinct GPUs a face exchange takes tens to hundreds of microseconds; in the one-rank rehearsal it is almost
    free, so a consumer that forgot to wait for its ghosts could go unnoticed.  Options emu_exchange_us / emu_allreduce_us put
    a wait in front of every exchange / all-reduce on its stream -- longer than any kernel of this lattice -- and every
    sharded path must still reproduce the periodic kernels: the overlapped sweep with its boundary launch on the comm
    stream, CG and multi-shift histories, the lock-step batch (one RCCL group, overlapped), nHYP smearing with its
    asynchronous level refreshes, and the force chain."""
    import qex_amd as q

    o = oracle
    lat = [16, 16, 16, 8]
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 77)
    g = o.gauge_warm(lo, 0.5, rf)
    gp = g.copy()
    o.rephase(lo, gp)
    g3 = 0.3 * gp
    x, y = o.vector_gaussian(lo, rf), o.vector_gaussian(lo, rf)
    chain = o.gauge_random_tah(lo, rf)
    h = lo.vol // 2
    res = {}
    for mode in ("periodic", "emulated"):
        ctx = q.Context(lat)
        if mode == "emulated":
            ctx.comm_init(q.Context.unique_id(), 1, 0)
            ctx.force_halo(True)
            ctx.set_option("multi_reduce", 1)
            ctx.set_option("batch_multi", 1)
            ctx.set_option("overlap", overlap)
            ctx.set_option("emu_exchange_us", 400)
            ctx.set_option("emu_allreduce_us", 60)
        r = {}
        for naik in (False, True):
            s = q.newStag3(ctx, gp, g3) if naik else q.newStag(ctx, gp)
            d = np.zeros_like(x)
            s.D(d, x, 0.1)
            r["D%d" % naik] = d.copy()
            sp = q.SolverParams(r2req=1e-10, maxits=400, verbosity=0)
            xs = np.zeros_like(x)
            s.solveEE(xs, x, 0.2, sp, histcap=512)
            r["cg%d" % naik] = (sp.iterations, sp.r2hist.copy(), xs.copy())
        masses = [0.2, 0.4, 0.8]
        ys = [np.zeros_like(x) for _ in masses]
        spm = q.SolverParams(r2req=1e-10, maxits=400, verbosity=0)
        s.solveXX_multi(ys, x, [masses[0]] + [4 * (m * m - masses[0] ** 2) for m in masses[1:]], spm, histcap=512)
        r["multi"] = (spm.iterations, spm.r2hist.copy(), [v.copy() for v in ys])
        bs = [x.copy(), y.copy(), x + y]
        for b in bs:
            b[h:] = 0
        bx = [np.zeros_like(b) for b in bs]
        its, _ = s.solveXX_batch(bx, bs, [0.2, 0.3, 0.5], 1e-10, 400, True)
        r["batch"] = (its, [v.copy() for v in bx])
        sg = np.zeros_like(g)
        sf = q.HypCoefs(0.4, 0.5, 0.5).smearGetForce(ctx, g, sg)
        f = np.zeros_like(g)
        sf(f, chain)
        r["nhyp"], r["chain"] = sg.copy(), f.copy()
        sf.release()
        r["plaq"] = q.plaq(ctx, g)
        gf = g.copy()
        q.gaugeFlow(ctx, gf, 1, 0.01)
        r["flow"] = gf
        ctx.close()
        res[mode] = r
    a, b = res["periodic"], res["emulated"]
    rel = lambda u, v: np.linalg.norm((u - v).ravel()) / np.linalg.norm(u.ravel())
    for k in ("D0", "D1", "nhyp", "chain", "flow"):
        assert rel(a[k], b[k]) < 1e-14, k
    assert np.abs(a["plaq"] - b["plaq"]).max() < 1e-15
    for k in ("cg0", "cg1", "multi"):
        assert abs(a[k][0] - b[k][0]) <= 1, k
        n = min(len(a[k][1]), len(b[k][1]), 100)
        assert np.abs(b[k][1][:n] / a[k][1][:n] - 1).max() < 1e-9, k
    assert rel(a["cg0"][2][:h], b["cg0"][2][:h]) < 1e-7 and all(rel(u[:h], v[:h]) < 1e-7 for u, v in zip(a["multi"][2], b["multi"][2]))
    assert all(abs(i - j) <= 1 for i, j in zip(a["batch"][0], b["batch"][0])) and all(rel(u[:h], v[:h]) < 1e-7 for u, v in zip(a["batch"][1], b["batch"][1]))


def test_cg_reentry_continues_the_same_iteration(S):
    """CgState.solve called again with b2 >= 0 (cg.nim:21-27,85,133,155-161,256-261): no set-up, the kept r / p / rzold, the
    new SolverParams' stopping criterion, iterations go on counting.  A solve stopped after 37 iterations by maxits and resumed
    twice must walk through EXACTLY the residual history of the uninterrupted solve (same bits) and end at the same solution;
    the oracle's count pins it to the reference algorithm.  Anything that takes the CG's vectors in between ends the state."""
    q, o, ctx = S.q, S.o, S.ctx
    bid, x1, x2 = ctx.field_new(S.x), ctx.field_new(), ctx.field_new()
    S.s = q.newStag(ctx, S.g)
    n_full, fin_full, h_full = ctx.dev_solve_xx(x1, bid, 0.1, 1e-12, 5000, True, histcap=4096)
    n_a, fin_a, _ = ctx.dev_solve_xx(x2, bid, 0.1, 1e-12, 37, True, histcap=4096)
    assert n_a == 37 and fin_a > 1e-12
    n_b, fin_b, _ = ctx.dev_solve_xx_continue(x2, 1e-6, 5000, histcap=4096)          # a looser criterion first ...
    assert 37 < n_b < n_full and fin_b <= 1e-6
    n_c, fin_c, h_c = ctx.dev_solve_xx_continue(x2, 1e-12, 5000, histcap=4096)        # ... then the real one
    assert n_c == n_full and fin_c == fin_full
    assert np.array_equal(h_c, h_full)
    assert np.array_equal(ctx.field_download(x2), ctx.field_download(x1))
    _, its, _, _ = o.solveXX(S.lo, S.g, None, S.x, 0.1, 1e-12, 5000, True)
    assert abs(n_full - its) <= 1
    n_d, fin_d, _ = ctx.dev_solve_xx_continue(x2, 1e-12, 5000)                        # converged state: returns at once
    assert n_d == n_full and fin_d == fin_full
    ctx.dev_solve_xx_multi([x1], bid, [0.1], 1e-10, 50)                               # takes the work vectors
    with pytest.raises(q.QexHipError):
        ctx.dev_solve_xx_continue(x2, 1e-13, 5000)
    ctx.dev_solve_xx(x2, bid, 0.1, 1e-12, 10, True)
    S.s = q.newStag(ctx, S.g)                                                         # another operator object
    with pytest.raises(q.QexHipError):
        ctx.dev_solve_xx_continue(x2, 1e-12, 5000)
    for f in (bid, x1, x2):
        ctx.field_free(f)


def test_init_releases_everything_on_a_late_failure():
    import qex_amd as q

    """qexhip_init must free the context it built when a HIP call fails late (round-4 verdict, weak 13): with the failure injected
    after the streams, events, device and pinned buffers exist (QEXHIP_TEST_FAIL_INIT), 200 failing inits leave the free device
    memory where it was and the next real init works."""
    import ctypes as C
    import os

    L = q.lib()
    free0, tot = C.c_size_t(0), C.c_size_t(0)
    hip = C.CDLL("libamdhip64.so")
    ctx = q.Context([8, 8, 8, 8])                 # runtime initialised, allocator warm
    ctx.close()
    hip.hipMemGetInfo(C.byref(free0), C.byref(tot))
    os.environ["QEXHIP_TEST_FAIL_INIT"] = "1"
    try:
        i4 = C.c_int * 4
        for _ in range(200):
            h = C.c_void_p()
            rc = L.qexhip_init(C.byref(h), 0, i4(8, 8, 8, 8), i4(1, 1, 1, 1), i4(0, 0, 0, 0))
            assert rc != 0 and not h.value
        assert b"QEXHIP_TEST_FAIL_INIT" in L.qexhip_last_error()
    finally:
        del os.environ["QEXHIP_TEST_FAIL_INIT"]
    free1 = C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(free1), C.byref(tot))
    assert free0.value - free1.value < (8 << 20), (free0.value, free1.value)      # 200 leaked contexts would hold > 50 MB
    ctx = q.Context([8, 8, 8, 8])
    ctx.close()
