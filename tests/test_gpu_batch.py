"""Lock-step solves of several systems on the same links (qexhip_stag_solve_xx_batch / solve_batch): the
Hasenbusch chains and pbp repetitions of the HMC drivers (staghmc_sh.nim:260-272,339-364,394-404) issued so
that the links are streamed once for all systems.  Every system must come out as the single-system call
delivers it: same iteration count, same solution."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)


def _setup(o, lat, kind, halo=False):
    import qex_amd as q

    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 4711)
    ctx = q.Context(lat)
    if halo:
        ctx.force_halo(True)          # the kernels and the face exchange of a t-sharded rank, on one GPU
    if kind == "random":            # 18-real links
        g = o.gauge_random(lo, rf); o.rephase(lo, g); s = q.newStag(ctx, g); fmt = 0
    elif kind == "warm":            # rows 0,1 + sign
        g = o.gauge_warm(lo, 0.5, rf); o.rephase(lo, g); s = q.newStag(ctx, g); fmt = 1
    elif kind == "nhyp":            # rows 0,1 + determinant
        g = o.gauge_warm(lo, 0.5, rf); s = q.Staggered(ctx, g, smear=q.HypCoefs(), bc="pppa"); fmt = 2
    else:                           # Naik: 16 links
        g = o.gauge_warm(lo, 0.5, rf); o.rephase(lo, g)
        g3 = 0.3 * o.gauge_warm(lo, 0.6, rf); o.rephase(lo, g3)
        s = q.newStag3(ctx, g, g3); fmt = 2
    assert s.links_info()[1] == fmt
    return q, lo, rf, s


@pytest.mark.parametrize("kind", ["random", "warm", "nhyp", "naik"])
@pytest.mark.parametrize("par_even", [True, False])
def test_solveXX_batch_equals_single(oracle, kind, par_even):
    q, lo, rf, s = _setup(oracle, [8, 4, 6, 4] if kind != "naik" else [4, 8, 6, 4], kind)
    ms = [0.1, 0.2, 0.4, 0.05]
    bs = [oracle.vector_gaussian(lo, rf) for _ in ms]
    bs[3][:] = 0.0                                           # a zero source finishes at once (cg.nim:155,174)
    for n in (4, 3, 1):
        xs = [np.zeros_like(b) for b in bs[:n]]
        its, fin = s.solveXX_batch(xs, bs[:n], ms[:n], [1e-14, 1e-12, 1e-16, 1e-12][:n], 5000, par_even)
        for j in range(n):
            sp = q.SolverParams(r2req=[1e-14, 1e-12, 1e-16, 1e-12][j], maxits=5000, verbosity=0)
            x1 = np.zeros_like(bs[j])
            s.solveXX(x1, bs[j], ms[j], sp, par_even)
            assert its[j] == sp.iterations
            assert np.array_equal(xs[j], x1) or relerr(xs[j], x1) < 1e-13
    # a shared iteration cap stops every system at that count
    xs = [np.zeros_like(b) for b in bs[:2]]
    its, _ = s.solveXX_batch(xs, bs[:2], ms[:2], 0.0, 7, par_even)
    assert its == [7, 7]


@pytest.mark.parametrize("kind", ["warm", "nhyp"])
def test_solve_batch_equals_single_and_oracle(oracle, kind):
    q, lo, rf, s = _setup(oracle, [8, 4, 6, 4], kind)
    ms = [0.1, 0.2, 0.4]
    bs = [oracle.vector_gaussian(lo, rf) for _ in ms]
    bs[0][lo.vol // 2:] = 0                                  # phi.odd := 0 (staghmc_sh.nim:753): reconstruct-right
    bs[1][:lo.vol // 2] = 0                                  # odd-only source
    sps = [q.SolverParams(r2req=1e-20, maxits=10000, verbosity=0) for _ in ms]
    xs = [np.zeros_like(b) for b in bs]
    its = s.solve_batch(xs, bs, ms, sps)
    for j in range(3):
        sp = q.SolverParams(r2req=1e-20, maxits=10000, verbosity=0)
        x1 = np.zeros_like(bs[j])
        s.solve(x1, bs[j], ms[j], sp)
        assert its[j] == sp.iterations == sps[j].iterations
        assert relerr(xs[j], x1) < 1e-13
        r = np.zeros_like(x1)
        s.D(r, xs[j], ms[j])
        assert ((r - bs[j]) ** 2).sum() / (bs[j] ** 2).sum() <= 1e-20


@pytest.mark.parametrize("kind", ["random", "warm", "naik"])
@pytest.mark.parametrize("multi", [False, True])
def test_batch_on_the_sharded_path(oracle, kind, multi):
    """ghost-zone kernels + face exchange of all systems (what each rank of a t-sharded job runs); `multi` also takes
    the multi-rank reduction branch (local sums -> one all-reduce for all systems -> bookkeeping)."""
    q, lo, rf, s = _setup(oracle, [8, 8, 8, 8], kind, halo=True)
    s.ctx.set_option("batch_multi", 1 if multi else 0)
    ms = [0.1, 0.2, 0.4]
    bs = [oracle.vector_gaussian(lo, rf) for _ in ms]
    xs = [np.zeros_like(b) for b in bs]
    its, _ = s.solveXX_batch(xs, bs, ms, 1e-14, 5000, True)
    for j in range(3):
        sp = q.SolverParams(r2req=1e-14, maxits=5000, verbosity=0)
        x1 = np.zeros_like(bs[j])
        s.solveXX(x1, bs[j], ms[j], sp, True)
        assert abs(its[j] - sp.iterations) <= (1 if multi else 0)
        assert relerr(xs[j], x1) < (1e-7 if multi else 1e-13)
    # and the full solve with reconstruction
    bs[0][lo.vol // 2:] = 0
    xs = [np.zeros_like(b) for b in bs]
    sps = [q.SolverParams(r2req=1e-18, maxits=10000, verbosity=0) for _ in ms]
    s.solve_batch(xs, bs, ms, sps)
    for j in range(3):
        r = np.zeros_like(bs[j])
        s.D(r, xs[j], ms[j])
        assert ((r - bs[j]) ** 2).sum() / (bs[j] ** 2).sum() <= 1e-18
    s.ctx.set_option("batch_multi", 0)


@pytest.mark.parametrize("kind", ["random", "warm", "naik"])
def test_batch_fused_sweep_on_the_peer_transport(oracle, kind):
    """The lock-step sweep of a t-sharded slab as ONE launch (k_dslash_mrhs_fused: push of all systems' faces | interior | boundary
    with a short wait | cleanup), one rank on the peer transport (its own neighbour through the receive arena): against the split
    by sites, with every boundary block parked (same bits as unparked: local hops first either way, partial in the block's own slot),
    and against single-system solves on the periodic kernels.  staghmc_sh.nim:339-364 (the Hasenbusch solves) is what runs on it."""
    import qex_amd as q

    lat = [8, 8, 8, 16]
    o = oracle
    lo = o.Layout(lat)
    rf = o.RngField(lo, o.RNG_MILC6, 4711)
    if kind == "naik":
        g = o.gauge_warm(lo, 0.5, rf); o.rephase(lo, g)
        g3 = 0.3 * o.gauge_warm(lo, 0.6, rf); o.rephase(lo, g3)
    else:
        g = (o.gauge_random if kind == "random" else lambda l, r: o.gauge_warm(l, 0.5, r))(lo, rf); o.rephase(lo, g)
        g3 = None
    ms = [0.1, 0.2, 0.4]
    bs = [o.vector_gaussian(lo, rf) for _ in ms]
    ref_ctx = q.Context(lat)
    sref = q.newStag3(ref_ctx, g, g3) if g3 is not None else q.newStag(ref_ctx, g)
    ref = []
    for j in range(3):
        sp = q.SolverParams(r2req=1e-14, maxits=5000, verbosity=0)
        x1 = np.zeros_like(bs[j])
        sref.solveXX(x1, bs[j], ms[j], sp, True)
        ref.append((sp.iterations, x1))
    keep = {}
    for name, opts in (("by_sites", dict(hop_split=0)), ("fused", dict(hop_split=2)), ("fused_parked", dict(hop_split=2, fused_spin_us=-2)),
                       ("fused_spin0", dict(hop_split=2, fused_spin_us=0))):
        ctx = q.Context(lat)
        ctx.set_option("transport", 2)
        ctx.comm_init(q.Context.unique_id(), 1, 0)
        ctx.force_halo(True)
        ctx.set_option("batch_multi", 1)
        ctx.set_option("overlap", 1)
        for k, v in opts.items():
            ctx.set_option(k, v)
        s = q.newStag3(ctx, g, g3) if g3 is not None else q.newStag(ctx, g)
        xs = [np.zeros_like(b) for b in bs]
        its, _ = s.solveXX_batch(xs, bs, ms, 1e-14, 5000, True)
        ctx.sync()
        for j in range(3):
            assert abs(its[j] - ref[j][0]) <= 1, (name, j, its, ref[j][0])
            assert relerr(xs[j], ref[j][1]) < 1e-7, (name, j)
        # a shared iteration cap with one system switched off early (a zero source is done at once): the credits still go back
        xz = [np.zeros_like(b) for b in bs]
        bz = [bs[0], np.zeros_like(bs[1]), bs[2]]
        itz, _ = s.solveXX_batch(xz, bz, ms, 0.0, 9, True)
        assert itz[0] == 9 and itz[2] == 9 and itz[1] == 0, itz
        keep[name] = ([x.copy() for x in xs], list(its), [x.copy() for x in xz])
        st = ctx.comm_transport()[1]
        assert st["exchanges"] > 20, st
        del s
        ctx.close()
    for name in ("fused_parked", "fused_spin0"):
        assert keep[name][1] == keep["fused"][1], name
        for a, b in zip(keep[name][0] + keep[name][2], keep["fused"][0] + keep["fused"][2]):
            assert np.array_equal(a, b), name
