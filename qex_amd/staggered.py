"""Host-side mirror of QEX's staggered operator / solver interface over the C ABI.

Same names, argument meaning and error behaviour as the reference so that tests read like
QEX's own (tests/examples/testStagProp.nim, src/physics/stagSolve.nim:516-680):

    newStag(g) / newStag3(g, g3)      src/physics/stagD.nim:522-564
    Staggered.D / Ddag                src/physics/stagD.nim:566-571
    Staggered.eoReconstruct           src/physics/stagD.nim:583-586
    stagD2 / stagD2ee / stagD2oo      src/physics/stagD.nim:349-395,463-469
    Staggered.solveEE / solveOO       src/physics/stagSolve.nim:134-138
    Staggered.solve (single / multi)  src/physics/stagSolve.nim:224-294,347-446
    SolverParams                      src/solvers/solverBase.nim:10-58

Fields are numpy arrays in the V=1 even-odd host format (qex_amd.layout).  All arithmetic
happens in libqexhip.so on the GPU; nothing here computes on field data.
"""
import ctypes as C
import time
import numpy as np

from . import _lib
from ._lib import check, lib

EVEN, ODD, ALL = 0, 1, 2
_SUBSET = {"even": EVEN, "odd": ODD, "all": ALL}


def _p(a):
    if a is None:
        return None
    if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("fields must be C-contiguous float64 arrays")
    return a.ctypes.data_as(C.c_void_p)


class SolverParams:
    """solverBase.nim:10-58 (fields the staggered path uses)."""

    def __init__(self, r2req=1e-6, maxits=50000, verbosity=1, usePrevSoln=False):
        self.r2req = r2req
        self.maxits = maxits
        self.verbosity = verbosity
        self.usePrevSoln = usePrevSoln
        self.subsetName = "all"
        self.resetStats()

    def resetStats(self):
        self.calls = 0
        self.iterations = 0
        self.iterationsMax = 0
        self.seconds = 0.0
        self.flops = 0.0
        self.r2 = 0.0
        self.r2hist = None

    @property
    def finalIterations(self):
        return self.iterations

    def getStats(self):
        gf = 1e-9 * self.flops / self.seconds if self.seconds > 0 else 0.0
        return f"its: {self.iterations}  secs: {self.seconds:.6g}  Gf/s: {gf:.6g}  r2: {self.r2:.6g}"


def device_count():
    """HIP devices this process can bind (qexhip_device_count): a host picks device = (rank on its node) mod this count"""
    n = C.c_int(0)
    check(lib().qexhip_device_count(C.byref(n)))
    return int(n.value)


class Context:
    """One GPU / one rank (qexhip_init).  rank_geom must be (1,1,1,N)."""

    def __init__(self, lat_local, device=0, rank_geom=(1, 1, 1, 1), rank_coord=(0, 0, 0, 0)):
        self.lat = [int(v) for v in lat_local]
        self.vol = int(np.prod(self.lat))
        self._h = C.c_void_p()
        i4 = C.c_int * 4
        check(lib().qexhip_init(C.byref(self._h), device, i4(*self.lat), i4(*rank_geom), i4(*rank_coord)))
        self.rank_geom = tuple(rank_geom)
        self.rank_coord = tuple(rank_coord)

    def close(self):
        if self._h:
            lib().qexhip_finalize(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        buf = C.create_string_buffer(512)
        check(lib().qexhip_device_info(self._h, buf, 512))
        return buf.value.decode()

    def sync(self):
        check(lib().qexhip_sync(self._h))

    # communicator
    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().qexhip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, nranks, rank):
        check(lib().qexhip_comm_init(self._h, uid, nranks, rank))

    def comm_info(self):
        """(nranks, rank, device, pci bus id) as RCCL / HIP report them; nranks = 0 without a communicator"""
        n, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
        bus = C.create_string_buffer(64)
        check(lib().qexhip_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d), bus, 64))
        return n.value, r.value, d.value, bus.value.decode()

    def comm_transport(self):
        """("none" | "rccl" | "peer", {exchanges, allreduces, arena_allocs, arena_bytes}) -- which transport comm_init chose"""
        buf = C.create_string_buffer(16)
        st = (C.c_long * 4)()
        check(lib().qexhip_comm_transport(self._h, buf, 16, st))
        return buf.value.decode(), dict(zip(("exchanges", "allreduces", "arena_allocs", "arena_bytes"), [int(v) for v in st]))

    def comm_count(self):
        """communicators held: 2 after comm_init (compute stream + overlapped face exchange), 1 with QEXHIP_COMM2=0"""
        n = C.c_int(0)
        check(lib().qexhip_comm_count(self._h, C.byref(n)))
        return n.value

    def sweep_info(self):
        """{"halo", "overlap", "interior_sites", "face_bytes"}: how a one-parity sweep is launched on this context"""
        o = (C.c_int * 8)()
        check(lib().qexhip_stag_sweep_info(self._h, o))
        r = {"halo": bool(o[0]), "overlap": bool(o[1]), "interior_sites": int(o[2]), "face_bytes": int(o[3]), "overlap_measured": bool(o[4]),
             "option_overlap": int(o[7])}
        if o[4]:
            r["measured_us_per_sweep"] = {"exchange_first": int(o[5]), "overlapped": int(o[6])}
        r.update(self.sweep_tuning())
        if o[4]:
            r["measured_us_per_sweep"]["fused"] = int(r["tuned_us_per_sweep"][2] + 0.5)
        return r

    def sweep_tuning(self):
        """what set_links measured and decided for the sweeps of a t-sharded slab (qexhip_stag_sweep_tuning)"""
        t = (C.c_double * 8)()
        check(lib().qexhip_stag_sweep_tuning(self._h, t))
        return {"exchange_us": float(t[0]), "boundary_at": float(t[1]), "tuned_us_per_sweep": [float(t[2]), float(t[3]), float(t[4])],
                "form": "fused" if int(t[5]) == 2 else "by_sites", "fused_spin_us": float(t[6])}

    def force_halo(self, on=True):
        check(lib().qexhip_comm_force_halo(self._h, 1 if on else 0))

    # timers
    def timers_enable(self, on=True):
        """on: False/0 off, True/1 every kernel class, 2 only the Dslash sweeps."""
        check(lib().qexhip_timers_enable(self._h, int(on)))

    def timers_reset(self):
        check(lib().qexhip_timers_reset(self._h))

    def set_option(self, name, value):
        check(lib().qexhip_set_option(self._h, name.encode(), int(value)))

    def timer(self, name):
        cnt, ms = C.c_long(0), C.c_double(0)
        check(lib().qexhip_timers_get(self._h, name.encode(), C.byref(cnt), C.byref(ms)))
        return cnt.value, ms.value

    # device-resident fields
    def field_new(self, host=None):
        fid = C.c_int(0)
        check(lib().qexhip_field_new(self._h, C.byref(fid)))
        if host is not None:
            check(lib().qexhip_field_upload(self._h, fid.value, _p(host)))
        return fid.value

    def field_free(self, fid):
        check(lib().qexhip_field_free(self._h, fid))

    def field_upload(self, fid, host):
        check(lib().qexhip_field_upload(self._h, fid, _p(host)))

    def field_download(self, fid):
        out = np.zeros((self.vol, 3, 2))
        check(lib().qexhip_field_download(self._h, fid, _p(out)))
        return out

    def field_zero(self, fid):
        check(lib().qexhip_field_zero(self._h, fid))

    def dev_dslash(self, r_id, x_id, parity, a=0.0, b=0.0):
        check(lib().qexhip_dev_dslash(self._h, r_id, x_id, parity, a, b))

    def dev_op_xx(self, r_id, x_id, m2, par_even=True):
        check(lib().qexhip_dev_op_xx(self._h, r_id, x_id, m2, 1 if par_even else 0))

    def dev_solve_xx(self, x_id, b_id, mass, r2req, maxits, par_even=True, histcap=0):
        its, fin = C.c_int(0), C.c_double(0)
        hist = np.zeros(max(histcap, 1))
        check(lib().qexhip_dev_solve_xx(self._h, x_id, b_id, mass, r2req, maxits, 1 if par_even else 0,
                                        C.byref(its), C.byref(fin), _p(hist), histcap))
        return its.value, fin.value, hist[: min(histcap, its.value + 1)]

    def dev_solve_xx_continue(self, x_id, r2req, maxits, histcap=0):
        """CgState re-entry (cg.nim:133 `if b2<0: # first call` not taken): go on iterating on the state the last dev_solve_xx
        on x_id left; maxits is the cumulative limit; returns (cumulative iterations, r2/b2, history from iteration 0)"""
        its, fin = C.c_int(0), C.c_double(0)
        hist = np.zeros(max(histcap, 1))
        check(lib().qexhip_dev_solve_xx_continue(self._h, x_id, float(r2req), int(maxits), C.byref(its), C.byref(fin), _p(hist), histcap))
        return its.value, fin.value, hist[: min(histcap, its.value + 1)]

    def dev_solve_xx_multi(self, x_ids, b_id, shifts, r2req, maxits, par_even=True, histcap=0):
        """multi-shift solveXX on resident fields (stagSolve.nim:296-345); shifts[0] = base mass"""
        n = len(x_ids)
        its = C.c_int(0)
        hist = np.zeros(max(histcap, 1))
        check(lib().qexhip_dev_solve_xx_multi(self._h, (C.c_int * n)(*[int(v) for v in x_ids]), b_id,
                                              (C.c_double * n)(*[float(v) for v in shifts]), n, float(r2req), int(maxits),
                                              1 if par_even else 0, C.byref(its), _p(hist), histcap))
        return its.value, hist[: min(histcap, its.value + 1)]

    def release_workspace(self):
        check(lib().qexhip_release_workspace(self._h))

    def dev_zero(self, fid, subset="all"):
        check(lib().qexhip_dev_zero(self._h, int(fid), _SUBSET[subset]))

    def dev_solve_batch(self, x_ids, b_ids, masses, r2req, maxits=1000000):
        """n x Staggered.solve on resident fields (lock-step batches of four); returns (iterations, r2) per system"""
        n = len(x_ids)
        rq = [float(r2req)] * n if np.isscalar(r2req) else [float(v) for v in r2req]
        its, fin = (C.c_int * n)(), (C.c_double * n)()
        check(lib().qexhip_dev_solve_batch(self._h, n, (C.c_int * n)(*[int(v) for v in x_ids]), (C.c_int * n)(*[int(v) for v in b_ids]),
                                           (C.c_double * n)(*[float(v) for v in masses]), (C.c_double * n)(*rq), int(maxits), its, fin))
        return list(its), list(fin)

    def dev_norm2(self, x_id, subset="all"):
        out = C.c_double(0)
        check(lib().qexhip_dev_norm2(self._h, x_id, _SUBSET[subset], C.byref(out)))
        return out.value

    def dev_redot(self, x_id, y_id, subset="all"):
        out = C.c_double(0)
        check(lib().qexhip_dev_redot(self._h, x_id, y_id, _SUBSET[subset], C.byref(out)))
        return out.value

    def dev_dot(self, x_id, y_id, subset="all"):
        """dot(x, y) = sum x^+ y of resident fields (fieldET.nim:677-693), complex"""
        out = (C.c_double * 2)()
        check(lib().qexhip_dev_dot(self._h, x_id, y_id, _SUBSET[subset], out))
        return complex(out[0], out[1])

    def dev_D(self, r_id, x_id, m, sc=1.0):
        """r = m x + sc D x on resident fields (Staggered.D: sc = 1, Ddag: sc = -1)"""
        check(lib().qexhip_dev_D(self._h, r_id, x_id, float(m), float(sc)))

    # field algebra hooks (fieldET.nim:605-625,704-724)
    def norm2(self, x, subset="all"):
        out = C.c_double(0)
        check(lib().qexhip_norm2(self._h, _p(x), _SUBSET[subset], C.byref(out)))
        return out.value

    def redot(self, x, y, subset="all"):
        out = C.c_double(0)
        check(lib().qexhip_redot(self._h, _p(x), _p(y), _SUBSET[subset], C.byref(out)))
        return out.value

    def dot(self, x, y, subset="all"):
        """dot(x, y) = sum x^+ y (fieldET.nim:677-693), complex"""
        out = (C.c_double * 2)()
        check(lib().qexhip_dot(self._h, _p(x), _p(y), _SUBSET[subset], out))
        return complex(out[0], out[1])

    def axpy(self, a, x, y, subset="all"):
        check(lib().qexhip_axpy(self._h, a, _p(x), _p(y), _SUBSET[subset]))

    def xpay(self, x, a, y, subset="all"):
        check(lib().qexhip_xpay(self._h, _p(x), a, _p(y), _SUBSET[subset]))


class Staggered:
    """Staggered[G,T] (stagD.nim:19-22): the links `g` carry BC and phases (rephase)."""

    def __init__(self, ctx, g, g3=None, smear=None, bc="pppa"):
        """smear = None: g (, g3) are the final links.  smear = HisqCoefs(): the operator uses the
        HISQ fat + long links of g, built on the device.  smear = HypCoefs(...): it uses
        rephase(nHYP(g)) with boundary string `bc` ('a' = antiperiodic, as input_hmc.xml:44); with
        g = None the links come from the closure of a preceding HypCoefs.smearGetForce."""
        self.ctx = ctx
        self.g = g
        self.g3 = g3
        if smear is None:
            self.nlinks = 8 if g3 is not None else 4
            check(lib().qexhip_stag_set_links(ctx._h, _p(g), _p(g3)))
        elif isinstance(smear, HisqCoefs):
            self.nlinks = 8
            check(lib().qexhip_stag_set_links_hisq(ctx._h, _p(g) if g is not None else None))
        else:
            self.nlinks = 4
            ap = (C.c_int * 4)(*[1 if ch == "a" else 0 for ch in bc])
            check(lib().qexhip_stag_set_links_nhyp(ctx._h, _p(g) if g is not None else None, float(smear.alpha1), float(smear.alpha2),
                                                   float(smear.alpha3), ap, None))

    # r = m*x + D*x  /  r = m*x - D*x
    def links_info(self):
        """(links per site, format, max deviation) of the operator's links; format 0: 18 reals,
        1: rows 0,1 + sign bit (SU(3) x sign), 2: rows 0,1 + determinant (U(3))"""
        n, cflag, dev = C.c_int(0), C.c_int(0), C.c_double(0)
        check(lib().qexhip_stag_links_info(self.ctx._h, C.byref(n), C.byref(cflag), C.byref(dev)))
        return n.value, cflag.value, dev.value

    def D(self, r, x, m):
        check(lib().qexhip_stag_D(self.ctx._h, _p(r), _p(x), float(m), 1.0))

    def Ddag(self, r, x, m):
        check(lib().qexhip_stag_D(self.ctx._h, _p(r), _p(x), float(m), -1.0))

    def peqDdag(self, r, x, m):
        """r += m*x - D*x  (stagD.nim:572-574)"""
        check(lib().qexhip_stag_D_acc(self.ctx._h, _p(r), _p(x), float(m), -1.0, 1.0))

    def stagDeriv(self, f, x):
        """stagDeriv(s, f, x) without the final rephase (stagD.nim:634-663): f[mu] +-= x (x) x(+mu)^+"""
        check(lib().qexhip_stag_outer(self.ctx._h, _p(f), _p(x), 1.0, -1.0, 1))

    def outer(self, f, psi, scale, accumulate):
        """f[mu][i][a,b] (:= | +=) scale * psi[i][a] * psi(i+mu)[b].adj  (staghmc_spv.nim:831-854)"""
        check(lib().qexhip_stag_outer(self.ctx._h, _p(f), _p(psi), float(scale), float(scale), 1 if accumulate else 0))

    def stagD(self, r, x, subset, m, sc=1.0, a=0.0):
        """stagD(s.se | s.so, r, s.g, x, m, sc, a) (stagD.nim:406-409): r[subset] = a*r + m*x + sc*D*x"""
        check(lib().qexhip_stag_stagD(self.ctx._h, _p(r), _p(x), _SUBSET[subset], float(m), float(sc), float(a)))

    def eoReduce(self, r, b, m):
        """r.even = (D^+ b).even  (stagD.nim:575-581); r.odd is kept"""
        check(lib().qexhip_stag_eo_reduce(self.ctx._h, _p(r), _p(b), float(m)))

    def eoReconstruct(self, r, b, m):
        check(lib().qexhip_stag_eo_reconstruct(self.ctx._h, _p(r), _p(b), float(m)))

    def stagD2(self, r, x, subset, a, b):
        """r[subset] = a*r + b*x + (2D)x  (stagD.nim:349-395)"""
        check(lib().qexhip_stag_dslash(self.ctx._h, _p(r), _p(x), _SUBSET[subset], float(a), float(b)))

    def stagD2ee(self, r, x, m2):
        check(lib().qexhip_stag_op_xx(self.ctx._h, _p(r), _p(x), float(m2), 1))

    def stagD2oo(self, r, x, m2):
        check(lib().qexhip_stag_op_xx(self.ctx._h, _p(r), _p(x), float(m2), 0))

    def _flops(self, its):
        # (s.g.len*4*72+60)*nEven*iterations  (stagSolve.nim:92)
        return float((self.nlinks * 4 * 72 + 60) * (self.ctx.vol // 2) * its)

    def solveXX(self, r, x, m, sp, parEven=True, histcap=0):
        """solveXX(s, r, x, m, sp0, parEven) (stagSolve.nim:57-132): r <- solution, x = rhs."""
        t0 = time.time()
        its, fin = C.c_int(0), C.c_double(0)
        hist = np.zeros(max(histcap, 1))
        check(lib().qexhip_stag_solve_xx(self.ctx._h, _p(r), _p(x), float(m), float(sp.r2req), int(sp.maxits),
                                         1 if parEven else 0, C.byref(its), C.byref(fin), _p(hist), histcap))
        sp.calls += 1
        sp.iterations += its.value
        sp.iterationsMax = max(sp.iterationsMax, its.value)
        sp.seconds += time.time() - t0
        sp.flops += self._flops(its.value)
        sp.r2 = fin.value
        sp.r2hist = hist[: min(histcap, its.value + 1)] if histcap else None
        if sp.verbosity > 1:
            print(("solveEE" if parEven else "solveOO") + "(HIP): " + sp.getStats())

    def solveEE(self, r, x, m, sp, histcap=0):
        self.solveXX(r, x, m, sp, True, histcap)

    def solveOO(self, r, x, m, sp, histcap=0):
        self.solveXX(r, x, m, sp, False, histcap)

    def solve(self, x, b, m, sp):
        """Staggered.solve: x (array or list of arrays) <- D(m)^-1 b  (stagSolve.nim:224-294,347-446)"""
        t0 = time.time()
        its, fin = C.c_int(0), C.c_double(0)
        if isinstance(x, (list, tuple)):
            ms = np.array([float(v) for v in m], dtype=np.float64)
            ptrs = (C.c_void_p * len(x))(*[_p(a).value for a in x])
            check(lib().qexhip_stag_solve_multi(self.ctx._h, ptrs, _p(b), _p(ms), len(x), float(sp.r2req),
                                                int(sp.maxits), C.byref(its), C.byref(fin)))
        else:
            check(lib().qexhip_stag_solve_prev(self.ctx._h, _p(x), _p(b), float(m), float(sp.r2req), int(sp.maxits),
                                               1 if sp.usePrevSoln else 0, C.byref(its), C.byref(fin)))
        sp.calls += 1
        sp.iterations += its.value
        sp.iterationsMax = max(sp.iterationsMax, its.value)
        sp.seconds += time.time() - t0
        sp.flops += self._flops(its.value)
        sp.r2 = fin.value
        if sp.verbosity > 1:
            print("stagSolve(HIP): " + sp.getStats())

    def _batch(self, fn_name, xs, bs, ms, r2req, maxits, parEven=None):
        n = len(xs)
        if not (1 <= n <= 4 and len(bs) == n and len(ms) == n):
            raise ValueError("batch solve: 1..4 systems, one source and one mass each")
        rq = [float(r2req)] * n if np.isscalar(r2req) else [float(v) for v in r2req]
        xp = (C.c_void_p * n)(*[_p(a).value for a in xs])
        bp = (C.c_void_p * n)(*[_p(a).value for a in bs])
        mv, rv = (C.c_double * n)(*[float(v) for v in ms]), (C.c_double * n)(*rq)
        its, fin = (C.c_int * n)(), (C.c_double * n)()
        if parEven is None:
            check(lib().qexhip_stag_solve_batch(self.ctx._h, n, xp, bp, mv, rv, int(maxits), its, fin))
        else:
            check(lib().qexhip_stag_solve_xx_batch(self.ctx._h, n, xp, bp, mv, rv, int(maxits), 1 if parEven else 0, its, fin))
        return list(its), list(fin)

    def solve_batch(self, xs, bs, ms, sps):
        """n (<= 4) x Staggered.solve on these links in lock-step: the links are streamed once per sweep for
        all systems.  sps: one SolverParams (shared r2req / maxits) or one per system; each gets the
        statistics of its own system, exactly as n calls of solve would record them."""
        sl = [sps] * len(xs) if isinstance(sps, SolverParams) else list(sps)
        t0 = time.time()
        its, fin = self._batch("solve", xs, bs, ms, [sp.r2req for sp in sl], min(sp.maxits for sp in sl))
        dt = (time.time() - t0) / len(xs)
        for sp, i, f in zip(sl, its, fin):
            sp.calls += 1
            sp.iterations += i
            sp.iterationsMax = max(sp.iterationsMax, i)
            sp.seconds += dt
            sp.flops += self._flops(i)
            sp.r2 = f
        return its

    def solveXX_batch(self, xs, bs, ms, r2req, maxits, parEven=True):
        """n (<= 4) x solveEE / solveOO in lock-step; returns (iterations, r2/b2) per system"""
        return self._batch("xx", xs, bs, ms, r2req, maxits, parEven)

    def solveXX_multi(self, xs, b, shifts, sp, parEven=True, histcap=0):
        """Staggered.solveXX(xs, b, ms, sp, subset) (stagSolve.nim:296-345): shifts[0] = base mass."""
        its = C.c_int(0)
        sh = np.array([float(v) for v in shifts], dtype=np.float64)
        hist = np.zeros(max(histcap, 1))
        ptrs = (C.c_void_p * len(xs))(*[_p(a).value for a in xs])
        check(lib().qexhip_stag_solve_xx_multi(self.ctx._h, ptrs, _p(b), _p(sh), len(xs), float(sp.r2req),
                                               int(sp.maxits), 1 if parEven else 0, C.byref(its), _p(hist), histcap))
        sp.iterations += its.value
        sp.r2hist = hist[: min(histcap, its.value + 1)] if histcap else None


def newStag(ctx, g):
    return Staggered(ctx, g)


def newStag3(ctx, g, g3):
    return Staggered(ctx, g, g3)


# ---- gauge observables / Wilson flow (src/gauge/gaugeUtils.nim:213-282, src/gauge/wflow.nim:21-67) ----
def plaq(ctx, g=None):
    if g is not None:
        check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    out = np.zeros(6)
    check(lib().qexhip_plaq(ctx._h, _p(out)))
    return out


_FLOW_KIND = {"Wilson": 0, "rect": 0, "adj": 1}


def gaugeForce(ctx, g, cplaq=1.0, rect=0.0, adjplaq=0.0):
    """gc.gaugeForce(g, f) (gaugeAction.nim:334-350) / gc.forceA(g, f) (:742-747) with
    gc = GaugeActionCoeffs(plaq, rect | adjplaq)."""
    if rect != 0.0 and adjplaq != 0.0:
        raise ValueError("rect and adjplaq are separate code paths in QEX (gaugeForce vs forceA)")
    check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    f = np.zeros_like(g)
    kind = 1 if adjplaq != 0.0 else 0
    check(lib().qexhip_gauge_force_general(ctx._h, _p(f), float(cplaq), float(adjplaq if kind else rect), kind))
    return f


def flowEQ(ctx, loop=1, g=None):
    """[E_s, E_t, Q] of the resident (or given) gauge field: g.fmunu(loop) -> densityE, topoQ
    (gaugeUtils.nim:1162-1271; `EQ` of src/flow/gauge_flow.nim:360-379)."""
    if g is not None:
        check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    out = np.zeros(3)
    check(lib().qexhip_flow_EQ(ctx._h, int(loop), _p(out)))
    return out


def flowMeasure(ctx, g=None):
    """(plaq[6], [E_s, E_t, Q]) of the resident (or given) field in one pass: what the measure block of a flow loop prints
    after every step (src/flow/gauge_flow.nim:139-156,360-379)"""
    if g is not None:
        check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    pl, eq = np.zeros(6), np.zeros(3)
    check(lib().qexhip_flow_measure(ctx._h, _p(pl), _p(eq)))
    return pl, eq


def gaugeAction(ctx, g=None, plaq=1.0, rect=0.0, adjplaq=0.0):
    """gc.gaugeAction1(g) / gc.actionA(g) (gaugeAction.nim:61-142,614-681) of g (or of the resident field)"""
    if g is not None:
        check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    out = C.c_double(0)
    check(lib().qexhip_gauge_action(ctx._h, float(plaq), float(rect), float(adjplaq), C.byref(out)))
    return out.value


def gaugeUpdate(ctx, g, p, t):
    """mdt: g := exp(t p) g in place (staghmc_sh.nim:429-435)"""
    check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    check(lib().qexhip_gauge_update(ctx._h, _p(p), float(t)))
    check(lib().qexhip_gauge_get(ctx._h, _p(g)))


class ResidentMD:
    """mdt / mdv / the force-gradient shifts of QEX's HMC drivers (staghmc_sh.nim:429-640) on links and momenta that
    stay on the device between the updates (qexhip_md_*).  Forces are left on the device by `gauge_force()` (source 0)
    and by the nHYP closure called with f = None (source 1: `sf.gforce(None, ...)`, `sf.fforce_solve(None, ...)` of
    `HypCoefs.smearGetForce(ctx, None)`, which smears the resident links)."""

    GAUGE, NHYP = 0, 1

    def __init__(self, ctx):
        self.ctx = ctx

    def begin(self, g, p):
        """upload links (None: keep the resident ones) and momenta (None: keep the resident ones, e.g. RngField.dev_momenta)"""
        check(lib().qexhip_md_begin(self.ctx._h, _p(g), _p(p)))

    def end(self, g=None, p=None):
        check(lib().qexhip_md_end(self.ctx._h, _p(g), _p(p)))

    def momentum_norm2(self):
        out = C.c_double(0)
        check(lib().qexhip_md_momentum_norm2(self.ctx._h, C.byref(out)))
        return out.value

    def update_links(self, t):
        """mdt: U <- exp(t p) U"""
        check(lib().qexhip_md_update_links(self.ctx._h, float(t)))

    def gauge_force(self, plaq=1.0, rect=0.0, adjplaq=0.0):
        check(lib().qexhip_md_gauge_force(self.ctx._h, float(plaq), float(rect), float(adjplaq)))

    def kick(self, source, t):
        """mdv: p += t f"""
        check(lib().qexhip_md_kick(self.ctx._h, int(source), float(t)))

    def shift_links(self, source, t):
        """fgv / fgvf: U <- exp(t f) U"""
        check(lib().qexhip_md_shift_links(self.ctx._h, int(source), float(t)))

    def save_links(self):
        check(lib().qexhip_md_save_links(self.ctx._h))

    def restore_links(self):
        check(lib().qexhip_md_restore_links(self.ctx._h))


def reunit(ctx, g):
    """g.projectSU in place (gaugeUtils.nim:1333-1334)"""
    check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    check(lib().qexhip_gauge_reunit(ctx._h))
    check(lib().qexhip_gauge_get(ctx._h, _p(g)))


def wline(ctx, path, g=None):
    """g.wline(path) (gaugeUtils.nim:1079-1112); path entries +-(mu+1)"""
    if g is not None:
        check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    out = (C.c_double * 2)()
    arr = (C.c_int * len(path))(*[int(v) for v in path])
    check(lib().qexhip_wline(ctx._h, arr, len(path), out))
    return complex(out[0], out[1])


def s4_gauge(ctx, g=None):
    """g.s4_gauge() (stagg_pv_hmc/staghmc_spv_meas.nim:25-65): peo[dir][even/odd], a (4, 2) array, of g or of the resident field"""
    if g is not None:
        check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    out = np.zeros(8)
    check(lib().qexhip_plaq_s4(ctx._h, _p(out)))
    return out.reshape(4, 2)


def ploops(ctx, g=None):
    """the four Polyakov loops [g.wline([mu+1] * L_mu) for mu in 0..3] (gauge_flow.nim:137-156 `meas_ploop`,
    staghmc_sh.nim:281-291 `ploop`) of g, or of the resident field, in one call"""
    if g is not None:
        check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    out = np.zeros(8)
    check(lib().qexhip_polyakov_loops(ctx._h, _p(out)))
    return [complex(out[2 * d], out[2 * d + 1]) for d in range(4)]


def gaugeFlow(ctx, g, steps, eps, measure=None, flow_act="Wilson", plaq=1.0, rect=0.0, adjplaq=0.0):
    """g.gaugeFlow(steps, eps): measure (wflow.nim:21-67), or the fork's
    gc.gaugeFlow(flow_act, g, steps, eps): measure (src/flow/flow.nim:22-90) with
    gc = GaugeActionCoeffs(plaq, rect, adjplaq).  g is modified in place."""
    kind = _FLOW_KIND[flow_act]
    c2 = adjplaq if kind else rect
    check(lib().qexhip_gauge_set(ctx._h, _p(g)))
    if measure is None:
        check(lib().qexhip_wflow_general(ctx._h, int(steps), float(eps), float(plaq), float(c2), kind))
    else:
        for n in range(1, steps + 1):
            check(lib().qexhip_wflow_general(ctx._h, 1, float(eps), float(plaq), float(c2), kind))
            measure(n * eps)
    check(lib().qexhip_gauge_get(ctx._h, _p(g)))


def gaugeSet(ctx, g):
    """upload g as the context's resident gauge field (what plaq / gaugeFlow(..) with g given do first)"""
    check(lib().qexhip_gauge_set(ctx._h, _p(g)))


def gaugeFlowResident(ctx, steps, eps):
    """`steps` RK3 steps of g.gaugeFlow(steps, eps) (wflow.nim:21-67) on the resident field; nothing crosses PCIe"""
    check(lib().qexhip_wflow(ctx._h, int(steps), float(eps)))


# ---- link construction (src/gauge/fat7l.nim, src/physics/hisqLinks.nim, src/gauge/hypsmear.nim) ----
class HisqCoefs:
    """hisqLinks.nim:3-24: `var hc: HisqCoefs; hc.init(); hc.smear(g, fl, ll)`"""

    def init(self):
        return self

    def force(self, ctx, g, dsdsu, dsdsul):
        """smearGetForce(...)'s smearedForce(dsdu, dsdsu, dsdsul) (hisqsmear.nim:55-90); returns dsdu"""
        f = np.zeros_like(g)
        check(lib().qexhip_hisq_force(ctx._h, _p(g), _p(dsdsu), _p(dsdsul), _p(f)))
        return f

    def smear(self, ctx, g, fl, ll):
        check(lib().qexhip_hisq_smear(ctx._h, _p(g), _p(fl), _p(ll)))

    def smearGetForce(self, ctx, g, fl=None, ll=None):
        """hisqsmear.nim:55-90: smear g (into fl, ll if given) and return the closure smearedForce(dsdu, dsdsu, dsdsul);
        u, v, w, su, sul stay on the device until release().  Staggered(ctx, None, smear=HisqCoefs()) then builds the
        operator from the closure's links."""
        check(lib().qexhip_hisq_prepare(ctx._h, _p(g), _p(fl), _p(ll)))

        def smearedForce(dsdu, dsdsu, dsdsul):
            check(lib().qexhip_hisq_closure_force(ctx._h, _p(dsdsu), _p(dsdsul), _p(dsdu)))

        def fermionForce(f, psis, scales):
            """fermionForce (hisqhmc.nim:496-541) for the fields psis; the closure must hold the PHASED links"""
            n = len(psis)
            arr = (C.c_void_p * n)(*[_p(p).value for p in psis])
            sc = (C.c_double * n)(*[float(v) for v in scales])
            check(lib().qexhip_hisq_fermion_force(ctx._h, _p(f), arr, sc, n))

        smearedForce.fermionForce = fermionForce
        smearedForce.release = lambda: check(lib().qexhip_hisq_release(ctx._h))
        return smearedForce


def fat7lDeriv(ctx, g, dfl, coef, dll=None, naik=0.0):
    """fat7lDeriv (fat7lderiv.nim): derivative through makeImpLinks(g, coef, naik) of the chains dfl (, dll)"""
    d = np.zeros_like(g)
    cf = (C.c_double * 5)(*[float(v) for v in coef])
    check(lib().qexhip_fat7_deriv(ctx._h, _p(g), _p(dfl), cf, _p(dll), float(naik), _p(d)))
    return d


class HypCoefs:
    """hypsmear.nim:15-18,260-275: `coef.smear(g, fl)` (forward smearing only)"""

    def __init__(self, alpha1=0.4, alpha2=0.5, alpha3=0.5):
        self.alpha1, self.alpha2, self.alpha3 = alpha1, alpha2, alpha3

    def smear(self, ctx, g, fl):
        check(lib().qexhip_nhyp_smear(ctx._h, _p(g), _p(fl), float(self.alpha1), float(self.alpha2), float(self.alpha3)))

    def smearGetForce(self, ctx, g, fl=None):
        """hypsmear.nim:49-247: smear g (into fl if given) and return the closure
        `smearedForce(f, chain)`; the intermediate fields stay on the device until the closure's
        `release()` (or the next smearGetForce on this context)."""
        check(lib().qexhip_nhyp_prepare(ctx._h, _p(g), float(self.alpha1), float(self.alpha2), float(self.alpha3), _p(fl)))

        def smearedForce(f, chain):
            check(lib().qexhip_nhyp_force(ctx._h, _p(f), _p(chain)))

        def gforce(f, plaq=1.0, rect=0.0, adjplaq=0.0):
            """gforce(act, g, sg, f, smear_force) (staghmc_spv.nim:217-228)"""
            check(lib().qexhip_nhyp_gauge_force(ctx._h, _p(f), float(plaq), float(rect), float(adjplaq)))

        def fforce(f, psis, scales, bc="aaaa"):
            """fforce + smeared_one_link_force (staghmc_spv.nim:716-865) for the fields psis"""
            n = len(psis)
            arr = (C.c_void_p * n)(*[_p(p).value for p in psis])
            sc = (C.c_double * n)(*[float(v) for v in scales])
            ap = (C.c_int * 4)(*[1 if ch == "a" else 0 for ch in bc])
            check(lib().qexhip_nhyp_fermion_force(ctx._h, _p(f), arr, sc, n, ap, None))

        def fforce_solve(f, phis, masses, scales, r2req, maxits=1000000, bc="aaaa"):
            """the whole fforce incl. its solves (staghmc_sh.nim:387-427) on the operator's current links
            (build them from this closure: Staggered(ctx, None, smear=...)); returns the iteration counts"""
            n = len(phis)
            arr = (C.c_void_p * n)(*[_p(p).value for p in phis])
            ms = (C.c_double * n)(*[float(v) for v in masses])
            sc = (C.c_double * n)(*[float(v) for v in scales])
            rq = (C.c_double * n)(*([float(r2req)] * n if np.isscalar(r2req) else [float(v) for v in r2req]))
            ap = (C.c_int * 4)(*[1 if ch == "a" else 0 for ch in bc])
            its = (C.c_int * n)()
            check(lib().qexhip_nhyp_fforce(ctx._h, _p(f), n, arr, ms, sc, rq, int(maxits), ap, None, its))
            return list(its)

        def fforce_solve_dev(f, phi_ids, masses, scales, r2req, maxits=1000000, bc="aaaa"):
            """fforce_solve with the pseudofermion fields resident (field ids of ctx)"""
            n = len(phi_ids)
            ids = (C.c_int * n)(*[int(v) for v in phi_ids])
            ms = (C.c_double * n)(*[float(v) for v in masses])
            sc = (C.c_double * n)(*[float(v) for v in scales])
            rq = (C.c_double * n)(*([float(r2req)] * n if np.isscalar(r2req) else [float(v) for v in r2req]))
            ap = (C.c_int * 4)(*[1 if ch == "a" else 0 for ch in bc])
            its = (C.c_int * n)()
            check(lib().qexhip_nhyp_fforce_dev(ctx._h, _p(f), n, ids, ms, sc, rq, int(maxits), ap, None, its))
            return list(its)

        smearedForce.gforce, smearedForce.fforce, smearedForce.fforce_solve = gforce, fforce, fforce_solve
        smearedForce.fforce_solve_dev = fforce_solve_dev
        smearedForce.release = lambda: check(lib().qexhip_nhyp_release(ctx._h))
        return smearedForce


def makeImpLinks(ctx, fl, g, coef, ll=None, naik=0.0):
    """makeImpLinks(fl, gf, coef, ll, gfLong, naik) with gfLong = gf (fat7l.nim:77-165);
    coef = (oneLink, threeStaple, fiveStaple, sevenStaple, lepage)."""
    cf = (C.c_double * 5)(*[float(v) for v in coef])
    check(lib().qexhip_fat7(ctx._h, _p(g), cf, _p(fl), _p(ll), float(naik)))
