"""ctypes binding of the CPU oracle (oracle/qex_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (qex_amd) never imports this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libqexoracle.so")


def build(force=False):
    if force or not os.path.exists(_LIB) or (
        os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "qex_oracle.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libqexoracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        L.qo_layout_new.restype = vp
        L.qo_layout_new.argtypes = [C.POINTER(ci)]
        L.qo_layout_free.argtypes = [vp]
        L.qo_vol.argtypes = [vp]
        L.qo_index.argtypes = [vp, C.POINTER(ci)]
        L.qo_coord.argtypes = [vp, ci, C.POINTER(ci)]
        L.qo_neighbor.argtypes = [vp, ci, ci, ci]
        L.qo_rngfield_new.restype = vp
        L.qo_rngfield_new.argtypes = [vp, ci, C.c_uint64]
        L.qo_rngfield_free.argtypes = [vp]
        L.qo_milc6_test.argtypes = [C.c_uint32, C.c_uint32, ci, vp, vp]
        L.qo_mrg32k3a_test.argtypes = [C.c_uint64, C.c_uint64, ci, vp]
        for f in ("qo_vector_gaussian", "qo_gauge_gaussian", "qo_gauge_random", "qo_gauge_random_tah"):
            getattr(L, f).argtypes = [vp, vp, vp]
        L.qo_gauge_warm.argtypes = [vp, vp, cd, vp]
        L.qo_gauge_unit.argtypes = [vp, vp]
        for f in ("qo_projectU", "qo_projectSU", "qo_projectTAH", "qo_exp"):
            getattr(L, f).argtypes = [vp, vp]
        L.qo_setBC.argtypes = [vp, vp]
        L.qo_stagPhase.argtypes = [vp, vp, C.POINTER(ci)]
        L.qo_plaq.argtypes = [vp, vp, vp]
        L.qo_s4_gauge.argtypes = [vp, vp, vp]
        L.qo_gauge_force.argtypes = [vp, vp, vp]
        L.qo_gauge_deriv.argtypes = [vp, vp, vp, cd]
        L.qo_wflow.argtypes = [vp, vp, ci, cd]
        L.qo_norm2.restype = cd
        L.qo_norm2.argtypes = [vp, vp, ci]
        L.qo_redot.restype = cd
        L.qo_redot.argtypes = [vp, vp, vp, ci]
        L.qo_stagD2.argtypes = [vp, vp, vp, vp, vp, ci, cd, cd]
        L.qo_stagD.argtypes = [vp, vp, vp, vp, vp, ci, cd, cd, cd]
        L.qo_D.argtypes = [vp, vp, vp, vp, vp, cd]
        L.qo_Ddag.argtypes = [vp, vp, vp, vp, vp, cd]
        L.qo_stagD2xx.argtypes = [vp, vp, vp, vp, vp, cd, ci]
        L.qo_eoReconstruct.argtypes = [vp, vp, vp, vp, vp, cd]
        L.qo_eoReduce.argtypes = [vp, vp, vp, vp, vp, cd]
        L.qo_solveXX.argtypes = [vp, vp, vp, vp, vp, cd, cd, ci, ci, vp, ci, vp]
        L.qo_solve.argtypes = [vp, vp, vp, vp, vp, cd, cd, ci, vp]
        L.qo_fat7.argtypes = [vp, vp, vp, vp, vp, vp, cd]
        L.qo_hisq_smear.argtypes = [vp, vp, vp, vp]
        L.qo_nhyp_smear.argtypes = [vp, vp, vp, cd, cd, cd]
        L.qo_nhyp_force.argtypes = [vp, vp, vp, vp, vp, cd, cd, cd]
        L.qo_projectUderiv.argtypes = [vp, vp, vp, vp]
        L.qo_force_projTAH.argtypes = [vp, vp, vp, ci]
        L.qo_gauge_exp_update.argtypes = [vp, vp, vp, cd]
        L.qo_gauge_projectSU.argtypes = [vp, vp]
        L.qo_fat7_deriv.argtypes = [vp, vp, vp, vp, vp, vp, cd]
        L.qo_hisq_force.argtypes = [vp, vp, vp, vp, vp]
        L.qo_field_uniform.argtypes = [vp, vp, ci, vp, ci]
        L.qo_stag_outer.argtypes = [vp, vp, vp, cd, cd, ci]
        L.qo_stag_outer_hop.argtypes = [vp, vp, vp, cd, cd, ci, ci]
        L.qo_solve_prev.argtypes = [vp, vp, vp, vp, vp, cd, cd, ci, ci, vp]
        L.qo_solveXX_multi.argtypes = [vp, vp, vp, vp, vp, vp, ci, cd, ci, ci, vp, ci]
        L.qo_solve_multi.argtypes = [vp, vp, vp, vp, vp, vp, ci, cd, ci, vp]
        L.qo_gauge_action.restype = cd
        L.qo_gauge_action.argtypes = [vp, vp, cd, cd, ci]
        L.qo_gauge_force_general.argtypes = [vp, vp, vp, cd, cd, ci]
        L.qo_wflow_general.argtypes = [vp, vp, ci, cd, cd, cd, ci]
        L.qo_flow_EQ.argtypes = [vp, vp, ci, vp]
        L.qo_wline.argtypes = [vp, vp, C.POINTER(ci), ci, vp]
        L.qo_set_num_threads.argtypes = [ci]
        L.qo_set_num_threads(_cpu_share())
        _lib = L
    return _lib


def _cpu_share():
    """Threads the oracle may use: OMP_NUM_THREADS if set, else the cgroup CPU quota / affinity
    (a GPU box exposes 128 hardware threads but grants ~16 CPUs; 128 spinning OpenMP threads on a
    16-CPU share are orders of magnitude slower than 16)."""
    env = os.environ.get("OMP_NUM_THREADS")
    if env:
        try:
            return max(1, int(env))
        except ValueError:
            pass
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 16))


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


RNG_MILC6, RNG_MRG32K3A = 0, 1
EVEN, ODD, ALL = 0, 1, 2


class Layout:
    """V=1 MILC even-odd layout (src/layout/qlayout.nim:110-131)."""

    def __init__(self, lat):
        self.lat = [int(v) for v in lat]
        self._h = C.c_void_p(lib().qo_layout_new((C.c_int * 4)(*self.lat)))
        self.vol = int(np.prod(self.lat))

    def __del__(self):
        try:
            lib().qo_layout_free(self._h)
        except Exception:
            pass

    def index(self, x):
        return lib().qo_index(self._h, (C.c_int * 4)(*x))

    def coord(self, idx):
        c = (C.c_int * 4)()
        lib().qo_coord(self._h, idx, c)
        return list(c)

    def neighbor(self, idx, mu, ln):
        return lib().qo_neighbor(self._h, idx, mu, ln)

    def new_vector(self):
        return np.zeros((self.vol, 3, 2))

    def new_gauge(self):
        return np.zeros((self.vol, 4, 3, 3, 2))


class RngField:
    def __init__(self, lo, kind=RNG_MILC6, seed=17 ** 7):
        self.lo = lo
        self._h = C.c_void_p(lib().qo_rngfield_new(lo._h, kind, seed))

    def __del__(self):
        try:
            lib().qo_rngfield_free(self._h)
        except Exception:
            pass


def gauge_random(lo, rf=None, seed=17 ** 7):
    rf = rf or RngField(lo, RNG_MILC6, seed)
    g = lo.new_gauge()
    lib().qo_gauge_random(lo._h, rf._h, _p(g))
    return g


def gauge_warm(lo, s, rf):
    g = lo.new_gauge()
    lib().qo_gauge_warm(lo._h, rf._h, s, _p(g))
    return g


def gauge_unit(lo):
    g = lo.new_gauge()
    lib().qo_gauge_unit(lo._h, _p(g))
    return g


def gauge_random_tah(lo, rf):
    g = lo.new_gauge()
    lib().qo_gauge_random_tah(lo._h, rf._h, _p(g))
    return g


def vector_gaussian(lo, rf):
    v = lo.new_vector()
    lib().qo_vector_gaussian(lo._h, rf._h, _p(v))
    return v


def setBC(lo, g):
    lib().qo_setBC(lo._h, _p(g))


def stagPhase(lo, g, phases=(8, 9, 11, 0)):
    lib().qo_stagPhase(lo._h, _p(g), (C.c_int * 4)(*phases))


def rephase(lo, g):
    """Staggered.rephase (stagD.nim:72-80): setBC then stagPhase."""
    setBC(lo, g)
    stagPhase(lo, g)


def plaq(lo, g):
    out = np.zeros(6)
    lib().qo_plaq(lo._h, _p(g), _p(out))
    return out


def s4_gauge(lo, g):
    """g.s4_gauge() (stagg_pv_hmc/staghmc_spv_meas.nim:25-65): peo[dir][even/odd] as a (4, 2) array"""
    out = np.zeros(8)
    lib().qo_s4_gauge(lo._h, _p(g), _p(out))
    return out.reshape(4, 2)


def gauge_force(lo, g):
    f = lo.new_gauge()
    lib().qo_gauge_force(lo._h, _p(g), _p(f))
    return f


def gauge_deriv(lo, g, cplaq=1.0):
    f = lo.new_gauge()
    lib().qo_gauge_deriv(lo._h, _p(g), _p(f), cplaq)
    return f


def wflow(lo, g, nsteps, eps):
    lib().qo_wflow(lo._h, _p(g), nsteps, eps)


def gauge_action(lo, g, cplaq=1.0, c2=0.0, kind=0):
    return lib().qo_gauge_action(lo._h, _p(g), cplaq, c2, kind)


def gauge_force_general(lo, g, cplaq=1.0, c2=0.0, kind=0):
    f = lo.new_gauge()
    lib().qo_gauge_force_general(lo._h, _p(g), _p(f), cplaq, c2, kind)
    return f


def wflow_general(lo, g, nsteps, eps, cplaq=1.0, c2=0.0, kind=0):
    lib().qo_wflow_general(lo._h, _p(g), nsteps, eps, cplaq, c2, kind)


def flow_EQ(lo, g, loop=1):
    out = np.zeros(3)
    lib().qo_flow_EQ(lo._h, _p(g), loop, _p(out))
    return out


def wline(lo, g, path):
    out = np.zeros(2)
    lib().qo_wline(lo._h, _p(g), (C.c_int * len(path))(*path), len(path), _p(out))
    return complex(out[0], out[1])


def su3_fn(name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    r = np.zeros_like(x)
    getattr(lib(), name)(_p(r), _p(x))
    return r


def norm2(lo, x, parity=ALL):
    return lib().qo_norm2(lo._h, _p(x), parity)


def redot(lo, x, y, parity=ALL):
    return lib().qo_redot(lo._h, _p(x), _p(y), parity)


def stagD2(lo, fat, lng, r, x, parity, a, b):
    lib().qo_stagD2(lo._h, _p(fat), _p(lng), _p(r), _p(x), parity, a, b)


def stagD(lo, fat, lng, r, x, parity, m, sc=1.0, a=0.0):
    lib().qo_stagD(lo._h, _p(fat), _p(lng), _p(r), _p(x), parity, m, sc, a)


def D(lo, fat, lng, x, m):
    r = np.zeros_like(x)
    lib().qo_D(lo._h, _p(fat), _p(lng), _p(r), _p(x), m)
    return r


def Ddag(lo, fat, lng, x, m):
    r = np.zeros_like(x)
    lib().qo_Ddag(lo._h, _p(fat), _p(lng), _p(r), _p(x), m)
    return r


def stagD2xx(lo, fat, lng, x, m2, par_even=True):
    r = np.zeros_like(x)
    lib().qo_stagD2xx(lo._h, _p(fat), _p(lng), _p(r), _p(x), m2, 1 if par_even else 0)
    return r


def eoReduce(lo, fat, lng, r, b, m):
    lib().qo_eoReduce(lo._h, _p(fat), _p(lng), _p(r), _p(b), m)


def eoReconstruct(lo, fat, lng, r, b, m):
    lib().qo_eoReconstruct(lo._h, _p(fat), _p(lng), _p(r), _p(b), m)


def solveXX(lo, fat, lng, b, m, r2req, maxits, par_even=True, histcap=0):
    x = np.zeros_like(b)
    hist = np.zeros(max(histcap, 1))
    fin = C.c_double(0)
    its = lib().qo_solveXX(lo._h, _p(fat), _p(lng), _p(x), _p(b), m, r2req, maxits,
                           1 if par_even else 0, _p(hist), histcap, C.byref(fin))
    return x, its, fin.value, hist[: min(histcap, its + 1)]


def solveXX_ext(lo, fat, lng, b, m, r2req, maxits, par_even=True, histcap=0):
    """the same CG in binary128 arithmetic throughout (oracle/qex_oracle_ext.inc): (iterations, residual history) -- the yardstick"""
    hist = np.zeros(max(histcap, 1))
    fin = C.c_double(0)
    L = lib()
    L.qo_solveXX_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int,
                                 C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    its = L.qo_solveXX_ext(lo._h, _p(fat), _p(lng), None, _p(b), m, r2req, maxits, 1 if par_even else 0, _p(hist), histcap, C.byref(fin))
    return its, hist[: min(histcap, its + 1)]


def solveXX_multi_ext(lo, fat, lng, b, shifts, r2req, maxits, par_even=True, histcap=0):
    sh = np.array(shifts, dtype=np.float64)
    hist = np.zeros(max(histcap, 1))
    L = lib()
    L.qo_solveXX_multi_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int]
    its = L.qo_solveXX_multi_ext(lo._h, _p(fat), _p(lng), _p(b), _p(sh), len(sh), r2req, maxits, 1 if par_even else 0, _p(hist), histcap)
    return its, hist[: min(histcap, its + 1)]


def solve(lo, fat, lng, b, m, r2req, maxits):
    x = np.zeros_like(b)
    fin = C.c_double(0)
    its = lib().qo_solve(lo._h, _p(fat), _p(lng), _p(x), _p(b), m, r2req, maxits, C.byref(fin))
    return x, its, fin.value


def fat7(lo, g, coef, naik=0.0, g_long=None):
    """makeImpLinks (fat7l.nim:77-161); coef = (oneLink, threeStaple, fiveStaple, sevenStaple, lepage)."""
    fl, ll = lo.new_gauge(), lo.new_gauge()
    c = np.array(coef, dtype=np.float64)
    lib().qo_fat7(lo._h, _p(fl), _p(g), _p(c), _p(ll), _p(g if g_long is None else g_long), naik)
    return fl, ll


def hisq_smear(lo, g):
    fl, ll = lo.new_gauge(), lo.new_gauge()
    lib().qo_hisq_smear(lo._h, _p(g), _p(fl), _p(ll))
    return fl, ll


def nhyp_smear(lo, g, a1=0.4, a2=0.5, a3=0.5):
    fl = lo.new_gauge()
    lib().qo_nhyp_smear(lo._h, _p(g), _p(fl), a1, a2, a3)
    return fl


def gauge_deriv_general(lo, g, cplaq, c2, kind):
    """gaugeForceCust (kind 0, plaq+rect) / forceACust (kind 1, plaq+adjplaq): the derivative, no projection"""
    f = lo.new_gauge()
    fn = lib().qo_gauge_deriv_rect if kind == 0 else lib().qo_gauge_deriv_adj
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    fn(lo._h, _p(g), _p(f), cplaq, c2)
    return f


def nhyp_force(lo, g, chain, a1=0.4, a2=0.5, a3=0.5):
    """smearGetForce(g) then smearedForce(f, chain): returns (smeared links, f)."""
    fl, f = lo.new_gauge(), lo.new_gauge()
    lib().qo_nhyp_force(lo._h, _p(g), _p(fl), _p(f), _p(np.ascontiguousarray(chain)), a1, a2, a3)
    return fl, f


def projectUderiv(x, chain):
    """projectUderiv(r, x, c) (matrixFunctions.nim:353-357): u = projectU(x) first."""
    x, chain = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(chain, dtype=np.float64)
    u, r = np.zeros_like(x), np.zeros_like(x)
    lib().qo_projectU(_p(u), _p(x))
    lib().qo_projectUderiv(_p(r), _p(u), _p(x), _p(chain))
    return r


def fat7_deriv(lo, g, cfl, coef, cll=None, naik=0.0):
    """d/dU^+ of sum Re tr(cfl^+ fl) + sum Re tr(cll^+ ll), (fl, ll) = fat7(g, coef, naik)"""
    d = lo.new_gauge()
    lib().qo_fat7_deriv(lo._h, _p(d), _p(g), _p(np.ascontiguousarray(cfl)), (C.c_double * 5)(*coef),
                        _p(np.ascontiguousarray(cll)) if cll is not None else None, float(naik))
    return d


def hisq_force(lo, g, dsdsu, dsdsul):
    """HisqCoefs.smearGetForce(...)'s smearedForce(dsdu, dsdsu, dsdsul) (hisqsmear.nim:55-90)"""
    f = lo.new_gauge()
    lib().qo_hisq_force(lo._h, _p(g), _p(np.ascontiguousarray(dsdsu)), _p(np.ascontiguousarray(dsdsul)), _p(f))
    return f


def gauge_exp_update(lo, g, p, t):
    """g := exp(t p) g (mdt, staghmc_sh.nim:429-435)"""
    lib().qo_gauge_exp_update(lo._h, _p(g), _p(p), float(t))


def gauge_projectSU(lo, g):
    lib().qo_gauge_projectSU(lo._h, _p(g))


def vector_u1(lo, rf):
    """ftmp.u1 r (distributionUtils.nim:182-211): each colour component exp(2 pi i u), u = r.uniform (float32)"""
    u = np.zeros((lo.vol, 3))
    lib().qo_field_uniform(lo._h, rf._h, 3, _p(u), 0)
    n = 2.0 * np.pi * u
    return np.stack([np.cos(n), np.sin(n)], axis=-1)


def force_projTAH(lo, f, g, adj=False):
    lib().qo_force_projTAH(lo._h, _p(f), _p(g), 1 if adj else 0)


def stag_outer(lo, f, x, scale_even, scale_odd, accumulate, hop=1):
    lib().qo_stag_outer_hop(lo._h, _p(f), _p(x), scale_even, scale_odd, 1 if accumulate else 0, int(hop))


def solve_prev(lo, fat, lng, x0, b, m, r2req, maxits):
    """Staggered.solve with sp.usePrevSoln = true (stagSolve.nim:234-243)."""
    x = x0.copy()
    fin = C.c_double(0)
    its = lib().qo_solve_prev(lo._h, _p(fat), _p(lng), _p(x), _p(b), m, r2req, maxits, 1, C.byref(fin))
    return x, its, fin.value


def _pp(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def solveXX_multi(lo, fat, lng, b, shifts, r2req, maxits, par_even=True, histcap=0):
    n = len(shifts)
    xs = [np.zeros_like(b) for _ in range(n)]
    sh = np.array(shifts, dtype=np.float64)
    hist = np.zeros(max(histcap, 1))
    its = lib().qo_solveXX_multi(lo._h, _p(fat), _p(lng), _pp(xs), _p(b), _p(sh), n, r2req, maxits,
                                 1 if par_even else 0, _p(hist), histcap)
    return xs, its, hist[: min(histcap, its + 1)]


def solve_multi(lo, fat, lng, b, masses, r2req, maxits):
    n = len(masses)
    xs = [np.zeros_like(b) for _ in range(n)]
    ms = np.array(masses, dtype=np.float64)
    fin = C.c_double(0)
    its = lib().qo_solve_multi(lo._h, _p(fat), _p(lng), _pp(xs), _p(b), _p(ms), n, r2req, maxits, C.byref(fin))
    return xs, its, fin.value


def milc6_stream(seed, index, n):
    u = np.zeros(n, dtype=np.float32)
    g = np.zeros(n)
    lib().qo_milc6_test(seed & 0xFFFFFFFF, index, n, u.ctypes.data_as(C.c_void_p), _p(g))
    return u, g


def mrg32k3a_uniforms(seed, index, n):
    u = np.zeros(n)
    lib().qo_mrg32k3a_test(seed, index, n, _p(u))
    return u


def num_threads():
    return lib().qo_num_threads()
