"""CPU baseline of bench.py: QEX's CPU hot path restated in its own data layout (AoSoA, V = 8 sites per SIMD vector,
even-odd blocks), C++ -O3 -march=native -fopenmp.  Test infrastructure: only tests/ and
bench.py's cpu_baseline leg may use it.  Built for the CPU it runs on (the .so carries a stamp of the machine it was
compiled on and is rebuilt when that changes: -march=native code must not travel between hosts)."""
import ctypes as C
import os
import platform
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libqexcpusimd.so")
_STAMP = _SO + ".stamp"
FLAGS = ["-O3", "-march=native", "-fopenmp", "-fPIC", "-std=c++17", "-shared"]
_lib = None


def _cpu_id():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                flags = ""
                for l2 in open("/proc/cpuinfo"):
                    if l2.startswith("flags"):
                        flags = "avx512f" if " avx512f" in l2 else ("avx2" if " avx2" in l2 else "sse")
                        break
                return ln.split(":", 1)[1].strip() + " / " + flags
    except OSError:
        pass
    return platform.processor() or "unknown"


def build(force=False):
    src = os.path.join(_HERE, "qex_cpu_simd.cpp")
    want = _cpu_id() + " | " + " ".join(FLAGS)
    have = open(_STAMP).read() if os.path.exists(_STAMP) else ""
    if force or not os.path.exists(_SO) or have != want or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["g++"] + FLAGS + ["-o", _SO, src])
        open(_STAMP, "w").write(want)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        L.qcs_new.restype = vp
        L.qcs_new.argtypes = [C.POINTER(ci), vp]
        L.qcs_free.argtypes = [vp]
        L.qcs_site_map.argtypes = [vp, vp]
        L.qcs_stagD2.argtypes = [vp, vp, vp, ci, cd]
        L.qcs_solveXX.restype = ci
        L.qcs_solveXX.argtypes = [vp, vp, vp, cd, cd, ci, ci, vp, ci, C.POINTER(cd)]
        L.qcs_num_threads.restype = ci
        L.qcs_set_num_threads.argtypes = [ci]
        _lib = L
        L.qcs_set_num_threads(_cpu_share())
    return _lib


def _cpu_share():
    env = os.environ.get("OMP_NUM_THREADS")
    if env:
        return int(env)
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


class Lattice:
    def __init__(self, lat, g):
        self.lat = [int(v) for v in lat]
        g = np.ascontiguousarray(g, dtype=np.float64)
        self._h = lib().qcs_new((C.c_int * 4)(*self.lat), g.ctypes.data_as(C.c_void_p))
        if not self._h:
            raise ValueError("cpu_simd needs x even and y, z, t extents divisible by 4 (inner geometry 1x2x2x2)")

    def __del__(self):
        try:
            if self._h:
                lib().qcs_free(self._h)
        except Exception:
            pass

    def site_map(self):
        """simd_of_v1[V=1 even-odd index] = outer * 8 + lane, the map this baseline stores its fields by"""
        m = np.zeros(int(np.prod(self.lat)), dtype=np.int32)
        lib().qcs_site_map(self._h, m.ctypes.data_as(C.c_void_p))
        return m

    def stagD2(self, r, x, parity, b=0.0):
        lib().qcs_stagD2(self._h, r.ctypes.data_as(C.c_void_p), np.ascontiguousarray(x).ctypes.data_as(C.c_void_p), int(parity), float(b))

    def solveXX(self, b, mass, r2req, maxits, par_even=True, histcap=0):
        x = np.zeros_like(b)
        hist = np.zeros(max(histcap, 1))
        secs = C.c_double(0)
        its = lib().qcs_solveXX(self._h, x.ctypes.data_as(C.c_void_p), np.ascontiguousarray(b).ctypes.data_as(C.c_void_p), float(mass),
                                float(r2req), int(maxits), 1 if par_even else 0, hist.ctypes.data_as(C.c_void_p), int(histcap), C.byref(secs))
        return x, its, hist[: min(histcap, its + 1)], secs.value


def bench_cg(lat, g, b, mass, budget_s):
    """bounded sample for bench.py: calibrate on 3 iterations, then ~budget_s worth of CG iterations (timed inside the
    library around the CG loop only: the AoSoA conversion of the inputs is setup)"""
    Lt = Lattice(lat, g)
    _, its, _, s = Lt.solveXX(b, mass, 0.0, 3)
    per = max(s / 3.0, 1e-6)
    n = int(max(5, min(2000, budget_s / per)))
    _, its, _, s = Lt.solveXX(b, mass, 0.0, n)
    vh = int(np.prod(lat)) // 2
    return {
        "value": round(1212 * vh * its / s / 1e9, 3), "unit": "GFLOP/s", "cores": lib().qcs_num_threads(), "kind": "port",
        "cg_iters_per_s": round(its / s, 3),
        "sample": "%d CG iterations of the same %dx%dx%dx%d workload (same links/source); QEX's layout restated: AoSoA, 8 sites per "
                  "SIMD vector, even-odd blocks; g++ %s on %s" % (its, lat[0], lat[1], lat[2], lat[3], " ".join(FLAGS[:3]), _cpu_id()),
    }
