"""SciDAC/LIME gauge files: the host mirror of loadGauge / saveGauge / getFileLattice
(src/gauge/gaugeUtils.nim:87-122, src/io/readerQiolite.nim:11-17).  Thin ctypes layer over
qexhip_io_* (csrc/scidac_io.cpp); fields are the library's host format [vol][4][3][3][2]."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def getFileLattice(fn):
    lat, prec, has = (C.c_int * 4)(), C.create_string_buffer(2), C.c_int(0)
    check(lib().qexhip_io_gauge_info(str(fn).encode(), lat, prec, C.byref(has)))
    return list(lat)


def gaugeFileInfo(fn):
    lat, prec, has = (C.c_int * 4)(), C.create_string_buffer(2), C.c_int(0)
    check(lib().qexhip_io_gauge_info(str(fn).encode(), lat, prec, C.byref(has)))
    return dict(lattice=list(lat), precision=prec.raw[:1].decode(), checksums=bool(has.value))


def loadGauge(fn, lat=None):
    """Returns (g, (suma, sumb)); raises QexHipError on a checksum mismatch."""
    lat = list(lat) if lat is not None else getFileLattice(fn)
    g = np.zeros((int(np.prod(lat)), 4, 3, 3, 2))
    a, b = C.c_uint(0), C.c_uint(0)
    check(lib().qexhip_io_read_gauge(str(fn).encode(), (C.c_int * 4)(*lat), g.ctypes.data_as(C.c_void_p),
                                     C.byref(a), C.byref(b)))
    return g, (a.value, b.value)


def loadGaugeSlab(fn, glat, t0, nt):
    """one rank's slab t0 <= t < t0 + nt of the global configuration in fn (local even-odd order)"""
    lat = list(glat[:3]) + [int(nt)]
    g = np.zeros((int(np.prod(lat)), 4, 3, 3, 2))
    check(lib().qexhip_io_read_gauge_slab(str(fn).encode(), (C.c_int * 4)(*[int(v) for v in glat]), int(t0), int(nt),
                                          g.ctypes.data_as(C.c_void_p)))
    return g


def saveGauge(g, lat, fn, prec="D", filemd=None, recordmd=None):
    if g.dtype != np.float64 or not g.flags["C_CONTIGUOUS"] or g.size != int(np.prod(lat)) * 72:
        raise ValueError("g must be a C-contiguous float64 [vol][4][3][3][2] array of this lattice")
    check(lib().qexhip_io_write_gauge(str(fn).encode(), (C.c_int * 4)(*lat), g.ctypes.data_as(C.c_void_p),
                                      prec.encode()[:1], filemd.encode() if filemd else None,
                                      recordmd.encode() if recordmd else None))
