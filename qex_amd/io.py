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


# ---- any other field: Writer.write / Reader.read (src/io/writerQiolite.nim:96-187, src/io/readerQiolite.nim:120-239) ----
def fileMetadata(fn):
    """(Reader.fileMetadata, Reader.recordMetadata) of the file's first record"""
    fl, rl = C.c_int(0), C.c_int(0)
    check(lib().qexhip_io_metadata(str(fn).encode(), None, 0, None, 0, C.byref(fl), C.byref(rl)))
    fb, rb = C.create_string_buffer(max(fl.value, 1)), C.create_string_buffer(max(rl.value, 1))
    check(lib().qexhip_io_metadata(str(fn).encode(), fb, fl.value, rb, rl.value, None, None))
    return fb.value.decode(), rb.value.decode()


def writeField(f, lat, fn, filemd=None, recordmd=None, datatype=None, colors=3):
    """One record holding the field f[vol, ...] (float64 -> precision D, float32 -> F; even-odd site order in memory)."""
    if f.dtype not in (np.float64, np.float32) or not f.flags["C_CONTIGUOUS"] or f.shape[0] != int(np.prod(lat)):
        raise ValueError("f must be a C-contiguous float32/float64 array [vol, ...] of this lattice")
    word = f.dtype.itemsize
    site = int(np.prod(f.shape[1:], dtype=np.int64)) * word
    name = datatype or "QDP_%s_%dx%d" % ("D" if word == 8 else "F", site // word, word)
    check(lib().qexhip_io_write_field(str(fn).encode(), (C.c_int * 4)(*lat), f.ctypes.data_as(C.c_void_p), site, word,
                                      name.encode(), b"D" if word == 8 else b"F", colors, 1,
                                      filemd.encode() if filemd is not None else None,
                                      recordmd.encode() if recordmd is not None else None))


def readField(fn, lat, site_shape, dtype=np.float64):
    """Returns (f[vol, *site_shape], datatype); QexHipError if the record's site size or checksum does not fit."""
    f = np.zeros((int(np.prod(lat)),) + tuple(site_shape), dtype=dtype)
    word = f.dtype.itemsize
    dt = C.create_string_buffer(64)
    check(lib().qexhip_io_read_field(str(fn).encode(), (C.c_int * 4)(*lat), f.ctypes.data_as(C.c_void_p),
                                     int(np.prod(site_shape, dtype=np.int64)) * word, word, dt))
    return f, dt.value.decode()
