"""Per-site random number fields and the configuration generators on them: the host mirror of
newRNGField / gaussian / uniform / u1 (src/rng/distributionUtils.nim) and random / warm / randomTAH
(src/gauge/gaugeUtils.nim:1348-1446) over qexhip_rng_* (csrc/rng.hip).  Host only, no GPU needed."""
import ctypes as C

import numpy as np

from ._lib import check, lib

RngMilc6, MRG32k3a = 0, 1


class RngField:
    """lo.newRNGField(RngMilc6 | MRG32k3a, seed); for a t-sharded rank pass the global lattice and t_offset."""

    def __init__(self, lat, kind=RngMilc6, seed=987654321, glat=None, t_offset=0):
        self.lat = [int(v) for v in lat]
        self.vol = int(np.prod(self.lat))
        self._h = C.c_void_p()
        gl = (C.c_int * 4)(*[int(v) for v in glat]) if glat is not None else None
        check(lib().qexhip_rng_new(C.byref(self._h), int(kind), C.c_ulonglong(int(seed) & (2 ** 64 - 1)), (C.c_int * 4)(*self.lat),
                                   gl, int(t_offset)))

    def __del__(self):
        try:
            if self._h:
                lib().qexhip_rng_free(self._h)
        except Exception:
            pass

    def _new(self, *shape):
        return np.zeros((self.vol,) + shape)

    def uniform(self, ncomp):
        v = self._new(ncomp)
        check(lib().qexhip_rng_uniform(self._h, int(ncomp), v.ctypes.data_as(C.c_void_p)))
        return v

    def gaussian_vector(self):
        v = self._new(3, 2)
        check(lib().qexhip_rng_gaussian_vector(self._h, v.ctypes.data_as(C.c_void_p)))
        return v

    def u1_vector(self):
        v = self._new(3, 2)
        check(lib().qexhip_rng_u1_vector(self._h, v.ctypes.data_as(C.c_void_p)))
        return v

    def randomTAH(self):
        p = self._new(4, 3, 3, 2)
        check(lib().qexhip_rng_random_tah(self._h, p.ctypes.data_as(C.c_void_p)))
        return p

    def random(self):
        g = self._new(4, 3, 3, 2)
        check(lib().qexhip_rng_gauge_random(self._h, g.ctypes.data_as(C.c_void_p)))
        return g

    def warm(self, s):
        g = self._new(4, 3, 3, 2)
        check(lib().qexhip_rng_gauge_warm(self._h, float(s), g.ctypes.data_as(C.c_void_p)))
        return g

    # ---- the same draws written straight into HBM (RngMilc6 only; the field's state advances exactly as on the host) ----
    def dev_gaussian_vector(self, ctx, field_id):
        """v.gaussian r into the resident colour vector `field_id` of ctx"""
        check(lib().qexhip_rng_dev_gaussian_vector(ctx._h, self._h, int(field_id)))

    def dev_u1_vector(self, ctx, field_id):
        check(lib().qexhip_rng_dev_u1_vector(ctx._h, self._h, int(field_id)))

    def dev_momenta(self, ctx):
        """p.randomTAH r into the resident MD momenta of ctx (qexhip_md_refresh_momenta)"""
        check(lib().qexhip_md_refresh_momenta(ctx._h, self._h))

    # ---- checkpoints: write_rng / read_rng of the fork (src/stagg_pv_hmc/staghmc_spv_rng.nim:135-182) ----
    def state(self):
        n = lib().qexhip_rng_state_words(self._h)
        w = np.zeros((self.vol, n), dtype=np.uint32)
        check(lib().qexhip_rng_get_state(self._h, w.ctypes.data_as(C.c_void_p)))
        return w

    def set_state(self, w):
        w = np.ascontiguousarray(w, dtype=np.uint32)
        if w.shape != (self.vol, lib().qexhip_rng_state_words(self._h)):
            raise ValueError("RNG state has the wrong shape")
        check(lib().qexhip_rng_set_state(self._h, w.ctypes.data_as(C.c_void_p)))

    def write(self, fn):
        """the generator field as one SciDAC record of typesize 36 (`writer.write(r.milc, recordMd)`)"""
        w = self.state()
        name = b"QDP_RngMilc6" if w.shape[1] == 9 else b"QDP_MRG32k3a"
        check(lib().qexhip_io_write_field(str(fn).encode(), (C.c_int * 4)(*self.lat), w.ctypes.data_as(C.c_void_p), 4 * w.shape[1], 4,
                                          name, b"F", 0, 1, None, b'<?xml version="1.0"?>\n<note>RNG field</note>\n'))

    def read(self, fn):
        w = np.zeros((self.vol, lib().qexhip_rng_state_words(self._h)), dtype=np.uint32)
        dt = C.create_string_buffer(64)
        check(lib().qexhip_io_read_field(str(fn).encode(), (C.c_int * 4)(*self.lat), w.ctypes.data_as(C.c_void_p), 4 * w.shape[1], 4, dt))
        self.set_state(w)
        return dt.value.decode()
