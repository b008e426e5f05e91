"""Host-side lattice layout: the V=1 even-odd site order QEX uses when it hands fields to an
external solver (src/quda/qudaWrapperImpl.nim:118,198-240; src/layout/qlayout.nim:110-131).

    lex = x0 + L0*(x1 + L1*(x2 + L2*x3));  idx = lex//2 + ((x0+x1+x2+x3)&1)*vol//2

Pure numpy host logic (index arithmetic only); no field arithmetic lives here.
"""
import numpy as np


class Layout:
    def __init__(self, lat):
        lat = [int(v) for v in lat]
        if len(lat) != 4 or any(v < 2 or v % 2 for v in lat):
            raise ValueError("lattice extents must be 4 even numbers >= 2")
        self.lat = lat
        self.physGeom = lat
        self.vol = int(np.prod(lat))
        self.nSites = self.vol
        self.nEven = self.vol // 2
        lex = np.arange(self.vol)
        x = np.empty((self.vol, 4), dtype=np.int64)
        r = lex.copy()
        for i in range(4):
            x[:, i] = r % lat[i]
            r //= lat[i]
        par = x.sum(axis=1) & 1
        idx = lex // 2 + par * (self.vol // 2)
        self.coords = np.empty_like(x)
        self.coords[idx] = x  # coords[idx] = (x0,x1,x2,x3)
        self._lex_of_idx = np.empty(self.vol, dtype=np.int64)
        self._lex_of_idx[idx] = lex
        self._idx_of_lex = idx

    def index(self, x):
        lex = 0
        for i in (3, 2, 1, 0):
            lex = lex * self.lat[i] + int(x[i]) % self.lat[i]
        return int(self._idx_of_lex[lex])

    def coord(self, idx):
        return [int(v) for v in self.coords[idx]]

    def ColorVector(self):
        return np.zeros((self.vol, 3, 2))

    def newGauge(self):
        return np.zeros((self.vol, 4, 3, 3, 2))

    # ---- sharding along t (rankGeom = [1,1,1,N], src/layout/layoutX.nim:80-92) ----
    def shard_indices(self, nranks, rank):
        """Global MILC indices of the sites of `rank`'s t-slab, in the slab's own MILC order."""
        T = self.lat[3]
        if T % nranks:
            raise ValueError("t extent not divisible by the number of ranks")
        lt = T // nranks
        if lt % 2:
            raise ValueError("local t extent must be even")
        loc = Layout(self.lat[:3] + [lt])
        gx = loc.coords.copy()
        gx[:, 3] += rank * lt
        lex = gx[:, 0] + self.lat[0] * (gx[:, 1] + self.lat[1] * (gx[:, 2] + self.lat[2] * gx[:, 3]))
        return loc, self._idx_of_lex[lex]
