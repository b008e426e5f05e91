import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L = q.lib()
lat = [8, 8, 8, 8]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
def flow(ring, n=1):
    ctx.set_option("flow_ring", ring); q.gaugeSet(ctx, g)
    check(L.qexhip_wflow(ctx._h, n, 0.01))
    out = np.zeros_like(g); check(L.qexhip_gauge_get(ctx._h, out.ctypes.data_as(C.c_void_p))); return out
a, b = flow(0), flow(1)
d = np.abs(a - b).reshape(2, -1, 4, 18).max(axis=3)     # [parity][c][mu]
print("max", d.max())
for p in range(2):
    for mu in range(4):
        print("parity", p, "mu", mu, "max %.2e" % d[p, :, mu].max(), "frac bad %.3f" % (d[p, :, mu] > 1e-10).mean())
bad = np.argwhere(d > 1e-10)
print("first bad (parity, c, mu):", bad[:20].tolist())
tiles = sorted(set((int(x[0]), int(x[1]) // 64) for x in bad))
print("bad tiles:", tiles[:64], len(tiles), "of", 2 * d.shape[1] // 64)
