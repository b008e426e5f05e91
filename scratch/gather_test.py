import ctypes as C, sys
sys.path.insert(0, '.')
import qex_amd as q
from qex_amd._lib import tune_lib
T = tune_lib()
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat); q.gaugeSet(ctx, g)
out = C.c_double(0)
print("48 matrices per 64-site tile, 32^4: 7.25 GB through L1 per launch")
for rnd in range(2):
    for nw, wgpc in ((4, 1), (8, 1), (4, 2), (2, 4), (4, 4), (8, 2)):
        for depth in (1, 2, 4, 6):
            rc = T.qexhip_tune_gather(ctx._h, nw, wgpc, depth, 5, C.byref(out))
            print("round %d  waves/WG %d  WG/CU %d  depth %d: %7.1f us  %.1f TB/s  %.1f GB/s per CU" % (rnd, nw, wgpc, depth, out.value, 7.25e3 / out.value, 7.25e6 / out.value / 256) if rc == 0 else "rc %d" % rc, flush=True)
