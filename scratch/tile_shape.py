"""round-5 tile-shape experiment (round-4 verdict, Next 3): the Wilson-flow stage's 48-operand gather stream with the link field
read (a) in the product's tile shape (64 consecutive checkerboard sites = 128 sites = 4 x-rows), (b) in 8x4x4x1 bricks.
usage: python scratch/tile_shape.py [time|a|b]   (a / b: run ONE shape a few times, for a rocprofv3 --pmc pass)"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
mode = sys.argv[1] if len(sys.argv) > 1 else "time"
lat = [32, 32, 32, 32]
T = C.CDLL(os.path.join(os.path.dirname(q.LIB_PATH), "libqexhip_tune.so"))
for f in (T.qexhip_tune_gather, T.qexhip_tune_gather_brick):
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
q.plaq(ctx, g)                                   # links resident in the natural layout
us = C.c_double(0)
bytes_l1 = 48 * 144 * 64 * 2 * (np.prod(lat) // 128)
if mode == "time":
    print("48 matrices per 64-site tile position and parity, 32^4: %.2f GB through L1 per launch" % (bytes_l1 / 1e9))
    for rnd in range(2):
        for nw, wgpc, depth in ((4, 1, 1), (4, 1, 2), (4, 2, 1), (4, 2, 2), (4, 4, 1), (4, 4, 2), (8, 2, 1)):
            check(T.qexhip_tune_gather(ctx._h, nw, wgpc, depth, 20, C.byref(us))); a = us.value
            check(T.qexhip_tune_gather_brick(ctx._h, nw, wgpc, depth, 20, C.byref(us))); b = us.value
            print("round %d  waves/WG %d  WG/CU %d  depth %d:  rows %7.1f us   bricks %7.1f us   (%.2fx)" % (rnd, nw, wgpc, depth, a, b, a / b), flush=True)
else:
    f = T.qexhip_tune_gather if mode == "a" else T.qexhip_tune_gather_brick
    check(f(ctx._h, 4, 2, 1, 5, C.byref(us)))
    print(mode, us.value)
