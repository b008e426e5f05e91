"""Round 6, VERDICT item 7, second (decisive) rung: the Wilson-flow stage's operand gather stream alone (48 matrices per 64-site tile
into registers, nothing else: libqexhip_tune.so k_gather_test) with whole matrices against rows 0,1 only."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qex_amd as q
from qex_amd import _lib
lat = [32, 32, 32, 32]
ctx = q.Context(lat)
rf = q.RngField(lat, q.RngMilc6, 987654321)
q.gaugeSet(ctx, rf.warm(0.5))
T = _lib.tune_lib()
for nw, wgpc, depth in ((8, 1, 2), (8, 1, 4), (4, 2, 4), (8, 2, 2), (6, 2, 4)):
    row = []
    for rows in (3, 2, 3, 2):
        us = C.c_double(0)
        rc = T.qexhip_tune_gather_rows(ctx._h, nw, wgpc, depth, rows, 10, C.byref(us))
        assert rc == 0, rc
        row.append((rows, round(us.value, 1)))
    print("waves/wg %d, wg/CU %d, matrices in flight %d:" % (nw, wgpc, depth), " ".join("%s %.1f us" % ("whole" if r == 3 else "rows01", v) for r, v in row), flush=True)
