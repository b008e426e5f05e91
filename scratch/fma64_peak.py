# fp64 vector throughput this chip actually holds (spec 78.6 TFLOP/s at 2.4 GHz): plain FMA chains and the exp squaring chain
import sys, ctypes as C; sys.path.insert(0, '.')
import qex_amd as q
from qex_amd._lib import tune_lib
L = tune_lib()   # libqexhip_tune.so: measurement scaffolding, not the product library
L.qexhip_tune_fma64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
ctx = q.Context([8, 8, 8, 8])
out = C.c_double(0)
for chains in (2, 4, 8, 16):
    for wps in (1, 2, 4, 8):
        iters = 4000000 // (chains * wps)
        L.qexhip_tune_fma64(ctx._h, 0, chains, wps, iters, C.byref(out))
        print("fma64 chains/lane %2d waves/SIMD %d: %6.2f TFLOP/s" % (chains, wps, out.value), flush=True)
for wps in (1, 2, 3, 4):
    L.qexhip_tune_fma64(ctx._h, 1, 0, wps, 40000 // wps, C.byref(out))
    print("exp squaring chain (3x3 complex r <- r(r+2)) waves/SIMD %d: %6.2f TFLOP/s" % (wps, out.value), flush=True)
