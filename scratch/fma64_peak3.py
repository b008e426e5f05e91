import sys, ctypes as C, time; sys.path.insert(0, '.')
import qex_amd as q
from qex_amd._lib import tune_lib
L = tune_lib()   # libqexhip_tune.so: measurement scaffolding, not the product library
L.qexhip_tune_fma64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
ctx = q.Context([8, 8, 8, 8])
out = C.c_double(0)
for wps in (1, 2, 3, 4):
    L.qexhip_tune_fma64(ctx._h, 2, 0, wps, 4000 // wps, C.byref(out))
    print("m3_exp + product in a loop (count 23 x 216 flop), waves/SIMD %d: %6.2f TFLOP/s" % (wps, out.value), flush=True)
for wps in (2, 3, 4):
    L.qexhip_tune_fma64(ctx._h, 3, 0, wps, 64 // wps, C.byref(out))
    print("one m3_exp per wavefront, %d wavefronts per SIMD in all, %d resident: %6.2f TFLOP/s" % (64 // wps * wps, wps, out.value), flush=True)
