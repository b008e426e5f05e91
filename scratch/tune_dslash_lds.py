# Dslash sweep: the product kernel's form (variant 13: non-temporal loads/stores, fence per pair) against the LDS-DMA staged
# variant 20, interleaved in one process; correctness of variant 20 against variant 13 on the same fields.
import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from qex_amd._lib import tune_lib
L = tune_lib()   # libqexhip_tune.so: measurement scaffolding, not the product library
L.qexhip_tune_dslash.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_int,C.POINTER(C.c_double)]
lat=[32,32,32,32]
lo=q.Layout(lat)
rf=q.RngField(lat,q.RngMilc6,987654321)
g=rf.random(); q.rephase(lo,g)
ctx=q.Context(lat); s=q.newStag(ctx,g)
x=rf.gaussian_vector(); r=np.zeros_like(x); s.stagD2(r,x,"even",0,0)  # allocates + fills the work fields the variants run on
gb=1248*lo.vol/2/1e9
out=C.c_double(0)
names={13:"registers: nt loads + nt stores + fence per pair (= product kernel)",0:"registers: plain",20:"LDS-DMA staged links + vectors, double-buffered"}
for rnd in range(3):
    for v in (13,20,0):
        for swz in (0,1):
            rc=L.qexhip_tune_dslash(ctx._h,v,swz,50,C.byref(out))
            print(f"round {rnd} var {v:2d} swz {swz}: {out.value:8.2f} us  {gb/out.value*1e6:7.1f} GB/s  {gb/out.value*1e6/8000:.3f}  rc {rc}  {names[v]}",flush=True)
L.qexhip_tune_dslash_norm2.argtypes=[C.c_void_p,C.POINTER(C.c_double)]
n2={}
for v in (13,20,0):
    L.qexhip_tune_dslash(ctx._h,v,0,1,C.byref(out)); L.qexhip_tune_dslash_norm2(ctx._h,C.byref(out)); n2[v]=out.value
print("|out|^2 per variant:",n2," relative spread %.2e"%(max(n2.values())/min(n2.values())-1))
