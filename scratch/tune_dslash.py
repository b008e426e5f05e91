import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from qex_amd._lib import tune_lib
L = tune_lib()   # libqexhip_tune.so: measurement scaffolding, not the product library
L.qexhip_tune_dslash.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_int,C.POINTER(C.c_double)]
L.qexhip_tune_stream.argtypes=[C.c_void_p,C.c_int,C.c_size_t,C.c_int,C.c_int,C.POINTER(C.c_double)]
lat=[32,32,32,32]
lo=q.Layout(lat)
g=q.unit(lo); rng=np.random.default_rng(1); g+=0.1*rng.standard_normal(g.shape)   # any full-range links
ctx=q.Context(lat); s=q.newStag(ctx,g)
x=q.synthetic_gaussian_vector(lo); r=np.zeros_like(x); s.stagD2(r,x,"even",0,0)  # allocates + fills work fields
names={0:"base",1:"nt-links",2:"fence-pair",3:"nt+fence-pair",4:"nt-store",5:"nt-links+nt-store",6:"BS128",7:"BS512",8:"BS64",9:"fence-dir",10:"nt+fence-dir",11:"minw4",12:"fence-pair minw3",13:"nt+fence+ntstore"}
gb=1248*lo.vol/2/1e9
out=C.c_double(0)
for rnd in range(3):
    for v in range(14):
        for swz in (1,0):
            L.qexhip_tune_dslash(ctx._h,v,swz,50,C.byref(out))
            print(f"round {rnd} var {v:2d} {names[v]:22s} swz {swz}: {out.value:8.2f} us  {gb/out.value*1e6:7.1f} GB/s  {gb/out.value*1e6/8000:.3f}",flush=True)
for mode in (0,1):
    for nb in (2048,8192,32768):
        L.qexhip_tune_stream(ctx._h,mode,1024,nb,20,C.byref(out)); print("stream mode",mode,"blocks",nb,f"{out.value:.0f} GB/s")
