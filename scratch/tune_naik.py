import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from qex_amd._lib import tune_lib
L = tune_lib()   # libqexhip_tune.so: measurement scaffolding, not the product library
L.qexhip_tune_dslash.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_int,C.POINTER(C.c_double)]
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.unit(lo); rng=np.random.default_rng(1); g+=0.1*rng.standard_normal(g.shape); g3=0.3*g
ctx=q.Context(lat); s=q.newStag3(ctx,g,g3)
x=q.synthetic_gaussian_vector(lo); r=np.zeros_like(x); s.stagD2(r,x,"even",0,0)
names={100:"nt+ntstore",101:"nt+ntstore+fence-pair",102:"nt+ntstore+fence-dir",103:"BS128",104:"BS512",105:"plain",106:"minw3"}
gb=2400*lo.vol/2/1e9; out=C.c_double(0)
for rnd in range(2):
    for v in range(100,107):
        for swz in (0,1):
            rc=L.qexhip_tune_dslash(ctx._h,v,swz,30,C.byref(out))
            print(f"round {rnd} var {v} {names[v]:24s} swz {swz}: {out.value:8.2f} us {gb/out.value*1e6:7.1f} GB/s {gb/out.value*1e6/8000:.3f}",flush=True)
