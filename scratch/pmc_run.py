# run under: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 scratch/pmc_run.py
import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np
import qex_amd as q
from qex_amd._lib import tune_lib
L = tune_lib()   # libqexhip_tune.so: measurement scaffolding, not the product library
L.qexhip_tune_stream.argtypes=[C.c_void_p,C.c_int,C.c_size_t,C.c_int,C.c_int,C.POINTER(C.c_double)]
lat=[32,32,32,32]
lo=q.Layout(lat)
if len(sys.argv)>1 and sys.argv[1]=='recon':
    g=q.synthetic_random_su3(lo); q.rephase(lo,g)     # SU(3) x signs -> compressed links
else:
    g=q.unit(lo); rng=np.random.default_rng(1); g+=0.1*rng.standard_normal(g.shape)   # not unitary -> 18-real links
ctx=q.Context(lat); s=q.newStag(ctx,g)
out=C.c_double(0)
for mode in (0,1,2,3):
    L.qexhip_tune_stream(ctx._h,mode,1024,2048,3,C.byref(out))   # 1 GiB known byte counts (5 launches each incl. warmup)
b=q.synthetic_gaussian_vector(lo)
bid=ctx.field_new(b); xid=ctx.field_new()
ctx.dev_solve_xx(xid,bid,0.1,0.0,10,True)
ctx.sync()
print("done", s.links_info())
