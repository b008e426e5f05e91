import sys; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
from oracle import oracle as o
for lat in ([8,8,8,8],[16,16,16,16]):
    lo=o.Layout(lat); ctx=q.Context(lat)
    for name,g in (("random",o.gauge_random(lo,seed=987654321)),("warm0.5",o.gauge_warm(lo,0.5,o.RngField(lo,o.RNG_MILC6,5))),("synthetic",q.synthetic_random_su3(q.Layout(lat)))):
        o.rephase(lo,g); s=q.newStag(ctx,g); print(lat,name,s.links_info(),flush=True)
        gc=g[...,0]+1j*g[...,1]
        uerr=np.abs(np.einsum('...ij,...kj->...ik',gc,gc.conj())-np.eye(3)).max()
        print("   max |UU^+ - 1|",uerr)
