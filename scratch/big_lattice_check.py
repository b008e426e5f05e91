# 48^3x96 on one GPU: plaquette, one Wilson-flow step, flow observables, HISQ links, nHYP closure + gauge force.
# Size-independent checks only (unitarity, plaquette monotonicity, algebra membership, D anti-Hermiticity).
import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[48,48,48,96]; lo=q.Layout(lat)
t=time.time(); g=q.synthetic_random_su3(lo, spread=0.25); print("config",round(time.time()-t,1),"s",flush=True)
ctx=q.Context(lat)
def cx(a): return a[...,0]+1j*a[...,1]
t=time.time(); p0=q.plaq(ctx,g); print("plaq",p0.sum(),round(time.time()-t,2),"s",flush=True)
t=time.time(); q.gaugeFlow(ctx,g,1,0.01); p1=q.plaq(ctx,g); print("flow step: plaq",p1.sum(),round(time.time()-t,2),"s",flush=True)
assert p1.sum()>p0.sum()
m=cx(g).reshape(-1,3,3)[::100003]; assert np.abs(np.einsum('nij,nkj->nik',m,m.conj())-np.eye(3)).max()<1e-12
t=time.time(); e=q.flowEQ(ctx,1); print("E,Q loop1",e,round(time.time()-t,2),"s",flush=True)
gp=g.copy(); q.rephase(lo,gp)
fl=np.zeros_like(g); ll=np.zeros_like(g)
t=time.time(); q.HisqCoefs().init().smear(ctx,gp,fl,ll); print("hisq",round(time.time()-t,2),"s", np.isfinite(fl).all(), np.isfinite(ll).all(),flush=True)
s=q.newStag3(ctx,fl,ll); print("links",s.links_info(),flush=True)
x=q.synthetic_gaussian_vector(lo,1); y=q.synthetic_gaussian_vector(lo,2)
Dx=np.zeros_like(x); Dy=np.zeros_like(x); s.D(Dx,x,0.0); s.D(Dy,y,0.0)
ah=abs(np.vdot(cx(y),cx(Dx))+np.vdot(cx(Dy),cx(x)))/np.sqrt((Dx*Dx).sum()*(y*y).sum()); print("HISQ D anti-hermiticity",ah,flush=True); assert ah<1e-12
del fl,ll,Dx,Dy
sg=np.zeros_like(g)
t=time.time(); sf=q.HypCoefs(0.4,0.5,0.5).smearGetForce(ctx,g,sg); print("nhyp prepare",round(time.time()-t,2),"s",flush=True)
m=cx(sg).reshape(-1,3,3)[::100003]; assert np.abs(np.einsum('nij,nkj->nik',m,m.conj())-np.eye(3)).max()<1e-12
f=np.zeros_like(g)
ctx.timers_enable(1); ctx.timers_reset()
t=time.time(); sf.gforce(f,plaq=6.0,adjplaq=-1.5); print("nhyp gforce",round(time.time()-t,2),"s; chain kernels",ctx.timer("nhyp_force"),flush=True)
fc=cx(f).reshape(-1,3,3)[::100003]
assert np.abs(fc+fc.conj().transpose(0,2,1)).max()<1e-10 and np.abs(np.trace(fc,axis1=1,axis2=2)).max()<1e-10
sf.release()
print("OK")
