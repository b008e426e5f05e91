# One rank's share of BASELINE configs[4] (nHYP stagg_pv_hmc force + HISQ Naik multi-shift CG, 48^3x96 on 8 GPUs),
# rehearsed on one GPU: local lattice 48^3x12 with forced ghost zones (every kernel in its sharded form, faces wrapped
# onto the rank itself through the one-rank exchange path).  Usage: config5_rehearsal.py [XxYxZxT] [halo 0|1]
import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[int(v) for v in (sys.argv[1].split('x') if len(sys.argv)>1 else [48,48,48,12])]
halo=int(sys.argv[2]) if len(sys.argv)>2 else 1
lo=q.Layout(lat)
spread=float(sys.argv[3]) if len(sys.argv)>3 else 0.3
g=q.synthetic_random_su3(lo, spread=spread) if spread>0 else q.RngField(lat,q.RngMilc6,987654321).warm(0.5)   # spread 0: QEX's warm(0.5) start
ctx=q.Context(lat)
if len(sys.argv)>4: ctx.set_option('recon',int(sys.argv[4]))
if halo: ctx.force_halo(True)
print("lattice",lat,"forced halo",halo,flush=True)
# --- nHYP closure + the two MD forces
fl=np.zeros_like(g); f=np.zeros_like(g)
hc=q.HypCoefs(0.4,0.5,0.5)
sf=hc.smearGetForce(ctx,g,fl)
ctx.timers_enable(1); ctx.timers_reset()
sf=hc.smearGetForce(ctx,g,fl); n,ms=ctx.timer("smear"); print("nHYP smear kernels %.2f ms"%ms,flush=True)
psis=[q.synthetic_gaussian_vector(lo,seed=5+k) for k in range(2)]
sf.gforce(f, plaq=1.0)
ctx.timers_reset(); sf.gforce(f, plaq=1.0); n,ms=ctx.timer("nhyp_force"); print("nHYP force chain %.2f ms"%ms,flush=True)
ctx.timers_reset(); sf.fforce(f, psis, [1.0,0.5]); n,ms=ctx.timer("nhyp_force"); n2,ms2=ctx.timer("outer"); print("fermion force: chain %.2f ms, outer products %.2f ms"%(ms,ms2),flush=True)
sf.release()
# --- HISQ links straight into the operator, Naik multi-shift CG
hq=q.HisqCoefs()
ctx.timers_reset()
t=time.time(); s=q.Staggered(ctx,g,smear=hq); ctx.sync(); dt=time.time()-t
n,ms=ctx.timer("smear"); print("HISQ fat+long links on device: staple kernels %.2f ms, wall %.1f ms (incl. 1 upload)"%(ms,dt*1e3), "links", s.links_info(),flush=True)
ctx.timers_enable(0)
b=q.synthetic_gaussian_vector(lo)
masses=[float(np.sqrt(k+2))*0.05 for k in range(10)]
shifts=[masses[0]]+[4*(m*m-masses[0]**2) for m in masses[1:]]
xs=[np.zeros_like(b) for _ in masses]
def run_multi(n):
    sp=q.SolverParams(r2req=1e-30,maxits=n,verbosity=0)
    t=time.time(); s.solveXX_multi(xs,b,shifts,sp,parEven=True,histcap=8); return time.time()-t, sp.iterations
def run_single(n):
    sp=q.SolverParams(r2req=1e-30,maxits=n,verbosity=0)
    x=np.zeros_like(b); t=time.time(); s.solveXX(x,b,masses[0],sp,parEven=True); return time.time()-t, sp.iterations
run_multi(20); run_single(20)
(t1,i1),(t2,i2)=run_multi(100),run_multi(400)
print("Naik multi-shift CG, 10 shifts: %.1f us/iteration (wall difference of %d and %d iterations)"%(1e6*(t2-t1)/(i2-i1),i2,i1),flush=True)
(t1,i1),(t2,i2)=run_single(100),run_single(400)
print("Naik single-mass CG: %.1f us/iteration"%(1e6*(t2-t1)/(i2-i1)),flush=True)
ctx.timers_enable(1); ctx.timers_reset(); run_single(100)
for name in ("dslash","dslash_bnd","blas","reduce"):
    n,ms=ctx.timer(name); print("   timer",name,n,"%.1f us each"%(1e3*ms/max(n,1)))
