# nHYP smear + force chain on resident fields (what bench.py's resident_md leg times), for A/B runs by environment
import sys
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
lat = [32, 32, 32, 32]
g0 = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
hc = q.HypCoefs(0.4, 0.5, 0.5)
md = q.ResidentMD(ctx)
md.begin(g0, None)
sr = hc.smearGetForce(ctx, None); sr.gforce(None, plaq=1.0)
res = []
for rnd in range(3):
    ctx.timers_enable(1); ctx.timers_reset()
    for _ in range(4):
        sr = hc.smearGetForce(ctx, None); sr.gforce(None, plaq=1.0)
    ctx.sync()
    n, ms = ctx.timer("nhyp_force"); ns, mss = ctx.timer("smear")
    res.append("%.3f/%.3f" % (ms / n, mss / 4))
f = np.zeros_like(g0)
md.kick(md.NHYP, 1.0); md.end(None, f)
print("chain/smear ms:", " ".join(res), "checksum %.12e" % float((f * f).sum()), flush=True)
