# sample rocm-smi (power, sclk, mclk) while a kernel mix runs in a child process: what clock does the chip hold under each mix?
import subprocess, sys, time, json, os
mix = sys.argv[1]
child = r'''
import sys, time
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
from qex_amd._lib import check
L = q.lib()
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
ctx = q.Context(lat)
mix = "%s"
t_end = time.time() + 9.0
if mix == "flow":
    q.gaugeSet(ctx, g)
    while time.time() < t_end:
        q.gaugeSet(ctx, g) if False else None
        check(L.qexhip_wflow(ctx._h, 40, 0.0001)); ctx.sync()
elif mix == "cg":
    s = q.Staggered(ctx, g)
    b = ctx.field_new(np.random.default_rng(5).standard_normal((ctx.vol, 3, 2))); x = ctx.field_new(None)
    while time.time() < t_end:
        ctx.field_zero(x); ctx.dev_solve_xx(x, b, 0.01, 0.0, 400); ctx.sync()
elif mix == "chain":
    hc = q.HypCoefs(0.4, 0.5, 0.5); md = q.ResidentMD(ctx); md.begin(g, None)
    while time.time() < t_end:
        for _ in range(5):
            sr = hc.smearGetForce(ctx, None); sr.gforce(None, plaq=1.0)
        ctx.sync()
print("child done", flush=True)
''' % mix
p = subprocess.Popen([sys.executable, "-c", child])
time.sleep(5.0)      # import + setup
for i in range(8):
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        for card, v in d.items():
            keys = {k: v[k] for k in v if any(t in k.lower() for t in ("power", "sclk", "mclk", "fclk"))}
            print(mix, "sample", i, card, keys, flush=True)
    except Exception as e:
        print("rocm-smi failed:", repr(e)[:200], flush=True)
        break
    time.sleep(0.5)
p.wait()
