# clover flow observables: the tile-per-workgroup kernel against the path walker (QEXHIP_OBS_CLOVER=0), same field
import sys, os, subprocess, json
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import numpy as np, qex_amd as q
    lat = [32, 32, 32, 32]
    g = q.RngField(lat, q.RngMilc6, 987654321).warm(0.5)
    ctx = q.Context(lat); q.plaq(ctx, g)
    ctx.timers_enable(1)
    e = q.flowEQ(ctx, 1)
    ctx.timers_reset()
    for i in range(5): e = q.flowEQ(ctx, 1)
    n, ms = ctx.timer("flowobs")
    print(json.dumps({"E": [float(v) for v in e], "kernel_us": 1e3 * ms / n}))
else:
    out = {}
    for v in ("0", "1", "0", "1"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, QEXHIP_OBS_CLOVER=v), capture_output=True, text=True, timeout=200)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line: print(r.stdout, r.stderr); sys.exit(1)
        d = json.loads(line[-1]); out[v] = d
        print("clover", v, d, flush=True)
    a, b = out["0"]["E"], out["1"]["E"]
    print("rel diff", [abs(x - y) / max(abs(x), 1e-300) for x, y in zip(a, b)])
