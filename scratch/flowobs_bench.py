import sys, time; sys.path.insert(0,'.')
import numpy as np, qex_amd as q
lat=[32,32,32,32]; lo=q.Layout(lat)
g=q.synthetic_random_su3(lo, spread=0.3)
ctx=q.Context(lat); q.plaq(ctx,g)
ctx.timers_enable(1)
for loop in (1,3,4,5):
    ctx.timers_reset(); e=q.flowEQ(ctx,loop); n,ms=ctx.timer("flowobs"); print("loop",loop,e,"kernel ms",ms/n,flush=True)
