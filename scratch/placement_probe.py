# is the 106 vs 112 us Dslash a property of the box or of where the buffers landed?  Fresh contexts in ONE process, with
# dummy allocations of varying size in between (torch caching allocator is not involved: the library uses hipMalloc).
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import qex_amd as q
lat = [32, 32, 32, 32]
g = q.RngField(lat, q.RngMilc6, 987654321).random()
hip = C.CDLL("libamdhip64.so")
def dummy(nbytes):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    return p
keep = []
b = np.random.default_rng(5).standard_normal((np.prod(lat), 3, 2))
for trial, pad in enumerate([0, 0, 1 << 20, 3 << 20, 64 << 20, 65 << 20, 257 << 20, 0]):
    if pad: keep.append(dummy(pad))
    ctx = q.Context(lat)
    s = q.Staggered(ctx, g)
    bid = ctx.field_new(b); xid = ctx.field_new(None)
    ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 50); ctx.sync()
    t0 = time.perf_counter(); ctx.dev_solve_xx(xid, bid, 0.1, 0.0, 400); ctx.sync(); dt = time.perf_counter() - t0
    print("trial %d (dummy allocation of %d MiB before): %.2f us/iteration" % (trial, pad >> 20, 1e6 * dt / 400), flush=True)
    ctx.close()
