"""Round 6: where the smeared gauge-force chain of a t-sharded slab waits under emulated transport.
  run:     rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scratch/gforce_timeline.py run
  digest:  python3 scratch/gforce_timeline.py digest DIR      (the last gforce call: kernels grouped, per-queue busy time, idle gaps of the compute queue)"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np, qex_amd as q
    lat = [48, 48, 48, 12]
    g = q.RngField(lat, q.RngMilc6, 987654321).warm(0.5)
    ctx = q.Context(lat)
    ctx.comm_init(q.Context.unique_id(), 1, 0)
    ctx.force_halo(True)
    ctx.set_option("multi_reduce", 1)
    if os.environ.get("QEX_EMU", "1") != "0":
        ctx.set_option("emu_exchange_us", 3); ctx.set_option("emu_link_gbs", 45); ctx.set_option("emu_allreduce_us", 3)
    fl, f = np.zeros_like(g), np.zeros_like(g)
    import time
    sf = q.HypCoefs(0.4, 0.5, 0.5).smearGetForce(ctx, g, fl)
    for _ in range(3):
        t = time.perf_counter()
        sf = q.HypCoefs(0.4, 0.5, 0.5).smearGetForce(ctx, g, fl)
        ctx.sync()
        print("smear wall %.2f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    for _ in range(4):
        t = time.perf_counter()
        sf.gforce(f, plaq=1.0)
        ctx.sync()
        print("gforce wall %.2f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    print("transport", ctx.comm_transport()[0])
else:
    d = sys.argv[2]
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r.get("Queue_Id", "?")) for r in rows))
    # the last call: everything behind the second-to-last k_sm_from_tiles (the download conversion that ends a gforce call)
    ends = [i for i, e in enumerate(ev) if e[2].startswith("k_sm_from_tiles")]
    cut = ends[-2] + 1 if len(ends) >= 2 else 0
    ev = ev[cut:ends[-1] + 1]
    t0 = ev[0][0]
    print("last call: %d kernels, %.2f ms from first start to last end" % (len(ev), (max(e[1] for e in ev) - t0) / 1e6))
    busy = {}
    for s, e, n, qd in ev:
        busy[qd] = busy.get(qd, 0) + (e - s)
    print("busy per queue (ms):", {k: round(v / 1e6, 2) for k, v in busy.items()})
    # grouped listing
    grp = []
    for s, e, n, qd in ev:
        if grp and grp[-1][2] == n and grp[-1][3] == qd and s - grp[-1][1] < 200_000:
            grp[-1][1] = e; grp[-1][4] += 1; grp[-1][5] += e - s
        else:
            grp.append([s, e, n, qd, 1, e - s])
    prev_end = {}
    for s, e, n, qd, k, b in grp:
        gap = (s - prev_end[qd]) / 1e6 if qd in prev_end else 0.0
        prev_end[qd] = e
        if gap > 0.3:
            print("%31s q%-3s idle %.2f ms" % ("", qd, gap))
        print("%9.2f %9.2f ms  q%-3s %3d x %-44s busy %.2f ms" % ((s - t0) / 1e6, (e - t0) / 1e6, qd, k, n, b / 1e6))
