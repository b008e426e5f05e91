"""Timeline of the last CG iterations from a rocprofv3 --kernel-trace CSV: per kernel start / end relative to the k_cg_xpay that opens
the iteration, which queue it ran on, and the gaps.  usage: python scratch/timeline.py <dir with *_kernel_trace.csv> [iterations]"""
import csv, glob, sys
d = sys.argv[1]
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "?")) for r in rows))
idx = [i for i, e in enumerate(ev) if e[2].startswith("k_cg_xpay")]
start = idx[-(nit + 3)]
t0 = ev[start][0]
for s, e, n, q in ev[start:idx[-3] + 1]:
    print("%9.1f %9.1f  %7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
