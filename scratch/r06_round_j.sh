#!/bin/bash
# the bench's own kernel timer against rocprofv3's kernel duration, same box, back to back
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/tr_j
python3 bench.py --steps 200 --warmup 20 --no-cpu --no-48x96 --no-extra > gpurun_out/r06_j_plain.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_j -- python3 bench.py --steps 200 --warmup 20 --no-cpu --no-48x96 --no-extra > gpurun_out/r06_j_rocprof.json 2> gpurun_out/r06_j_rocprof.err
python3 bench.py --steps 200 --warmup 20 --no-cpu --no-48x96 --no-extra > gpurun_out/r06_j_plain2.json 2>/dev/null
python3 - <<'P'
import json, glob, csv
for f in ("r06_j_plain", "r06_j_rocprof", "r06_j_plain2"):
    d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
    print(f, "value", d["value"], "ms_per_step", d["ms_per_step"], "roofline avg_us", d["roofline"]["avg_us"], "frac", d["roofline"]["frac"])
st = glob.glob("gpurun_out/tr_j/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(st[0])):
    if "k_dslash" in r["Name"]:
        print("rocprofv3:", r["Name"][:60], "calls", r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3))
P
cp $(find gpurun_out/tr_j -name "*kernel_stats.csv" | head -1) gpurun_out/r06_bench_only_kernel_stats.csv
rm -rf gpurun_out/tr_j
