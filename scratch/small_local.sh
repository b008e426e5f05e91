#!/bin/bash
# one-rank rehearsal of the per-rank problem of 32^4 on 8/4/2 ranks: wall per iteration against the sum of kernel durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in 4 8 16; do
  O=$R/gpurun_out/small_local/t$T
  rm -rf $O; mkdir -p $O
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --lat 32 32 32 $T --halo --no-cpu --no-extra --no-48x96 --steps 400 --warmup 40 > $O/bench.log 2>$O/bench.err || exit 1
  python3 - $O <<'PY'
import sys,glob,csv,json
o=sys.argv[1]
d=json.loads([l for l in open(o+"/bench.log") if l.startswith("{")][-1])
print("lat",d["config"].get("lattice"),"ms/it",d["ms_per_step"],"it/s",d["cg_iters_per_s"])
f=glob.glob(o+"/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]: print("  %-60s calls %6s avg %9.1f us"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
