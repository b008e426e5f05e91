#!/bin/bash
# counters of the loader/consumer flow stage (k_flow_stage) against k_force_lds: L2 hit rate, L2->L1 requests, HBM bytes, SQ waits
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/ring_pmc
rm -rf $OUT; mkdir -p $OUT
run() { # name, envs, counters...
  local name=$1; shift
  local envs=$1; shift
  ( for e in $envs; do export $e; done
    timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 profiles/pmc_workload.py flow > $OUT/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $OUT/$name.log; }
}
for cfg in "ring QEXHIP_FLOW_RING=1" "ringdbg1 QEXHIP_FLOW_RING=1 QEXHIP_FLOW_STAGE_DBG=1" "ringdma QEXHIP_FLOW_RING=1 QEXHIP_FLOW_STAGE_RS=0" "lds QEXHIP_FLOW_RING=0"; do
  set -- $cfg; tag=$1; shift; envs="$*"
  run ${tag}_l2 "$envs" TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
  run ${tag}_fetch "$envs" FETCH_SIZE
  run ${tag}_write "$envs" WRITE_SIZE
  run ${tag}_sq "$envs" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
  run ${tag}_tcp "$envs" TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
out = "gpurun_out/ring_pmc"
rows = defaultdict(dict)
for d in sorted(glob.glob(out + "/*/")):
    tag = os.path.basename(d.rstrip("/")).split("_")[0]
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(s in k for s in ("k_force_lds", "k_flow_stage")): continue
            a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(s in k for s in ("k_force_lds", "k_flow_stage")): continue
            a = acc[(k, "dur_us")]; a[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3; a[1] += 1
    for (k, c), (s, n) in acc.items():
        rows[(tag, k)][c] = s / n
cs = sorted({c for v in rows.values() for c in v})
with open(out + "/summary.csv", "w") as fh:
    w = csv.writer(fh); w.writerow(["config", "kernel"] + cs)
    for (tag, k), v in sorted(rows.items()): w.writerow([tag, k] + ["%.5g" % v[c] if c in v else "" for c in cs])
print(open(out + "/summary.csv").read())
PY
rm -rf $OUT/*/  # raw output is scratch
