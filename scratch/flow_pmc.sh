#!/bin/bash
# SQ / TCP / TA counters of the flow kernels (separate --pmc passes), unfused (k_force + k_exp_update) and fused
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/flow_pmc
rm -rf $OUT; mkdir -p $OUT
run() { # name, env..., counters
  local name=$1; shift
  local envs=$1; shift
  env $envs true   # (syntax check of the assignment list)
  for e in $envs; do export $e; done
  timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 profiles/pmc_workload.py flow > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; }
}
for cfg in "ldsch QEXHIP_FORCE_LDS=1 QEXHIP_FLOW_EXP=1" "nolds QEXHIP_FORCE_LDS=0 QEXHIP_FLOW_EXP=1"; do
  set -- $cfg; tag=$1; shift; envs="$*"
  if [ -n "$FLOW_PMC_ONLY" ] && [ "$FLOW_PMC_ONLY" != "$tag" ]; then continue; fi
  run ${tag}_sq "$envs" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM
  run ${tag}_sq2 "$envs" SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVES SQ_IFETCH GRBM_GUI_ACTIVE
  run ${tag}_tcp "$envs" TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
  run ${tag}_tcp2 "$envs" TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
out = "gpurun_out/flow_pmc"
rows = defaultdict(dict)
for d in sorted(glob.glob(out + "/*/")):
    tag = os.path.basename(d.rstrip("/")).split("_")[0]
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not any(s in k for s in ("k_force", "k_exp_update", "k_plaq<", "k_flow_obs")): continue
            a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in acc.items():
        rows[(tag, k)][c] = s / n
cs = sorted({c for v in rows.values() for c in v})
with open(out + "/summary.csv", "w") as fh:
    w = csv.writer(fh); w.writerow(["config", "kernel"] + cs)
    for (tag, k), v in sorted(rows.items()): w.writerow([tag, k] + ["%.4g" % v[c] if c in v else "" for c in cs])
print(open(out + "/summary.csv").read())
PY
