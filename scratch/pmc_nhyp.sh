#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit-miss of the staple kernels, generic blocked order vs per-plane order
export TMPDIR=/tmp
OUT=gpurun_out/pmc_nhyp; rm -rf $OUT; mkdir -p $OUT
for o in 0 1; do
  for cnt in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    tag=ord${o}_$(echo $cnt | cut -d' ' -f1)
    QEXHIP_ORD_PLANE=$o timeout -k 5 150 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $OUT/$tag -- python3 profiles/pmc_workload.py nhyp > $OUT/$tag.log 2>&1 || echo "$tag failed"
  done
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
rows = defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_nhyp/*/")):
    tag = os.path.basename(d.rstrip("/")).split("_")[0]
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(s in k for s in ("k_staple_deriv", "k_gen_staple", "k_projUderiv", "k_read16")): continue
            a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(s in k for s in ("k_staple_deriv", "k_gen_staple", "k_projUderiv")): continue
            a = acc[(k, "dur_us")]; a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[1] += 1
    for (k, c), (s, n) in acc.items(): rows[(tag, k)][c] = s / n
for (tag, k), v in sorted(rows.items()):
    fe = v.get("FETCH_SIZE", 0) * 1024 * 2.0 / 1e9; wr = v.get("WRITE_SIZE", 0) * 1024 / 1e9
    print("%s %-40s dur %.1f us  FETCH(x2) %.3f GB  WRITE %.3f GB  L2 hit %.3g miss %.3g  TCP->TCC reads %.3g" % (tag, k[:40], v.get("dur_us", 0), fe, wr, v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0), v.get("TCP_TCC_READ_REQ_sum", 0)))
PY
rm -rf $OUT/*/
