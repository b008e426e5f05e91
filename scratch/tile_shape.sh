#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 scratch/tile_shape.py time > gpurun_out/r05_tile_shape_time.log 2>&1 || { tail -5 gpurun_out/r05_tile_shape_time.log; exit 1; }
cat gpurun_out/r05_tile_shape_time.log
for m in a b; do
  for pmc in "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    rm -rf gpurun_out/ts_$m_$tag
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d gpurun_out/ts_${m}_$tag -- python3 scratch/tile_shape.py $m > /dev/null 2>&1
    python3 - "$m" "$tag" <<'PY'
import csv, glob, sys
m, tag = sys.argv[1], sys.argv[2]
f = glob.glob("gpurun_out/ts_%s_%s/**/*counter_collection.csv" % (m, tag), recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_gather" in r["Kernel_Name"]]
agg = {}
for r in rows:
    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("shape %s  %s: mean per launch %.4g over %d launches" % ({"a": "rows", "b": "bricks"}[m], k, sum(v) / len(v), len(v)))
PY
    rm -rf gpurun_out/ts_${m}_$tag
  done
done
