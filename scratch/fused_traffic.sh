#!/bin/bash
# HBM traffic of the fused sweep on the 48^3 x 12 slab (peer transport, one-rank rehearsal): FETCH_SIZE and WRITE_SIZE in passes of their own
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_fused_$ctr
  QEXHIP_TRANSPORT=peer rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_fused_$ctr -- python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 30 --warmup 5 --repeats 1 --halo --lat 48 48 48 12 --set-option overlap=1 --set-option hop_split=2 > gpurun_out/pmc_fused_$ctr.json 2> gpurun_out/pmc_fused_$ctr.err
done
python3 - <<'P'
import csv, glob, collections
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_fused_%s/**/*counter_collection.csv" % ctr, recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    out[ctr] = {k: (v[0] / v[1], v[1]) for k, v in acc.items()}
for k in sorted(out["FETCH_SIZE"]):
    if "dslash" not in k and "cg_" not in k: continue
    fe, n = out["FETCH_SIZE"][k]; wr = out["WRITE_SIZE"].get(k, (0, 0))[0]
    # corrections of profiles/r06_dslash_traffic.json (1 GiB read / copy calibration kernels of the same round): FETCH x 2.0, WRITE x 1.0; KiB units
    print("%-48s launches %4d  fetch %8.1f MB  write %7.1f MB  total %8.1f MB" % (k[:48], n, fe * 1024 * 2.0 / 1e6, wr * 1024 / 1e6, (fe * 2.0 + wr) * 1024 / 1e6))
P
rm -rf gpurun_out/pmc_fused_FETCH_SIZE gpurun_out/pmc_fused_WRITE_SIZE
