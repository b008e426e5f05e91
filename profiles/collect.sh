#!/bin/bash
# One script regenerates every rocprofv3 summary the bench line and DESIGN.md quote (run on the GPU box):
#     bash profiles/collect.sh TAG            e.g. TAG = r03
# writes gpurun_out/prof_TAG/ (raw, scratch) and the summaries gpurun_out/prof_TAG/TAG_*.csv|json, which are then
# copied into profiles/ and committed.  Counters are collected in their own passes (--pmc with --kernel-trace only).
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
# 1. per-kernel time of the default bench command (the 32^4 workload alone: no CPU leg, no 48^3x96 leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench_32x4 -- python3 bench.py --steps 200 --warmup 20 --no-cpu --no-48x96 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# 2. per-kernel time of the profiling workload (flow, nHYP chain, Naik multi-shift)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_workload -- python3 profiles/pmc_workload.py > $OUT/workload.log 2>&1
# 3. PMC passes, one counter group each
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 profiles/pmc_workload.py >> $OUT/workload.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 profiles/pmc_workload.py >> $OUT/workload.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -- python3 profiles/pmc_workload.py flow nhyp >> $OUT/workload.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 profiles/pmc_workload.py flow nhyp >> $OUT/workload.log 2>&1
python3 profiles/summarize.py $OUT $TAG
ls -la $OUT/*.csv $OUT/*.json
# raw rocprofv3 output is scratch: keep it only while it is small (gpurun merges at most 64 MiB back)
if [ $(du -sm $OUT | cut -f1) -gt 40 ]; then rm -rf $OUT/trace_* $OUT/pmc_*; fi
