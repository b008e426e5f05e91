#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_nhyp; mkdir -p gpurun_out/prof_nhyp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nhyp/t -- python3 profiles/pmc_workload.py nhyp > gpurun_out/prof_nhyp/log 2>&1
f=$(find gpurun_out/prof_nhyp/t -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/prof_nhyp/kernel_stats.csv; rm -rf gpurun_out/prof_nhyp/t
head -12 gpurun_out/prof_nhyp/kernel_stats.csv | cut -c1-200
