#!/bin/bash
# one GPU call: the two changed tests, the shared-device ladder on the new build, emulated scaling, the default bench line
mkdir -p gpurun_out
step() { echo "=== $* ==="; "$@"; rc=$?; echo "=== rc $rc ==="; [ $rc -ge 124 ] && { echo "a step had to be killed: stopping"; exit $rc; }; return 0; }
step timeout -k 10 400 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_misc_ops.py -m gpu -q -k "failures_are_errors or placement_follows or overlap_decision" 2>&1 | tail -15
step bash scratch/shared_bisect.sh 2>&1 | tail -12
step bash scratch/emulated_scaling_r06.sh > gpurun_out/r06_emulated_scaling.log 2>&1
tail -40 gpurun_out/r06_emulated_scaling.log
step timeout -k 10 500 python bench.py > gpurun_out/r06_bench_32x4.json 2> gpurun_out/r06_bench_32x4.err
tail -c 1500 gpurun_out/r06_bench_32x4.json
