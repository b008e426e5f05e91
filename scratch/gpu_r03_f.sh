#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_smear.py tests/test_golden_hmc.py tests/test_spv_hmc.py -m gpu -q -x > gpurun_out/tests_r03f.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/tests_r03f.log
for qd in 1 0 1 0; do
QEXHIP_SDERIV_QUAD=$qd timeout -k 10 200 python3 scratch/nhyp_force_bench.py > gpurun_out/nhyp_bench_q$qd.log 2>&1; echo "quad=$qd rc=$?"; grep "gforce" gpurun_out/nhyp_bench_q$qd.log | tail -1
done
