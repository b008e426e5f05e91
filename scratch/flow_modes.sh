#!/bin/bash
# A/B of the RK3-stage kernels: QEXHIP_FORCE_MODE 3 (lane per (site,mu)) vs 4/5 (lane per site, with/without fences)
set -o pipefail
for m in "$@"; do
  echo "== QEXHIP_FORCE_MODE=$m"
  QEXHIP_FORCE_MODE=$m timeout -k 10 300 python3 scratch/flow_bench.py || exit 1
  QEXHIP_FORCE_MODE=$m timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "wflow_golden or gauge_force or full_size_plaq_and_flow" 2>&1 | tail -3 || exit 1
done
