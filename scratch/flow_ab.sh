#!/bin/bash
# usage: flow_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   one flow_bench run per environment string + the flow parity tests
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python3 scratch/flow_bench.py 2>&1 | grep -E "staple|expupdate|unitarity" || exit 1
  env $cfg timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "wflow_golden or gauge_force or full_size_plaq_and_flow" 2>&1 | tail -1 || exit 1
done
