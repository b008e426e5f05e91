#!/bin/bash
# A/B of the streaming (non-temporal) accumulator / output accesses, same box, interleaved
cd $GRAFT_REPO_ROOT
for B in 0 1 0 1; do
  echo "== FORCE_NT $B"; QEXHIP_FORCE_NT=$B timeout -k 5 120 python3 scratch/flow_bench.py 2>&1 | grep -E "staple|4 flow" || exit 1
done
for B in 0 1 0 1; do
  echo "== PROJ_NT $B"; QEXHIP_PROJ_NT=$B timeout -k 5 120 python3 scratch/nhyp_force_bench.py 2>&1 | grep -E "gforce" || exit 1
done
for B in 0 1 0 1; do
  echo "== STAPLE_NT $B"; QEXHIP_STAPLE_NT=$B timeout -k 5 120 python3 scratch/nhyp_force_bench.py 2>&1 | grep -E "prepare wall" || exit 1
done
