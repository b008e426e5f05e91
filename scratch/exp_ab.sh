#!/bin/bash
# A/B of the flow stage: previous build (scratch/ab/libqexhip_prev.so) against the tree's, alternating processes
for i in 1 2 3; do
  echo "prev:"; QEXHIP_LIB=$PWD/scratch/ab/libqexhip_prev.so timeout -k 5 120 python3 scratch/order_sweep.py 2>&1 | grep order
  echo "tree:"; timeout -k 5 120 python3 scratch/order_sweep.py 2>&1 | grep order
done
