#!/bin/bash
# QEXHIP_ORD_* sweep of the flow stage; default first and last (box drift)
run() { QEXHIP_ORD_RZ=$1 QEXHIP_ORD_Y=$2 QEXHIP_ORD_Z=$3 QEXHIP_ORD_T=$4 timeout -k 5 120 python3 scratch/order_sweep.py 2>&1 | grep order; }
run 1 8 4 4
run 2 8 4 4
run 2 4 4 8
run 2 8 2 8
run 2 4 8 4
run 2 8 8 2
run 4 8 4 4
run 4 4 4 8
run 4 8 2 8
run 1 4 8 4
run 1 16 4 2
run 1 16 2 4
run 1 8 4 4
