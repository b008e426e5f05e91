#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 scratch/flow_ring_ab.py > gpurun_out/ring_ab.log 2>&1; rc=$?; echo "rc=$rc"; tail -12 gpurun_out/ring_ab.log
[ $rc -eq 0 ] || exit 1
QEXHIP_FLOW_STAGE_DBG=1 timeout -k 10 200 python3 scratch/flow_ring_ab.py > gpurun_out/ring_ab_dbg1.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/ring_ab_dbg1.log | grep "ring=1"
QEXHIP_FLOW_STAGE_DBG=2 timeout -k 10 200 python3 scratch/flow_ring_ab.py > gpurun_out/ring_ab_dbg2.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/ring_ab_dbg2.log | grep "ring=1"
QEXHIP_FLOW_STAGE_RS=0 timeout -k 10 200 python3 scratch/flow_ring_ab.py > gpurun_out/ring_ab_rs0.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/ring_ab_rs0.log | grep "ring=1"
