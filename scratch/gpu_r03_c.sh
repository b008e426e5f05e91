#!/bin/bash
for d in 1 3 4; do
QEXHIP_FLOW_STAGE_DBG=$d timeout -k 10 200 python3 scratch/flow_ring_ab.py > gpurun_out/ring_ab_dbg$d.log 2>&1; echo "dbg $d rc=$?"; tail -6 gpurun_out/ring_ab_dbg$d.log | grep "ring=1"
done
