#!/bin/bash
# Round 6, VERDICT item 1(a): bisect the fused sweep's shared-device failure by slab size and by operator (8 links: 2 F/256 boundary
# workgroups per sweep; Naik: 6 F/256).  Every case: two fresh processes on device 0 under their own timeout; a failed case does not
# stop the ladder (a wait that runs out is an error return, not a hang), a case that has to be KILLED does.
# Output: gpurun_out/shared_bisect.log (one BISECT line per rank and case).
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4 QEXHIP_PEER_TIMEOUT=${QEXHIP_PEER_TIMEOUT:-12} MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
LOG=gpurun_out/shared_bisect.log
: > $LOG
port=29600
run_case() {   # lat4 [--naik]
  port=$((port + 1))
  echo "=== case $* ===" | tee -a $LOG
  for r in 0 1; do
    RANK=$r WORLD_SIZE=2 LOCAL_RANK=$r MASTER_PORT=$port timeout -k 5 150 python3 tests/shared_device_worker.py "$@" >> $LOG.rank$r 2>&1 &
    pids[$r]=$!
  done
  rc=0
  for r in 0 1; do wait ${pids[$r]}; c=$?; [ $c -gt $rc ] && rc=$c; done
  for r in 0 1; do grep -h "^BISECT\|Error\|error" $LOG.rank$r | tail -4 >> $LOG; : > $LOG.rank$r; done
  echo "=== rc $rc ===" | tee -a $LOG
  [ $rc -ge 124 ] && { echo "case had to be killed: stopping the ladder" | tee -a $LOG; exit 1; }
  return 0
}
# 8 links: boundary workgroups 2 F/256 -- 128 ... 432 ... 1024
run_case 32 32 32 32
run_case 48 48 48 24
run_case 48 48 48 96
run_case 64 64 64 16
# Naik: 6 F/256 -- 384 (32^3), 750 (40^3), 864 (48 48 32), 1296 (48^3)
run_case 32 32 32 32 --naik
run_case 40 40 40 16 --naik
run_case 48 48 32 16 --naik
run_case 48 48 48 24 --naik
echo "ladder done" | tee -a $LOG
