#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/peer_dbg.log
: > $out
export QEXHIP_PEER_TIMEOUT=20 OMP_NUM_THREADS=4 HSA_ENABLE_IPC_MODE_LEGACY=0 QEX_WORKER_VERBOSE=1
port=29800
for ov in -1 -1 0 1 -1; do
  port=$((port+1))
  echo "=== overlap $ov" >> $out
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port tests/two_rank_worker.py 8 8 8 8 --overlap $ov --share-device --skip-gauge 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|^  File\|^    \|elastic\|^=====\|^-----" | cut -c1-700 >> $out
  echo "rc=${PIPESTATUS[0]}" >> $out
done
cat $out
