#!/bin/bash
# first runs of the peer transport: 2 ranks sharing device 0, then 4
mkdir -p gpurun_out
out=gpurun_out/peer_first.log
: > $out
export QEXHIP_PEER_TIMEOUT=20 OMP_NUM_THREADS=4 HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  echo "=== $*" >> $out
  timeout -k 10 400 "$@" >> $out 2>&1
  rc=$?
  echo "rc=$rc" >> $out
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "killed: stopping" >> $out; tail -50 $out; exit 1; fi
}
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tests/two_rank_worker.py 8 8 8 8 --overlap 1 --share-device
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 tests/two_rank_worker.py 16 16 16 32 --overlap -1 --share-device
run python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 tests/two_rank_worker.py 8 8 8 16 --overlap 1 --share-device
tail -c 6000 $out
