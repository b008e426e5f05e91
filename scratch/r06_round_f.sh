#!/bin/bash
mkdir -p gpurun_out
step() { echo "=== $* ==="; "$@"; rc=$?; echo "=== rc $rc ==="; [ $rc -ge 124 ] && { echo "a step had to be killed: stopping"; exit $rc; }; return 0; }
step timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -15
step timeout -k 10 600 python -m pytest tests/test_gpu_two_ranks.py -m gpu -x -q -k "sharing_one_device_against" 2>&1 | tail -8
QEXHIP_TRANSPORT=peer QEX_EMU=3,45,3 step timeout -k 5 200 python3 scratch/batch_halo_bench.py 48x48x48x12 1 1 2>&1 | tee -a gpurun_out/r06_batch_halo_fused.log
QEXHIP_TRANSPORT=peer QEXHIP_HOP_SPLIT=0 QEX_EMU=3,45,3 step timeout -k 5 200 python3 scratch/batch_halo_bench.py 48x48x48x12 1 1 2>&1 | tee -a gpurun_out/r06_batch_halo_fused.log
