#!/bin/bash
mkdir -p gpurun_out
step() { echo "=== $* ==="; "$@"; rc=$?; echo "=== rc $rc ==="; [ $rc -ge 124 ] && { echo "a step had to be killed: stopping"; exit $rc; }; return 0; }
step bash scratch/fused_ab2.sh 2>&1 | tee gpurun_out/r06_fused_ab3.log
for tr in mbox peer; do
  QEXHIP_TRANSPORT=$tr QEX_EMU=3,45,3 step timeout -k 5 200 python3 scratch/batch_halo_bench.py 48x48x48x12 1 1 2>&1 | tee -a gpurun_out/r06_batch_halo.log
done
step timeout -k 5 200 python3 scratch/batch_halo_bench.py 48x48x48x12 0 2>&1 | tee -a gpurun_out/r06_batch_halo.log
for tr in mbox peer; do
  QEXHIP_TRANSPORT=$tr step timeout -k 5 400 python3 scratch/config4_emulated.py 2>&1 | tee -a gpurun_out/r06_config4_emulated.log
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step timeout -k 10 900 bash profiles/collect.sh r06 > gpurun_out/r06_collect.log 2>&1
tail -5 gpurun_out/r06_collect.log
