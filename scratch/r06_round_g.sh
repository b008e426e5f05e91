#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for emu in 1 0; do
  rm -rf gpurun_out/tl_gf
  QEX_EMU=$emu QEXHIP_TRANSPORT=mbox timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_gf -- python3 scratch/gforce_timeline.py run > gpurun_out/tl_gf.log 2>&1
  rc=$?; [ $rc -ge 124 ] && exit $rc
  python3 scratch/gforce_timeline.py digest gpurun_out/tl_gf > gpurun_out/r06_gforce_timeline_emu$emu.txt 2>&1
  rm -rf gpurun_out/tl_gf
done
cat gpurun_out/r06_gforce_timeline_emu1.txt
