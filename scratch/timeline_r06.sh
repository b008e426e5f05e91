#!/bin/bash
# kernel timeline of the emulated 48^3 x 12 slab iteration: RCCL faces + mailbox sums (`mbox`, split by sites) and the peer transport (fused)
export TMPDIR=/tmp
for tr in mbox; do
  rm -rf gpurun_out/tl_$tr
  QEXHIP_TRANSPORT=$tr rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tr -- python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 40 --warmup 10 --repeats 1 --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 > gpurun_out/tl_$tr.json 2> gpurun_out/tl_$tr.err
  python3 scratch/timeline.py gpurun_out/tl_$tr 2 > gpurun_out/r06_timeline_$tr.txt 2>&1
  rm -rf gpurun_out/tl_$tr
done
cat gpurun_out/r06_timeline_mbox.txt
