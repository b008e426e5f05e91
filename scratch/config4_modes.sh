#!/bin/bash
# configs[4]'s gauge-sector pieces on the 48^3 x 12 slab: every mode in a process of its own (see scratch/config4_emulated.py)
for tr in mbox peer; do
  for mode in periodic halo halo+emu; do
    QEXHIP_TRANSPORT=$tr timeout -k 5 300 python3 scratch/config4_emulated.py 48x48x48x12 $mode 2>&1 | grep "transport"
  done
done
