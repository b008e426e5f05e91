#!/bin/bash
for m in "-resident" "-device"; do
timeout -k 10 300 python3 examples/staghmc_sh.py -run 0 -trajs 1 -lat 32 32 32 32 -time $m > gpurun_out/traj32_$m.log 2>&1; echo "$m rc=$?"; grep -E "TIME|Begin|End|ACCEPT|REJECT|MEASpbp" gpurun_out/traj32_$m.log | cut -c1-400
done
timeout -k 10 100 python3 examples/staghmc_sh.py -run 0 -trajs 1 -device > gpurun_out/traj8_device.log 2>&1; echo rc=$?; grep -E "Begin|End|MEASp" gpurun_out/traj8_device.log | cut -c1-200
