#!/bin/bash
# one GPU call: forced-fused bench with 2 / 4 real ranks on the one device, the gather-rows experiment, then the whole GPU suite
mkdir -p gpurun_out
bash scratch/r06_round_c.sh 2>&1 | tee gpurun_out/r06_round_c_forced.log
timeout -k 5 200 python3 scratch/gather_rows.py 2>&1 | tee gpurun_out/r06_gather_rows.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; tail -15 gpurun_out/r06_gputests.log
