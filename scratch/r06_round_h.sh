#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; tail -8 gpurun_out/r06_gputests.log
timeout -k 10 500 python bench.py > gpurun_out/r06_bench_32x4.json 2> gpurun_out/r06_bench_32x4.err; tail -c 600 gpurun_out/r06_bench_32x4.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
