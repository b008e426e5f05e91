for m in 3 4; do for f in 1 0; do QEXHIP_FORCE_MODE=$m QEXHIP_FLOW_FUSED=$f timeout -k 10 200 python3 scratch/flow_sizes.py || exit 1; done; done
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1
