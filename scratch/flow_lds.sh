for cfg in "QEXHIP_FORCE_LDS=0 QEXHIP_FLOW_EXP=1" "QEXHIP_FORCE_LDS=1 QEXHIP_FLOW_EXP=1" "QEXHIP_FORCE_LDS=1 QEXHIP_FLOW_EXP=0" "QEXHIP_FORCE_LDS=0 QEXHIP_FLOW_EXP=0"; do
  echo "== $cfg"; env $cfg timeout -k 10 300 python3 scratch/flow_bench.py 2>&1 | grep -E "staple|unitarity"
done
QEXHIP_FORCE_LDS=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_flow_obs.py tests/test_gauge_actions.py tests/test_golden_hmc.py -q -m gpu 2>&1 | tail -3
