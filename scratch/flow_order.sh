set -e
o=gpurun_out/flow_order.log; : > $o
QEXHIP_FORCE_SYNC=0 python3 scratch/flow_order.py >> $o 2>&1
QEXHIP_FORCE_SYNC=1 python3 scratch/flow_order.py >> $o 2>&1
QEXHIP_FORCE_SYNC=0 python3 scratch/flow_order.py >> $o 2>&1
QEXHIP_FORCE_SYNC=1 python3 scratch/flow_order.py >> $o 2>&1
cat $o
