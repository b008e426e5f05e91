set -e
o=gpurun_out/flow_order.log; : > $o
for m in 3 5 6 3 5 6; do
QEXHIP_FORCE_MODE=$m python3 scratch/flow_order.py >> $o 2>&1
done
cat $o
