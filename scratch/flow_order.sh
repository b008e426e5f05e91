set -e
o=gpurun_out/flow_order.log; : > $o
for pm in 0 1 2; do
QEXHIP_PLAQ_MODE=$pm QEXHIP_FORCE_MODE=3 QEXHIP_ORD_Y=8 QEXHIP_ORD_Z=2 QEXHIP_ORD_T=4 python3 scratch/flow_order.py >> $o 2>&1
QEXHIP_PLAQ_MODE=$pm QEXHIP_FORCE_MODE=3 QEXHIP_ORD_Y=8 QEXHIP_ORD_Z=4 QEXHIP_ORD_T=4 python3 scratch/flow_order.py >> $o 2>&1
QEXHIP_PLAQ_MODE=$pm QEXHIP_FORCE_MODE=3 QEXHIP_ORD_Y=4 QEXHIP_ORD_Z=2 QEXHIP_ORD_T=2 python3 scratch/flow_order.py >> $o 2>&1
QEXHIP_PLAQ_MODE=$pm QEXHIP_FORCE_MODE=3 QEXHIP_ORD_Y=32 QEXHIP_ORD_Z=1 QEXHIP_ORD_T=1 python3 scratch/flow_order.py >> $o 2>&1
done
cat $o
