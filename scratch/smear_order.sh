set -e
o=gpurun_out/smear_order.log; : > $o
for cfg in "8 2 4" "8 4 4" "8 8 4" "32 1 4" "4 2 4" "8 2 2" "16 2 4" "8 1 4" "4 4 4" "8 2 1" "32 32 1"; do
  set -- $cfg
  echo "== ORD $cfg" >> $o
  QEXHIP_ORD_Y=$1 QEXHIP_ORD_Z=$2 QEXHIP_ORD_T=$3 python3 scratch/nhyp_force_bench.py 2>&1 | grep "prepare wall\|gforce" | tail -2 >> $o
  QEXHIP_ORD_Y=$1 QEXHIP_ORD_Z=$2 QEXHIP_ORD_T=$3 python3 scratch/hisq_force_bench.py 2>&1 | tail -1 >> $o
  QEXHIP_ORD_Y=$1 QEXHIP_ORD_Z=$2 QEXHIP_ORD_T=$3 python3 scratch/flow_order.py 2>&1 | tail -1 >> $o
done
cat $o
