#!/bin/bash
# bench.py with 2 and 4 real ranks SHARING the one device, fused sweeps forced (QEXHIP_OVERLAP=1 QEXHIP_HOP_SPLIT=2; left to itself
# set_links finds that overlapping does not pay between processes on one chip: profiles/r06_bench_ranks_one_device.log)
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4 QEXHIP_PEER_TIMEOUT=20
mkdir -p gpurun_out
for n in 2 4; do
  for hs in 2; do
    t0=$(date +%s)
    QEXHIP_OVERLAP=1 QEXHIP_HOP_SPLIT=$hs timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29560 + n)) bench.py --gpus $n --steps 20 --warmup 5 > gpurun_out/r06_bench_${n}ranks_one_device_hs$hs.json 2> gpurun_out/r06_bench_${n}ranks_hs$hs.err
    rc=$?
    echo "n=$n hop_split=$hs rc=$rc $(( $(date +%s) - t0 )) s"
    python3 - <<P
import json
try:
    d=json.loads([l for l in open('gpurun_out/r06_bench_${n}ranks_one_device_hs$hs.json') if l.startswith('{')][-1]); x=d.get('cg_48x48x48x96',{})
    print(' 32^4:', d.get('error'), d.get('transport'), d['shard_check']['ok'], d['ms_per_step'], d['multi_gpu']['sweep'].get('form'), d['multi_gpu']['sweep'].get('tuned_us_per_sweep'), '| 48^3x96:', x.get('error'), x.get('shard_check',{}).get('ok'), x.get('ms_per_step'), x.get('multi_gpu',{}).get('sweep',{}).get('form'), x.get('multi_gpu',{}).get('sweep',{}).get('tuned_us_per_sweep'), '| naik:', json.dumps(d.get('naik_multishift_48x48x48x96'))[:200])
except Exception as e:
    print(' no line:', e)
P
    [ $rc -ge 124 ] && exit $rc
  done
done
