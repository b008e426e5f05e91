#!/bin/bash
# Reproducer for the one open failure of round 5: two ranks that SHARE a GPU, fused sweeps forced (QEXHIP_HOP_SPLIT=2), the 48^3 x 96 leg
# of bench.py: a boundary workgroup's bounded wait for the other process's push expires (code 0x510).  Green: the same with
# QEXHIP_HOP_SPLIT=0 (the default for ranks sharing a GPU), one rank at that size, 2 / 4 ranks up to 16^3 x 32.
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4 QEXHIP_PEER_TIMEOUT=${QEXHIP_PEER_TIMEOUT:-10} QEXHIP_HOP_SPLIT=${QEXHIP_HOP_SPLIT:-2}
timeout -k 5 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu 2> gpurun_out/shared_fused.err | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); x=d.get('cg_48x48x48x96',{})
print('32^4:', d.get('error'), d['shard_check']['ok'], '| 48^3x96:', x.get('error'), x.get('shard_check',{}).get('ok'), x.get('ms_per_step'))"
