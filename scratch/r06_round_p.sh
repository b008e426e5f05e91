#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_ranks.py tests/test_gpu_misc_ops.py -m gpu -x -q 2>&1 | tail -5
run() { env "$1" timeout -k 5 240 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:2}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); m=d.get('multi_gpu',{})
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration', {k: m.get(k) for k in ('interior_us','boundary_us','exchange_us','allreduce_us')}, flush=True)" || { rc=$?; [ $rc -ge 124 ] && exit $rc; }; }
run QEXHIP_TRANSPORT=rccl --lat 48 48 48 96
for rep in 1 2; do
run QEXHIP_TRANSPORT=rccl --halo --lat 48 48 48 12 --emulate-transport 3 15 --set-option emu_link_gbs=45 --set-option overlap=-2
run QEXHIP_TRANSPORT=mbox --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
run QEXHIP_TRANSPORT=mbox --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=-2
run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=0
run QEXHIP_TRANSPORT=mbox --halo --lat 48 48 48 24 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
done
run QEXHIP_TRANSPORT=mbox --naik --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
