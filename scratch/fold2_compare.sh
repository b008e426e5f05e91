# the CG's two rank sums inside the consuming kernels (option peer_fold = 2) against one-workgroup launches (0), alternating, timers off
run() { env "$1" "$2" timeout -k 5 240 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:3}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration', flush=True)" || exit 1; }
for rep in 1 2; do
  for f in 2 0; do
    run QEXHIP_TRANSPORT=peer QEX_BENCH_TIMERS=0 --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option peer_fold=$f
    run QEXHIP_TRANSPORT=peer QEX_BENCH_TIMERS=0 --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=1 --set-option peer_fold=$f
    run QEXHIP_TRANSPORT=peer QEX_BENCH_TIMERS=0 --halo --lat 32 32 32 4 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option peer_fold=$f
  done
done
