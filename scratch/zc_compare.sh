# peer transport, one rank's slab under emulated transport: chained sweeps + zero-copy receive (defaults) / zero-copy only / neither
run() { env "$1" timeout -k 5 240 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:2}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); m=d.get('multi_gpu',{}); sw=m.get('sweep',{})
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration; overlap', sw.get('overlap'), 'measured', sw.get('measured_us_per_sweep'), 'transport', d.get('transport'), flush=True)" || exit 1; }
for v in "peer_zc=1 sweep_chain=1" "peer_zc=1 sweep_chain=0" "peer_zc=0 sweep_chain=0"; do
  set -- $v
  o="--set-option $1 --set-option $2"
  for lt in 12 24; do
    run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 $lt --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 $o
  done
  run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=1 $o
  run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --set-option overlap=1 $o
  run QEXHIP_TRANSPORT=peer --naik --halo --lat 48 48 48 24 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 $o
  run QEXHIP_TRANSPORT=peer --halo --lat 32 32 32 8 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 $o
done
