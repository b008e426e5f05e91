run() { env $1 timeout -k 5 240 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:2}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); m=d.get('multi_gpu',{}); sw=m.get('sweep',{})
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration; overlap', sw.get('overlap'), 'measured', sw.get('measured_us_per_sweep'), {k: m.get(k) for k in ('interior_us','boundary_us','exchange_us','allreduce_us')}, flush=True)" || exit 1; }
for k in 0 8 16 32; do
  run "QEXHIP_TRANSPORT=peer QEXHIP_COMM_CUS=$k" --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1
done
run "QEXHIP_TRANSPORT=peer QEXHIP_COMM_CUS=8" --lat 48 48 48 96
run "QEXHIP_TRANSPORT=peer QEXHIP_COMM_CUS=0" --lat 48 48 48 96
