# one-rank rehearsal of every rank's slab with EMULATED transport: a wait of the transfer time of a face (bytes / 45 GB/s per xGMI
# direction) in front of every exchange and of 15 us in front of every all-reduce; the overlap decision is measured under
# that latency (option overlap = -2), as it would be on a real communicator
run() { timeout -k 5 240 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); m=d.get('multi_gpu',{}); sw=m.get('sweep',{})
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration; overlap', sw.get('overlap'), 'measured', sw.get('measured_us_per_sweep'), flush=True)" || exit 1; }
run --lat 48 48 48 96
for lt in 48 24 12; do run --halo --lat 48 48 48 $lt --emulate-transport 59 15 --set-option overlap=-2; done
for lt in 48 24 12; do run --halo --lat 48 48 48 $lt --emulate-transport 118 30 --set-option overlap=-2; done
run --lat 32 32 32 32
for lt in 16 8 4; do run --halo --lat 32 32 32 $lt --emulate-transport 18 15 --set-option overlap=-2; done
