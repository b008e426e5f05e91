# one-rank rehearsal: us/iteration of the sharded CG for slab sizes x overlap modes (the split's cost against the exposed exchange)
run() { timeout -k 5 200 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration', flush=True)" || exit 1; }
for lt in 16 8 4; do for ov in 0 1; do run --halo --lat 32 32 32 $lt --set-option overlap=$ov; done; done
for lt in 48 24 12; do for ov in 0 1; do run --halo --lat 48 48 48 $lt --set-option overlap=$ov; done; done
run --lat 32 32 32 32
run --lat 48 48 48 96
