# one rank's 48^3 x 12 share of 48^3 x 96 at N = 8, rehearsed on one GPU: where the iteration's time goes (timers on: perturbed)
for ov in -1 0 1; do
python3 bench.py --halo --lat 48 48 48 12 --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 --set-option overlap=$ov 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
m=d['multi_gpu']
print('overlap=$ov', 'us/iteration', round(1e3*d['ms_per_step'],1), d['repeats']['ms_per_step'], 'sweep us', d['dslash_us_per_sweep'], 'anatomy', {k:m[k] for k in ('interior_us','boundary_us','exchange_us','allreduce_us','iteration_us_in_this_pass','overlap')})
"
done
QEX_BENCH_TIMERS=1 python3 bench.py --halo --lat 48 48 48 12 --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('all timers on:', 'us/iteration', round(1e3*d['ms_per_step'],1), 'kernel_ms per 200 its', d['kernel_ms'])
"
python3 bench.py --lat 48 48 48 12 --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('periodic 48^3x12 (no ghosts, no RCCL):', round(1e3*d['ms_per_step'],1), 'us/iteration; sweep', d['dslash_us_per_sweep'])
"
