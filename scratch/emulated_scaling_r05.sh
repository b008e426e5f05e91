# round 5: the one-rank rehearsal of a rank's slab under emulated transport, RCCL arm against the peer-memory arm.
# exchange: 3 us latency + bytes / 45 GB/s per xGMI direction (size-aware, option emu_link_gbs) -- 62 us for a 48^3 face;
# all-reduce: +15 us on top of the one-rank ncclAllReduce (round 4's figure), +3 us on top of the mailbox kernel (one xGMI store
# latency; the kernel itself is real).  Pessimistic column: half the bandwidth, twice the latencies.
run() { env "$1" timeout -k 5 240 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:2}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); m=d.get('multi_gpu',{}); sw=m.get('sweep',{})
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration; overlap', sw.get('overlap'), 'measured', sw.get('measured_us_per_sweep'), 'transport', d.get('transport'), 'anatomy', {k: m.get(k) for k in ('interior_us','boundary_us','exchange_us','allreduce_us')}, flush=True)" || exit 1; }
run QEXHIP_TRANSPORT=rccl --lat 48 48 48 96
for lt in 48 24 12; do
  run QEXHIP_TRANSPORT=rccl --halo --lat 48 48 48 $lt --emulate-transport 3 15 --set-option emu_link_gbs=45 --set-option overlap=-2
  run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 $lt --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
done
run QEXHIP_TRANSPORT=rccl --halo --lat 48 48 48 12 --emulate-transport 6 30 --set-option emu_link_gbs=22 --set-option overlap=-2
run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=-2
run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --set-option overlap=-2
run QEXHIP_TRANSPORT=rccl --halo --lat 48 48 48 12 --set-option overlap=-2
# 32^4 (latency-dominated, SURVEY 8e: reported, not tuned for): a 32^3 face is 0.79 MB = 20 us at 45 GB/s
run QEXHIP_TRANSPORT=rccl --lat 32 32 32 32
for lt in 16 8 4; do
  run QEXHIP_TRANSPORT=rccl --halo --lat 32 32 32 $lt --emulate-transport 3 15 --set-option emu_link_gbs=45 --set-option overlap=-2
  run QEXHIP_TRANSPORT=peer --halo --lat 32 32 32 $lt --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
done
# configs[4]'s solver: HISQ Naik links, one rank's slab
run QEXHIP_TRANSPORT=rccl --naik --lat 48 48 48 96
run QEXHIP_TRANSPORT=rccl --naik --halo --lat 48 48 48 12 --emulate-transport 3 15 --set-option emu_link_gbs=45 --set-option overlap=-2
run QEXHIP_TRANSPORT=peer --naik --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
