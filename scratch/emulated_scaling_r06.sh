# round 6: the one-rank rehearsal of a rank's slab under emulated transport -- RCCL alone, RCCL faces + mailbox sums ("mbox", the
# auto choice between distinct devices of one node since round 6), and the peer-memory transport with its fused sweep.
# exchange: 3 us latency + bytes / 45 GB/s per xGMI direction (62 us for a 48^3 face); all-reduce: +15 us on top of the one-rank
# ncclAllReduce, +3 us inside the mailbox kernel.  Pessimistic rows: half the bandwidth, twice the latencies.
run() { env "$1" timeout -k 5 240 python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:2}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); m=d.get('multi_gpu',{}); sw=m.get('sweep',{})
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration; transport', d.get('transport'), 'overlap', sw.get('overlap'), 'form', sw.get('form'), 'exchange_us', round(sw.get('exchange_us',0),1), 'boundary_at', round(sw.get('boundary_at',0),2), 'tuned', [round(v) for v in sw.get('tuned_us_per_sweep',[])], 'anatomy', {k: m.get(k) for k in ('interior_us','boundary_us','exchange_us','allreduce_us')}, flush=True)" || { rc=$?; [ $rc -ge 124 ] && exit $rc; }; }
run QEXHIP_TRANSPORT=rccl --lat 48 48 48 96
for lt in 12 24 48; do
  run QEXHIP_TRANSPORT=rccl --halo --lat 48 48 48 $lt --emulate-transport 3 15 --set-option emu_link_gbs=45 --set-option overlap=-2
  run QEXHIP_TRANSPORT=mbox --halo --lat 48 48 48 $lt --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
  run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 $lt --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
done
run QEXHIP_TRANSPORT=rccl --halo --lat 48 48 48 12 --emulate-transport 6 30 --set-option emu_link_gbs=22 --set-option overlap=-2
run QEXHIP_TRANSPORT=mbox --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=-2
run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=-2
run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --set-option overlap=-2
run QEXHIP_TRANSPORT=mbox --halo --lat 48 48 48 12 --set-option overlap=-2
# the fused sweep with EVERY boundary block parked (cleanup workgroups do all slab-leaving hops): what a late neighbour costs at worst
run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2 --set-option fused_spin_us=-2
run QEXHIP_TRANSPORT=peer --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=0
# 32^4 (latency-dominated, SURVEY 8e: reported, not tuned for)
run QEXHIP_TRANSPORT=rccl --lat 32 32 32 32
for lt in 16 8 4; do
  run QEXHIP_TRANSPORT=mbox --halo --lat 32 32 32 $lt --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
  run QEXHIP_TRANSPORT=peer --halo --lat 32 32 32 $lt --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
done
# configs[4]'s solver: HISQ Naik links, one rank's slab
run QEXHIP_TRANSPORT=rccl --naik --lat 48 48 48 96
run QEXHIP_TRANSPORT=mbox --naik --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
run QEXHIP_TRANSPORT=peer --naik --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=-2
