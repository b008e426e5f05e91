# round 6: fused sweep of round 5's build (boundary workgroups spin) against this round's (short wait, park, cleanup), same box, alternating
OLD=scratch/r05_lib/libqexhip.so; NEW=qex_amd/libqexhip.so
run() { env "$1" timeout -k 5 240 python3 scratch/bench_with_lib.py "$2" --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:3}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration', flush=True)" || { rc=$?; [ $rc -ge 124 ] && exit $rc; }; }
for rep in 1 2; do
  for lib in $OLD $NEW; do
    run QEXHIP_TRANSPORT=peer $lib --halo --lat 32 32 32 4 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2
    run QEXHIP_TRANSPORT=peer $lib --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2
    run QEXHIP_TRANSPORT=peer $lib --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=1 --set-option hop_split=2
    run QEXHIP_TRANSPORT=peer $lib --naik --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2
  done
done
