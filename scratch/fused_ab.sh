# round 6: the fused sweep before (round 5's build: boundary workgroups spin until the faces are in) and after (short wait, park,
# cleanup workgroups), and the plain kernels of both builds, on ONE box, alternating
OLD=scratch/r05_lib/libqexhip.so; NEW=qex_amd/libqexhip.so
for rep in 1 2; do
  for lib in $OLD $NEW; do
    timeout -k 5 200 python3 scratch/sweep_ab.py $lib || exit 1
    timeout -k 5 200 python3 scratch/sweep_ab.py $lib --naik || exit 1
  done
done
run() { env "$1" timeout -k 5 240 python3 scratch/bench_with_lib.py "$2" --no-cpu --no-extra --no-48x96 --no-shard-check --steps 200 --warmup 20 --repeats 3 "${@:3}" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration', flush=True)" || { rc=$?; [ $rc -ge 124 ] && exit $rc; }; }
for rep in 1 2; do
  for lib in $OLD $NEW; do
    run QEXHIP_TRANSPORT=peer $lib --halo --lat 32 32 32 4 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2
    run QEXHIP_TRANSPORT=peer $lib --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2
    run QEXHIP_TRANSPORT=peer $lib --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=1 --set-option hop_split=2
  done
done
for ncl in; do   # (the QEXHIP_TUNE_FUSED_NCL hook of this A/B left the library with the decision: 32; profiles/r06_fused_ab.log)
  export QEXHIP_TUNE_FUSED_NCL=$ncl
  echo "ncl $ncl"
  run QEXHIP_TRANSPORT=peer $NEW --halo --lat 32 32 32 4 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2
  run QEXHIP_TRANSPORT=peer $NEW --halo --lat 48 48 48 12 --emulate-transport 6 6 --set-option emu_link_gbs=22 --set-option overlap=1 --set-option hop_split=2
  run QEXHIP_TRANSPORT=peer $NEW --halo --lat 48 48 48 12 --emulate-transport 3 3 --set-option emu_link_gbs=45 --set-option overlap=1 --set-option hop_split=2 --set-option fused_spin_us=-2
done
unset QEXHIP_TUNE_FUSED_NCL
timeout -k 5 200 python3 scratch/su3_gather_plaq.py
