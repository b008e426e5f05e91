#!/bin/bash
# kernel timeline of an emulated slab iteration, peer arm: usage timeline_zc.sh <Lt> <lat_us> <gbs> <tag> [options as name=value ...]
export TMPDIR=/tmp
lt=$1; lat=$2; gbs=$3; tag=$4; shift 4
o=""; for kv in "$@"; do o="$o --set-option $kv"; done
rm -rf gpurun_out/tl_$tag
QEXHIP_TRANSPORT=${TL_TRANSPORT:-peer} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -- python3 bench.py --no-cpu --no-extra --no-48x96 --no-shard-check --steps 40 --warmup 10 --repeats 1 --halo ${TL_EXTRA:-} --lat 48 48 48 $lt --emulate-transport $lat ${TL_AR:-$lat} --set-option emu_link_gbs=$gbs --set-option overlap=1 $o > gpurun_out/tl_$tag.json 2> gpurun_out/tl_$tag.err
python3 scratch/timeline.py gpurun_out/tl_$tag 2 > gpurun_out/r05_timeline_peer_$tag.txt 2>&1
rm -rf gpurun_out/tl_$tag
echo "== $tag"; cat gpurun_out/r05_timeline_peer_$tag.txt
