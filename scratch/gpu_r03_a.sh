#!/bin/bash
# round 3, first GPU visit: shard-check fixture, GPU tests, default bench, --halo bench, 2-rank control-flow rehearsal
set -o pipefail
mkdir -p gpurun_out
run() {
  local name=$1 limit=$2; shift 2
  echo "== $name: $*"
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err
  local rc=$?
  echo "== $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name was killed at its limit: stopping"; tail -5 gpurun_out/$name.err; exit 1; fi
  return $rc
}
run shardfix 500 python3 tests/golden/make_shard_checks.py --out gpurun_out/shard_checks.json || { tail -20 gpurun_out/shardfix.err; exit 1; }
tail -5 gpurun_out/shardfix.log
cp gpurun_out/shard_checks.json tests/golden/shard_checks.json
run tests_r03a 900 python3 -m pytest tests -m gpu -q -x --durations=10
tail -25 gpurun_out/tests_r03a.log
run bench_r03a 500 python3 bench.py
tail -c 3000 gpurun_out/bench_r03a.log
run bench_r03a_halo 300 python3 bench.py --halo --no-cpu
tail -c 1500 gpurun_out/bench_r03a_halo.log
run bench_r03a_n2 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-no-rccl --no-extra --steps 40 --warmup 5 --repeats 2
tail -c 1500 gpurun_out/bench_r03a_n2.log
exit 0
