#!/bin/bash
# one GPU-box visit: default bench line, the whole GPU test suite, then the rocprofv3 summaries.
# A step that was killed at its time limit ends the visit (no further GPU work after a hang).
set -o pipefail
TAG=${1:-r02}
mkdir -p gpurun_out
run() {   # run NAME LIMIT cmd... : rc 124/137 = killed -> stop everything
  local name=$1 limit=$2; shift 2
  echo "== $name: $*"
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2> gpurun_out/$name.err
  local rc=$?
  echo "== $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name was killed at its limit: stopping"; tail -5 gpurun_out/$name.err; exit 1; fi
  return $rc
}
run bench_$TAG 400 python3 bench.py
tail -c 6000 gpurun_out/bench_$TAG.log
run tests_$TAG 900 python3 -m pytest tests -m gpu -q -x --durations=15
tail -40 gpurun_out/tests_$TAG.log
if [ "$2" != "noprof" ]; then
  run collect_$TAG 600 bash profiles/collect.sh $TAG
  tail -30 gpurun_out/collect_$TAG.log
fi
exit 0
