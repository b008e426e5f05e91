#!/bin/bash
# one GPU-box visit: default bench line, the whole GPU test suite, then the rocprofv3 summaries.
# A step that was killed at its time limit ends the visit (no further GPU work after a hang).
set -o pipefail
TAG=${1:-r03}
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
run bench_$TAG 500 python3 bench.py
tail -c 1500 gpurun_out/bench_$TAG.log
run smoke_$TAG 200 python3 -c "import __graft_entry__ as g; g.smoke()"
tail -2 gpurun_out/smoke_$TAG.log
run tests_$TAG 900 python3 -m pytest tests -m gpu -q --durations=10
tail -20 gpurun_out/tests_$TAG.log
if [ "$2" != "noprof" ]; then
  run collect_$TAG 600 bash profiles/collect.sh $TAG
  tail -8 gpurun_out/collect_$TAG.log
fi
exit 0
