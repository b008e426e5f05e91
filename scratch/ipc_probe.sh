#!/bin/bash
# runs the IPC probe variants; stops at the first step that had to be killed (a hung GPU step must not be followed by another)
mkdir -p gpurun_out
out=gpurun_out/ipc_probe.log
: > $out
run() {
  echo "=== $*" >> $out
  timeout -k 10 150 "$@" >> $out 2>&1
  rc=$?
  echo "rc=$rc" >> $out
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "killed: stopping" >> $out; cat $out; exit 1; fi
}
run ./scratch/ipc_probe 2 200 1048576 0
run ./scratch/ipc_probe 2 200 1048576 1
run ./scratch/ipc_probe 2 200 1048576 2
run ./scratch/ipc_probe 4 200 2654208 0
run ./scratch/ipc_probe 2 50 201326592 0
cat $out
