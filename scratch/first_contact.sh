#!/bin/bash
# first_contact.sh -- the ladder for the first node with MORE THAN ONE GPU.  In six rounds nothing of this
# tree has exchanged a byte between two distinct devices; every number in DESIGN.md section 5 is a one-GPU rehearsal.  This script is
# what to run first on such a node, bottom rung first, each rung under its own timeout with its log kept, stopping at the first rung
# that fails (non-zero exit) -- and immediately, without another GPU step, at the first one that had to be KILLED.
#
#   bash scratch/first_contact.sh [--dry-run] [--gpus N] [--out DIR]
#
#   rung 1  scratch/ipc_probe between distinct devices (IPC_PROBE_DISTINCT=1): hipIpc mapping across devices, a pushed payload with
#           every word checked, flag handoff, mailbox all-reduce, achieved GB/s -- for coarse-grained, fine-grained and uncached memory
#   rung 2  tests/two_rank_worker.py, 2 ranks on 2 devices: every operator / solver / gauge-sector result of each rank's slab against
#           the global oracle -- over RCCL alone, over RCCL + mailbox sums (what `auto` picks), over the peer-memory transport
#   rung 3  tests/shared_device_worker.py across devices: the fused sweep at 48^3 slabs (8 links, 16 links), real neighbours
#   rung 4  bench.py --gpus 2, 4, 8 (as many as the node has): the driver's own command line; every line self-verifies
#           (shard_check at 32^4 and 48^3 x 96 against the oracle-pinned fixture) and carries the multi_gpu block
# What the logs overwrite in DESIGN.md section 5: the 45 GB/s / 3 us link constants (rung 1: GB/s and handoff latency), the
# "predicted strong scaling" table (rung 4: cg_48x48x48x96.ms_per_step per N, its multi_gpu.sweep block: measured exchange us, where
# the boundary workgroups went, the three forms' timings), and the sentence "not yet executed: any exchange between distinct devices".
# Every rank of every rung is a FRESH child process started before anything touches a GPU (torch.distributed.run / fork-before-HIP):
# nothing is re-launched from a process that has initialised the GPU.
set -u
DRY=0; NG=0; OUT=""
while [ $# -gt 0 ]; do
  case "$1" in
    --dry-run) DRY=1;;
    --gpus) NG="$2"; shift;;
    --out) OUT="$2"; shift;;
    *) echo "unknown argument $1" >&2; exit 2;;
  esac
  shift
done
cd "$(dirname "$0")/.." || exit 2
[ -n "$OUT" ] || OUT="profiles/first_contact_$(date +%Y%m%d_%H%M%S)"
mkdir -p "$OUT" || exit 2
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 NCCL_DEBUG=${NCCL_DEBUG:-WARN}
if [ "$NG" -le 0 ] 2>/dev/null; then
  if [ $DRY = 1 ]; then NG=8; else NG=$(python3 -c "import qex_amd as q; print(q.device_count())" 2>/dev/null || echo 0); fi
fi
echo "first contact: $NG GPU(s), logs under $OUT, dry run $DRY" | tee "$OUT/summary.txt"
if [ "$NG" -lt 2 ]; then echo "needs at least two GPUs" | tee -a "$OUT/summary.txt"; exit 2; fi
port=29700
nrung=0
rung() {   # name timeout_s command...
  local name="$1" tmo="$2"; shift 2
  nrung=$((nrung + 1))
  port=$((port + 1))
  local log="$OUT/$(printf %02d $nrung)_$name.log"
  echo "--- rung $nrung: $name (limit ${tmo}s): $*" | tee -a "$OUT/summary.txt"
  if [ $DRY = 1 ]; then echo "dry run: not executed" > "$log"; eval "${FIRST_CONTACT_DRY_HOOK:-true}"; rc=$?
  else MASTER_PORT=$port timeout -k 15 "$tmo" "$@" > "$log" 2>&1; rc=$?; fi
  echo "    rc $rc" | tee -a "$OUT/summary.txt"
  if [ $rc -ge 124 ] && [ $rc -le 137 ]; then echo "rung $nrung had to be killed: no further GPU step" | tee -a "$OUT/summary.txt"; exit 3; fi
  if [ $rc -ne 0 ]; then echo "rung $nrung failed: stopping (log: $log)" | tee -a "$OUT/summary.txt"; exit 1; fi
}
launch() { echo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$1" --master-addr 127.0.0.1 --master-port $((port + 1)); }

[ $DRY = 1 ] || { [ -x scratch/ipc_probe ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 scratch/ipc_probe.cpp -o scratch/ipc_probe -lrt || exit 2; }
for kind in 0 1 2; do
  IPC_PROBE_DISTINCT=1 rung "ipc_probe_2ranks_kind$kind" 150 env IPC_PROBE_DISTINCT=1 ./scratch/ipc_probe 2 200 2654208 $kind
done
IPC_PROBE_DISTINCT=1 rung "ipc_probe_${NG}ranks" 150 env IPC_PROBE_DISTINCT=1 ./scratch/ipc_probe "$([ "$NG" -gt 8 ] && echo 8 || echo "$NG")" 200 2654208 1
for tr in rccl mbox peer; do
  rung "two_rank_worker_$tr" 600 env QEXHIP_TRANSPORT=$tr QEXHIP_PEER_TIMEOUT=60 $(launch 2) tests/two_rank_worker.py 16 16 16 32
done
rung "fused_sweep_48x48x48_8links" 300 env QEXHIP_TRANSPORT=peer QEXHIP_PEER_TIMEOUT=30 $(launch 2) tests/shared_device_worker.py 48 48 48 24 --distinct-devices
rung "fused_sweep_48x48x48_16links" 300 env QEXHIP_TRANSPORT=peer QEXHIP_PEER_TIMEOUT=30 $(launch 2) tests/shared_device_worker.py 48 48 48 24 --naik --distinct-devices
for n in 2 4 8; do
  [ "$n" -le "$NG" ] || continue
  rung "bench_gpus$n" 1100 $(launch $n) bench.py --gpus $n --steps 200 --warmup 20
done
echo "first contact: every rung passed" | tee -a "$OUT/summary.txt"
