#!/bin/bash
# one-rank rehearsal of the per-rank pieces of 48^3x96 and 32^4 at N = 2, 4, 8 (ghost zones, one-rank RCCL communicator,
# multi-rank reduction branches): microseconds per CG iteration, to compare with the whole lattice on the same GPU
cd $GRAFT_REPO_ROOT
run() { timeout -k 5 200 python3 bench.py --no-cpu --no-extra --no-48x96 --steps 200 --warmup 20 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', '->', round(1e3*d['ms_per_step'],1), 'us/iteration', d['cg_iters_per_s'], 'it/s')" || exit 1; }
run --lat 32 32 32 32
run --halo --lat 32 32 32 16
run --halo --lat 32 32 32 8
run --halo --lat 32 32 32 4
run --lat 48 48 48 96
run --halo --lat 48 48 48 48
run --halo --lat 48 48 48 24
run --halo --lat 48 48 48 12
