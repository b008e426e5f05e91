#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -q --durations=8 > gpurun_out/tests_r03d.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/tests_r03d.log
timeout -k 10 400 python3 bench.py --no-cpu --no-48x96 > gpurun_out/bench_r03d.log 2> gpurun_out/bench_r03d.err; echo "bench rc=$?"
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_r03d.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','repeats')})
print(d['flow_step_32x4'].get('observables_kernel_us'), d['flow_step_32x4'].get('stage_kernel_us'), d['flow_step_32x4'].get('error'))
print(d['nhyp_smear_force_32x4'])
PY
