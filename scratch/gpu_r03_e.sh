#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -q --durations=8 > gpurun_out/tests_r03e.log 2>&1; echo "tests rc=$?"; tail -40 gpurun_out/tests_r03e.log
