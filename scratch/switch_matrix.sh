#!/bin/bash
# the GPU test suite under the A/B switches that select an alternative code path (they must stay green)
set -o pipefail
run() { echo "== $*"; env "$@" timeout -k 10 500 python3 -m pytest tests -m gpu -q -x -k "gauge or flow or golden or shape or parity or misc" 2>&1 | tail -2; }
run QEXHIP_FORCE_PAIR=0
run QEXHIP_RECT_FAST=0 QEXHIP_WLINE_LINES=0
run QEXHIP_FORCE_LDS=0
run QEXHIP_OBS_CLOVER=0 QEXHIP_FLOW_FUSED=0
run QEXHIP_COMM2=0 QEXHIP_OVERLAP=1
