#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python tools/emul_timing.py > gpurun_out/r2e_timing.log 2>&1 || { tail -5 gpurun_out/r2e_timing.log; exit 1; }
cat gpurun_out/r2e_timing.log
